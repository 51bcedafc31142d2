// jellyfish/thread_exec.hpp (compat): start(id) on nb threads (include/jellyfish/thread_exec.hpp:27-60).
#pragma once
#include <thread>
#include <vector>
namespace jellyfish {
class thread_exec {
public:
  thread_exec() {}
  virtual ~thread_exec() {}
  virtual void start(int id) = 0;
  void exec(int nb_threads) {
    for(int i = 0; i < nb_threads; ++i) threads_.emplace_back([this, i]() { this->start(i); });
  }
  void join() {
    for(auto& t : threads_) t.join();
    threads_.clear();
  }
  void exec_join(int nb_threads) { exec(nb_threads); join(); }
private:
  std::vector<std::thread> threads_;
};
}  // namespace jellyfish
