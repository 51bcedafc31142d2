// jellyfish/mer_overlap_sequence_parser.hpp (compat): FASTA / FASTQ files -> buffers of sequence in which every k-mer
// appears exactly once (include/jellyfish/mer_overlap_sequence_parser.hpp:30-185).  One reader thread (the engine's host
// reader, same observable behaviour) fills a bounded queue, any number of consumers (mer_iterator) take buffers from it.
#pragma once
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <jellyfish_amd/sequence_parser.hpp>
namespace jellyfish {
template <typename StreamIterator>
class mer_overlap_sequence_parser {
public:
  // (mer_len, max_producers, size = number of buffers, buf_size, streams): buf_size is taken as a hint only
  mer_overlap_sequence_parser(uint16_t mer_len, uint32_t max_producers, uint32_t size, size_t buf_size, StreamIterator& streams)
      : k_(mer_len), depth_(size < 4 ? 4 : size) {
    (void)max_producers; (void)buf_size;
    paths_ = streams.paths();
    reader_ = std::thread([this]() { this->read_all(); });
  }
  ~mer_overlap_sequence_parser() {
    { std::lock_guard<std::mutex> l(mu_); abandoned_ = true; }
    space_.notify_all();
    if(reader_.joinable()) reader_.join();
  }
  mer_overlap_sequence_parser(const mer_overlap_sequence_parser&) = delete;

  // next buffer, nullptr when the input is exhausted (thread-safe); a read error is rethrown here
  std::unique_ptr<std::string> next() {
    std::unique_lock<std::mutex> l(mu_);
    data_.wait(l, [this]() { return !q_.empty() || done_; });
    if(q_.empty()) { if(!error_.empty()) throw std::runtime_error(error_); return nullptr; }
    std::unique_ptr<std::string> b = std::move(q_.front());
    q_.pop_front();
    space_.notify_one();
    return b;
  }
  uint16_t mer_len() const { return k_; }

private:
  void read_all() {
    try {
      jellyfish_amd::sequence_parser parser(k_, (size_t)1 << 20);
      for(const auto& path : paths_)
        parser.parse_file(path.c_str(), [this](const char* buf, size_t n) {
          std::unique_ptr<std::string> b(new std::string(buf, n));
          std::unique_lock<std::mutex> l(mu_);
          space_.wait(l, [this]() { return q_.size() < depth_ || abandoned_; });
          if(abandoned_) return;
          q_.push_back(std::move(b));
          data_.notify_one();
        });
    } catch(std::exception& e) { std::lock_guard<std::mutex> l(mu_); error_ = e.what(); }
    { std::lock_guard<std::mutex> l(mu_); done_ = true; }
    data_.notify_all();
  }
  uint16_t k_;
  size_t depth_;
  std::vector<std::string> paths_;
  std::thread reader_;
  std::mutex mu_;
  std::condition_variable data_, space_;
  std::deque<std::unique_ptr<std::string>> q_;
  bool done_ = false, abandoned_ = false;
  std::string error_;
};
}  // namespace jellyfish
