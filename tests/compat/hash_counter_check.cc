// tests/compat/hash_counter_check.cc -- what unit_tests/test_hash_counter.cc (HashCounterCooperative.SizeDouble) checks,
// written without gtest against the same API (-Ijellyfish_amd/compat): a cooperative hash_counter of 128 slots takes
// 200 random 35-mers per thread with add(m, UINT64_MAX) -- or set(m) -- from worker threads, doubles itself, and a
// lazy iterator over ary() returns every key with the value the threads recorded.
#include <iostream>
#include <limits>
#include <map>
#include <vector>

#include <jellyfish/hash_counter.hpp>
#include <jellyfish/thread_exec.hpp>
#include <jellyfish/mer_dna.hpp>

using jellyfish::thread_exec;
using jellyfish::mer_dna;
typedef jellyfish::cooperative::hash_counter<mer_dna> hash_counter;
typedef hash_counter::array::lazy_iterator lazy_iterator;

enum OPERATION { ADD, SET };

class hash_adder : public thread_exec {
  typedef std::map<mer_dna, uint64_t> map;
  hash_counter& hash_;
  int nb_;
  std::vector<map> check_;
  OPERATION op_;

public:
  hash_adder(hash_counter& hash, int nb, int nb_threads, OPERATION op) : hash_(hash), nb_(nb), check_(nb_threads), op_(op) {}
  void start(int id) {
    mer_dna m;
    for(int i = 0; i < nb_; ++i) {
      m.randomize();
      if(op_ == ADD) hash_.add(m, std::numeric_limits<uint64_t>::max()); else hash_.set(m);
      check_[id][m] = std::numeric_limits<uint64_t>::max();
    }
    hash_.done();
  }
  uint64_t val(const mer_dna& m) const {
    uint64_t res = 0;
    for(const auto& mp : check_) { auto it = mp.find(m); if(it != mp.end()) res += it->second; }
    return res;
  }
  size_t keys() const { size_t n = 0; for(const auto& mp : check_) n += mp.size(); return n; }
};

#define CHECK(cond) do { if(!(cond)) { std::cerr << "FAILED line " << __LINE__ << ": " #cond "\n"; return 1; } } while(0)

int main() {
  const int mer_len = 35, nb = 200;
  const size_t init_size = 128;
  mer_dna::k(mer_len);
  for(int nb_threads : {1, 4}) {
    {
      hash_counter hash(init_size, mer_len * 2, 5, nb_threads);
      CHECK(hash.do_size_doubling());
      CHECK(hash.key_len() == mer_len * 2);
      CHECK(hash.val_len() == 5);
      hash_adder adder(hash, nb, nb_threads, ADD);
      adder.exec_join(nb_threads);
      lazy_iterator it = hash.ary()->iterator_all<lazy_iterator>();
      size_t seen = 0;
      while(it.next()) { CHECK(adder.val(it.key()) == it.val()); ++seen; }
      CHECK(seen == adder.keys());
      CHECK((size_t)(nb_threads * nb) < hash.size());
      bool is_new = true; size_t id = 0;
      mer_dna m; m.randomize();
      hash.add(m, 3, &is_new, &id); CHECK(is_new);
      hash.add(m, 4, &is_new, &id); CHECK(!is_new);
      uint64_t v = 0;
      CHECK(hash.ary()->get_val_for_key(m, &v) && v == 7);
      CHECK(hash.update_add(m, 10) && hash.ary()->get_val_for_key(m, &v) && v == 17);
      mer_dna other; other.randomize();
      CHECK(!hash.update_add(other, 1) && !hash.ary()->has_key(other));
    }
    {
      hash_counter hash(init_size, mer_len * 2, 0, nb_threads);
      CHECK(hash.val_len() == 0);
      hash_adder adder(hash, nb, nb_threads, SET);
      adder.exec_join(nb_threads);
      lazy_iterator it = hash.ary()->iterator_all<lazy_iterator>();
      size_t seen = 0;
      while(it.next()) { CHECK(it.val() == 0); CHECK(adder.val(it.key()) != 0); ++seen; }
      CHECK(seen == adder.keys());
      CHECK((size_t)(nb_threads * nb) < hash.size());
    }
  }
  std::cout << "OK\n";
  return 0;
}
