// tests/compat/count_dump.cc -- a client program written against the REFERENCE's C++ API (namespace jellyfish,
// include/jellyfish/*.hpp) in the way its examples are (examples/jf_count_dump): worker threads pull k-mers from a
// mer_iterator over a mer_overlap_sequence_parser and add them to a cooperative hash_counter, then the table is read
// back through get_val_for_key, the eager iterators and a lazy iterator.  Compiled with -Ijellyfish_amd/compat it runs on
// the MI355X engine; tests/test_compat.py compares its output with the oracle's counts.
//   usage: count_dump <k> <canonical 0|1> <threads> file...
#include <iostream>
#include <cstdlib>

#include <jellyfish/mer_dna.hpp>
#include <jellyfish/thread_exec.hpp>
#include <jellyfish/hash_counter.hpp>
#include <jellyfish/stream_manager.hpp>
#include <jellyfish/mer_overlap_sequence_parser.hpp>
#include <jellyfish/mer_iterator.hpp>

typedef jellyfish::cooperative::hash_counter<jellyfish::mer_dna> mer_hash_type;
typedef jellyfish::mer_overlap_sequence_parser<jellyfish::stream_manager<char**>> sequence_parser_type;
typedef jellyfish::mer_iterator<sequence_parser_type, jellyfish::mer_dna> mer_iterator_type;

class mer_counter : public jellyfish::thread_exec {
  mer_hash_type& mer_hash_;
  jellyfish::stream_manager<char**> streams_;
  sequence_parser_type parser_;
  const bool canonical_;

public:
  mer_counter(int nb_threads, mer_hash_type& mer_hash, char** file_begin, char** file_end, bool canonical)
      : mer_hash_(mer_hash), streams_(file_begin, file_end),
        parser_(jellyfish::mer_dna::k(), streams_.nb_streams(), 3 * nb_threads, 4096, streams_), canonical_(canonical) {}

  virtual void start(int) {
    mer_iterator_type mers(parser_, canonical_);
    for(; mers; ++mers) mer_hash_.add(*mers, 1);
    mer_hash_.done();
  }
};

int main(int argc, char* argv[]) {
  if(argc < 5) { std::cerr << "usage: count_dump k canonical threads file...\n"; return 2; }
  jellyfish::mer_dna::k(atoi(argv[1]));
  const bool canonical = atoi(argv[2]) != 0;
  const int threads = atoi(argv[3]);
  mer_hash_type mer_hash(1000, jellyfish::mer_dna::k() * 2, 7, threads, 126);     // far too small: the table has to double
  mer_counter counter(threads, mer_hash, argv + 4, argv + argc, canonical);
  counter.exec_join(threads);

  const auto jf_ary = mer_hash.ary();
  uint64_t total = 0, distinct = 0;
  const auto end = jf_ary->end();
  for(auto it = jf_ary->begin(); it != end; ++it) {
    auto& key_val = *it;
    std::cout << key_val.first << ' ' << key_val.second << '\n';
    total += key_val.second; ++distinct;
  }
  // the same through a lazy iterator and point look-ups
  typedef mer_hash_type::array::lazy_iterator lazy_iterator;
  lazy_iterator lit = jf_ary->iterator_all<lazy_iterator>();
  uint64_t total2 = 0, found = 0;
  while(lit.next()) {
    total2 += lit.val();
    uint64_t v = 0;
    if(found < 500 && jf_ary->get_val_for_key(lit.key(), &v) && v == lit.val()) ++found;
  }
  jellyfish::mer_dna absent;
  absent.polyA();
  uint64_t v = 0;
  const bool has_polyA = jf_ary->get_val_for_key(absent, &v);
  std::cerr << "distinct " << distinct << " total " << total << " total_lazy " << total2 << " lookups_ok " << found
            << " polyA " << (has_polyA ? v : 0) << " size " << mer_hash.size() << '\n';
  return total == total2 ? 0 : 1;
}
