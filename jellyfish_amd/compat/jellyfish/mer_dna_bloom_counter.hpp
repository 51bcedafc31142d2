// jellyfish/mer_dna_bloom_counter.hpp (compat): read side of a `jellyfish bc` file -- hash_pair<mer_dna> and
// mer_dna_bloom_counter::check (include/jellyfish/mer_dna_bloom_counter.hpp:19-40, bloom_counter2.hpp:40-142) -- the body
// read from the stream as the reference's constructor does, answered by the engine's bloom_query.
#pragma once
#include <istream>
#include <memory>
#include <vector>
#include <jellyfish_amd/dumpers.hpp>
#include <jellyfish/mer_dna.hpp>
#include <jellyfish/rectangular_binary_matrix.hpp>
namespace jellyfish {
template <typename Key> struct hash_pair;
template <>
struct hash_pair<mer_dna> {
  RectangularBinaryMatrix m1, m2;
  hash_pair() {}
  hash_pair(RectangularBinaryMatrix&& a, RectangularBinaryMatrix&& b) : m1(a), m2(b) {}
  void operator()(const mer_dna& k, uint64_t* hashes) const { hashes[0] = m1.times(k); hashes[1] = m2.times(k); }
};
class mer_dna_bloom_counter {
public:
  // bloom_counter2(size_t m, unsigned long k, std::istream& is, const HashPair& fns): m cells, k hashes, body from `is`
  mer_dna_bloom_counter(size_t m, unsigned long k, std::istream& is, const hash_pair<mer_dna>& fns) : body_(m / 5 + (m % 5 != 0)) {
    is.read(body_.data(), (std::streamsize)body_.size());
    q_.reset(new jellyfish_amd::bloom_query(body_.data(), (size_t)is.gcount(), m, (unsigned)k, fns.m1, fns.m2));
  }
  unsigned int check(const mer_dna& k) const { return q_->check(k); }
  unsigned int operator[](const mer_dna& k) const { return check(k); }
private:
  std::vector<char> body_;
  std::unique_ptr<jellyfish_amd::bloom_query> q_;
};
}  // namespace jellyfish
