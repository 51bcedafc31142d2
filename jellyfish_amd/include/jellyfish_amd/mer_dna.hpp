// jellyfish_amd/include/jellyfish_amd/mer_dna.hpp
//
// Host-side k-mer value type with the API shape of the reference's
// jellyfish::mer_dna (include/jellyfish/mer_dna.hpp:143-573, 625-717): a
// process-global k, ceil(k/32) little-endian 64-bit words, base i of the string
// (0 = leftmost) at bits 2(k-1-i), A=0 C=1 G=2 T=3.  data() is exactly the word
// array the C ABI (include/jfgpu.h) exchanges and binary_writer serialises
// (binary_dumper.hpp:37).  Written from the documented layout, not from the
// reference's source.
#pragma once
#include <cstdint>
#include <cstring>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <random>
#include <string>
#include <vector>

namespace jellyfish_amd {

class mer_dna {
public:
  // mer_dna.hpp:660-667: k is process-global and static
  static unsigned int k() { return k_; }
  static unsigned int k(unsigned int new_k) { unsigned int old = k_; k_ = new_k; return old; }
  static unsigned int nb_words(unsigned int k) { return (2 * k + 63) / 64; }

  mer_dna() : w_(nb_words(k_), 0), mk_(k_) {}
  explicit mer_dna(unsigned int k) : w_(nb_words(k), 0), mk_(k) {}
  explicit mer_dna(const char* s) : w_(nb_words(k_), 0), mk_(k_) { from_chars(s, strlen(s)); }
  explicit mer_dna(const std::string& s) : w_(nb_words(k_), 0), mk_(k_) { from_chars(s.data(), s.size()); }

  unsigned int mer_k() const { return mk_; }
  unsigned int nb_words() const { return (unsigned)w_.size(); }

  // mer_dna.hpp:38-55
  static int code(char c) {
    switch(c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
  }
  static char rev_code(int x) { return "ACGT"[x & 3]; }
  static int complement(int x) { return 3 - x; }
  static bool not_dna(int c) { return c < 0; }

  // operator=(string): throws std::length_error when too short (mer_dna.hpp:303-315)
  mer_dna& operator=(const std::string& s) { from_chars(s.data(), s.size()); return *this; }
  mer_dna& operator=(const char* s) { from_chars(s, strlen(s)); return *this; }

  bool from_chars(const char* s, size_t len) {
    if(len < mk_) throw std::length_error("Input string is to short");
    std::fill(w_.begin(), w_.end(), 0);
    for(unsigned i = 0; i < mk_; ++i) {
      int c = code(s[i]);
      if(c < 0) throw std::length_error(std::string("Invalid character '") + s[i] + "' in mer");
      unsigned j = mk_ - 1 - i;
      w_[j / 32] |= (uint64_t)c << (2 * (j % 32));
    }
    return true;
  }

  // base(i): i counts from the RIGHT end (mer_dna.hpp:261-262)
  int base(unsigned i) const { return (int)((w_[i / 32] >> (2 * (i % 32))) & 3); }
  void set_base(unsigned i, int c) {
    w_[i / 32] = (w_[i / 32] & ~((uint64_t)3 << (2 * (i % 32)))) | ((uint64_t)(c & 3) << (2 * (i % 32)));
  }

  // shift in a base on the right / left; returns the base pushed off (mer_dna.hpp:322-370)
  int shift_left(int c) {
    const int out = base(mk_ - 1);
    for(size_t i = w_.size(); i-- > 1;) w_[i] = (w_[i] << 2) | (w_[i - 1] >> 62);
    w_[0] = (w_[0] << 2) | (uint64_t)(c & 3);
    clean_msw();
    return out;
  }
  int shift_right(int c) {
    const int out = base(0);
    for(size_t i = 0; i + 1 < w_.size(); ++i) w_[i] = (w_[i] >> 2) | (w_[i + 1] << 62);
    w_.back() >>= 2;
    set_base(mk_ - 1, c);
    return out;
  }
  char shift_left(char c) { int x = code(c); if(x < 0) return 'N'; return rev_code(shift_left(x)); }
  char shift_right(char c) { int x = code(c); if(x < 0) return 'N'; return rev_code(shift_right(x)); }

  void reverse_complement() {
    mer_dna r(mk_);
    for(unsigned i = 0; i < mk_; ++i) r.set_base(mk_ - 1 - i, 3 - base(i));
    w_.swap(r.w_);
  }
  mer_dna get_reverse_complement() const { mer_dna r(*this); r.reverse_complement(); return r; }
  void canonicalize() { mer_dna r = get_reverse_complement(); if(r < *this) w_.swap(r.w_); }
  mer_dna get_canonical() const { mer_dna r = get_reverse_complement(); return r < *this ? r : *this; }

  bool operator==(const mer_dna& o) const { return mk_ == o.mk_ && w_ == o.w_; }
  bool operator!=(const mer_dna& o) const { return !(*this == o); }
  // numeric compare from the top word = lexicographic on the string (mer_dna.hpp:227-250)
  bool operator<(const mer_dna& o) const {
    for(size_t i = w_.size(); i-- > 0;) if(w_[i] != o.w_[i]) return w_[i] < o.w_[i];
    return false;
  }
  bool operator>(const mer_dna& o) const { return o < *this; }
  bool operator<=(const mer_dna& o) const { return !(o < *this); }
  bool operator>=(const mer_dna& o) const { return !(*this < o); }

  std::string to_str() const {
    std::string s(mk_, 'A');
    for(unsigned i = 0; i < mk_; ++i) s[i] = rev_code(base(mk_ - 1 - i));
    return s;
  }

  uint64_t word(unsigned i) const { return w_[i]; }
  uint64_t& word__(unsigned i) { return w_[i]; }
  const uint64_t* data() const { return w_.data(); }
  uint64_t* data__() { return w_.data(); }
  void clean_msw() {
    const unsigned top = (2 * mk_) & 63;
    if(top) w_.back() &= (~(uint64_t)0) >> (64 - top);
  }

  // get_bits(start, len) (mer_dna.hpp:467-498), len <= 64
  uint64_t get_bits(unsigned start, unsigned len) const {
    if(len == 0) return 0;
    const unsigned q = start / 64, r = start % 64;
    uint64_t res = w_[q] >> r;
    if(r && q + 1 < w_.size()) res |= w_[q + 1] << (64 - r);
    return len >= 64 ? res : res & (((uint64_t)1 << len) - 1);
  }

  // randomize() (mer_dna.hpp:344-352): every base uniform; one generator per thread
  void randomize() {
    static thread_local std::mt19937_64 gen(std::random_device{}());
    for(auto& w : w_) w = gen();
    clean_msw();
  }
  void randomize(uint64_t seed) {
    std::mt19937_64 gen(seed);
    for(auto& w : w_) w = gen();
    clean_msw();
  }
  void polyA() { std::fill(w_.begin(), w_.end(), 0); }
  void polyT() { std::fill(w_.begin(), w_.end(), ~(uint64_t)0); clean_msw(); }
  bool is_homopolymer() const {
    for(unsigned i = 1; i < mk_; ++i) if(base(i) != base(0)) return false;
    return true;
  }
  template <typename Rng> void randomize(Rng& rng) {
    for(auto& w : w_) w = ((uint64_t)rng() << 32) ^ (uint64_t)rng();
    clean_msw();
  }

  // read<1>(istream) of the binary/sorted body: ceil(2k/8) raw bytes (mer_dna.hpp:193-198)
  bool read_bytes(std::istream& is) {
    std::fill(w_.begin(), w_.end(), 0);
    is.read((char*)w_.data(), (2 * mk_ + 7) / 8);
    clean_msw();
    return is.good();
  }

private:
  std::vector<uint64_t> w_;
  unsigned int mk_;
  static inline unsigned int k_ = 22;
};

inline std::ostream& operator<<(std::ostream& os, const mer_dna& m) { return os << m.to_str(); }
inline std::istream& operator>>(std::istream& is, mer_dna& m) {
  std::string s;
  is >> s;
  if(is) { try { m = s; } catch(std::length_error&) { is.setstate(std::ios::failbit); } }
  return is;
}

}  // namespace jellyfish_amd
