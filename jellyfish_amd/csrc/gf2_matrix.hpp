// jellyfish_amd/csrc/gf2_matrix.hpp -- host-side GF(2) hash matrix.
//
// Same object as the reference's RectangularBinaryMatrix
// (include/jellyfish/rectangular_binary_matrix.hpp, lib/rectangular_binary_matrix.cc):
// r x c binary matrix, r = log2(table size) <= 64 rows, c = 2k columns, stored as
// one uint64 per column in FILE-HEADER order, where columns[c-1-j] is the image
// of key bit j (rectangular_binary_matrix.hpp:223-261).  What the engine needs
// from it:
//   * pos = M * key                               (hash; byte tables for the GPU)
//   * the low r x r block (images of key bits 0..r-1) invertible, so that the
//     low r key bits can be recovered from (pos, high key bits) -- the
//     reference's "pseudo inverse" (rectangular_binary_matrix.cc:160-210,
//     large_hash_array.hpp:992-1001).
// Two constructions.  Default: the reference's own -- it draws from glibc random(), which it never
// seeds, so for a given (lsize, 2k) the first matrix of a process is always the same one; GlibcRandom
// below restates that generator and gf2_reference_matrix the reference's randomize + pseudo-inverse, so
// default tables lay out, and their files read, exactly like the reference's (same (pos, key) order,
// byte-identical file bodies).  With an explicit seed: splitmix64, for callers that want their own
// family of matrices.  Any matrix with an invertible low block is legal for the file format (readers
// take it from the header, file_header.hpp:35-47).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include "kmer_core.hpp"

namespace jfgpu {

struct Gf2Matrix {
  uint32_t r = 0, c = 0;
  std::vector<uint64_t> columns;  // size c, file-header order
  bool identity = false;          // low identity (size == 4^k), large_hash_array.hpp:997-1000

  uint64_t col_for_bit(uint32_t j) const { return columns[c - 1 - j]; }

  uint64_t times(uint64_t key) const {  // single-word keys
    uint64_t res = 0;
    for(uint32_t j = 0; j < c && j < 64; ++j)
      if((key >> j) & 1) res ^= col_for_bit(j);
    return res;
  }
};

inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Inverse of the low r x r block B (B * lo = XOR_{j<r} lo_j * col_for_bit(j)).
// Returns false if singular.  binv_cols[i] = i-th column of B^-1.
inline bool gf2_invert_low_block(const Gf2Matrix& m, std::vector<uint64_t>& binv_cols) {
  const uint32_t r = m.r;
  // Row form of [B | I]: row i has bit j set iff bit i of col_for_bit(j) is set.
  std::vector<uint64_t> brow(r, 0), irow(r, 0);
  for(uint32_t i = 0; i < r; ++i) {
    for(uint32_t j = 0; j < r; ++j)
      if((m.col_for_bit(j) >> i) & 1) brow[i] |= 1ull << j;
    irow[i] = 1ull << i;
  }
  for(uint32_t col = 0; col < r; ++col) {
    uint32_t piv = col;
    while(piv < r && !((brow[piv] >> col) & 1)) ++piv;
    if(piv == r) return false;
    std::swap(brow[piv], brow[col]);
    std::swap(irow[piv], irow[col]);
    for(uint32_t i = 0; i < r; ++i)
      if(i != col && ((brow[i] >> col) & 1)) { brow[i] ^= brow[col]; irow[i] ^= irow[col]; }
  }
  // irow is now B^-1 in row form: lo_i = parity(irow[i] & y).  Convert to columns.
  binv_cols.assign(r, 0);
  for(uint32_t i = 0; i < r; ++i)
    for(uint32_t j = 0; j < r; ++j)
      if((irow[i] >> j) & 1) binv_cols[j] |= 1ull << i;
  return true;
}

inline Gf2Matrix gf2_identity(uint32_t r, uint32_t c) {
  Gf2Matrix m; m.r = r; m.c = c; m.identity = true; m.columns.assign(c, 0);
  // lib/rectangular_binary_matrix.cc:50-63 init_low_identity
  const uint32_t row = r < c ? r : c, col = c - row;
  m.columns[col] = 1ull << (row - 1);
  for(uint32_t i = col + 1; i < c; ++i) m.columns[i] = m.columns[i - 1] >> 1;
  return m;
}

// glibc random() as documented (TYPE_3 additive feedback generator: x_k = x_{k-3} + x_{k-31} mod 2^32,
// result x_k >> 1; the state is seeded from a Lehmer sequence and the first 310 values are discarded;
// seed 1 when srandom() is never called, which is the reference's case).
struct GlibcRandom {
  uint32_t x[34];
  uint32_t k = 0;                        // number of values produced so far, x[] is a ring of the last 34
  explicit GlibcRandom(uint32_t seed = 1) {
    int32_t v[34];
    v[0] = (int32_t)(seed ? seed : 1);
    for(int i = 1; i < 31; ++i) {
      const int64_t hi = v[i - 1] / 127773, lo = v[i - 1] % 127773;
      int64_t w = 16807 * lo - 2836 * hi;
      if(w < 0) w += 2147483647;
      v[i] = (int32_t)w;
    }
    for(int i = 31; i < 34; ++i) v[i] = v[i - 31];
    for(int i = 0; i < 34; ++i) x[i] = (uint32_t)v[i];
    k = 34;
    for(int i = 0; i < 310; ++i) step();
  }
  uint32_t step() {
    const uint32_t v = x[(k - 31) % 34] + x[(k - 3) % 34];
    x[k % 34] = v; ++k;
    return v;
  }
  uint32_t next() { return step() >> 1; }
  // random_bits(64), lib/misc.cc:66-72: three 31-bit draws XORed in at bit 0, 30 and 60
  uint64_t bits64() {
    uint64_t res = next();
    res ^= (uint64_t)next() << 30;
    res ^= (uint64_t)next() << 60;
    return res;
  }
};

// RectangularBinaryMatrix(r, c).randomize_pseudo_inverse() (lib/rectangular_binary_matrix.cc:160-247): draw
// c random columns, eliminate on the block of the last min(r, c) columns while applying the same column
// operations to a low-identity matrix; the transformed identity is the hash matrix.  A singular draw is
// thrown away and the next c columns are drawn, exactly as the reference's retry loop consumes them.
inline Gf2Matrix gf2_reference_matrix(uint32_t r, uint32_t c, GlibcRandom& rng) {
  Gf2Matrix m; m.r = r; m.c = c; m.columns.assign(c, 0);
  const uint64_t cmask = r >= 64 ? ~0ull : ((1ull << r) - 1);
  const uint32_t srow = r < c ? r : c, scol = c - srow;
  std::vector<uint64_t> pivot(c), res(c);
  while(true) {
    for(uint32_t i = 0; i < c; ++i) pivot[i] = rng.bits64() & cmask;
    for(uint32_t i = 0; i < c; ++i) res[i] = i >= scol ? (1ull << (c - 1 - i)) : 0ull;
    bool singular = false;
    uint64_t mask = 1ull << (srow - 1);
    for(uint32_t i = scol; i < c && !singular; ++i, mask >>= 1) {       // lower triangular
      if(!(pivot[i] & mask)) {
        uint32_t j = i + 1;
        while(j < c && !(pivot[j] & mask)) ++j;
        if(j == c) { singular = true; break; }
        pivot[i] ^= pivot[j]; res[i] ^= res[j];
      }
      for(uint32_t j = i + 1; j < c; ++j)
        if(pivot[j] & mask) { pivot[j] ^= pivot[i]; res[j] ^= res[i]; }
    }
    if(singular) continue;
    mask = 1ull << (srow - 1);
    for(uint32_t i = scol; i < c; ++i, mask >>= 1)                        // lower identity
      for(uint32_t j = 0; j < i; ++j)
        if(pivot[j] & mask) { pivot[j] ^= pivot[i]; res[j] ^= res[i]; }
    m.columns = res;
    return m;
  }
}

inline Gf2Matrix gf2_random(uint32_t r, uint32_t c, uint64_t seed) {
  if(r >= c) return gf2_identity(r, c);
  Gf2Matrix m; m.r = r; m.c = c; m.columns.assign(c, 0);
  const uint64_t cmask = r >= 64 ? ~0ull : ((1ull << r) - 1);
  uint64_t s = seed;
  std::vector<uint64_t> binv;
  while(true) {
    for(uint32_t i = 0; i < c; ++i) m.columns[i] = splitmix64(s) & cmask;
    if(gf2_invert_low_block(m, binv)) return m;
  }
}

// The xor-shift family (kmer_core.hpp: xs_hash, xs_hash_wide): the matrix of that linear map, column by column -- the image
// of key bit j is the function applied to the key with only that bit set.  Keys of one and two words (c <= 128); r >= c is
// the identity case like everywhere else.
inline uint64_t gf2_xorshift_image(uint32_t j, uint32_t r, uint32_t c) {
  if(c <= 64) return xs_hash(1ull << j, r, c);
  return j < 64 ? xs_hash_wide(1ull << j, 0, r) : xs_hash_wide(0, 1ull << (j - 64), r);
}
inline Gf2Matrix gf2_xorshift_matrix(uint32_t r, uint32_t c) {
  if(r >= c) return gf2_identity(r, c);
  Gf2Matrix m; m.r = r; m.c = c; m.columns.assign(c, 0);
  for(uint32_t j = 0; j < c; ++j) m.columns[c - 1 - j] = gf2_xorshift_image(j, r, c);
  return m;
}
// is this matrix the family's member for its shape?  (a table created from explicit columns -- a file header's -- gets the
// register hash too when it is)
inline bool gf2_is_xorshift(const Gf2Matrix& m) {
  if(m.c > 128 || m.r >= m.c || m.r >= 64 || m.columns.size() != m.c) return false;
  for(uint32_t j = 0; j < m.c; ++j) if(m.columns[m.c - 1 - j] != gf2_xorshift_image(j, m.r, m.c)) return false;
  return true;
}

inline bool gf2_is_low_identity(const Gf2Matrix& m) {  // rectangular_binary_matrix.cc:65-79
  const uint32_t row = m.r < m.c ? m.r : m.c, col = m.c - row;
  for(uint32_t i = 0; i < col; ++i) if(m.columns[i]) return false;
  if(m.columns[col] != 1ull << (row - 1)) return false;
  for(uint32_t i = col + 1; i < m.c; ++i) if(m.columns[i] != m.columns[i - 1] >> 1) return false;
  return true;
}

// Byte tables for the GPU: tbl[b*256 + v] = XOR_{i<8, bit i of v} image(bit 8b+i).
inline void gf2_byte_tables(const std::vector<uint64_t>& image_of_bit, uint32_t nbytes, std::vector<uint64_t>& tbl) {
  tbl.assign((size_t)nbytes * 256, 0);
  for(uint32_t b = 0; b < nbytes; ++b)
    for(uint32_t v = 0; v < 256; ++v) {
      uint64_t x = 0;
      for(uint32_t i = 0; i < 8; ++i) {
        const uint32_t bit = 8 * b + i;
        if(((v >> i) & 1) && bit < image_of_bit.size()) x ^= image_of_bit[bit];
      }
      tbl[(size_t)b * 256 + v] = x;
    }
}

// Forward tables (key -> pos) and inverse tables ((rem << r | pos) -> low r key bits).
inline bool gf2_build_tables(const Gf2Matrix& m, std::vector<uint64_t>& fwd, std::vector<uint64_t>& inv) {
  const uint32_t nbytes = (m.c + 7) / 8;
  std::vector<uint64_t> img(m.c);
  for(uint32_t j = 0; j < m.c; ++j) img[j] = m.col_for_bit(j);
  gf2_byte_tables(img, nbytes, fwd);
  std::vector<uint64_t> binv;
  if(!gf2_invert_low_block(m, binv)) return false;
  auto apply_binv = [&](uint64_t y) { uint64_t x = 0; for(uint32_t i = 0; i < m.r; ++i) if((y >> i) & 1) x ^= binv[i]; return x; };
  std::vector<uint64_t> inv_img(m.c);
  for(uint32_t i = 0; i < m.r; ++i) inv_img[i] = binv[i];                       // pos bit i
  for(uint32_t j = m.r; j < m.c; ++j) inv_img[j] = apply_binv(m.col_for_bit(j)); // rem bit (key bit j)
  gf2_byte_tables(inv_img, nbytes, inv);
  return true;
}

}  // namespace jfgpu
