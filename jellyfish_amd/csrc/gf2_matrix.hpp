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
// Construction differs from the reference on purpose: it draws from unseeded
// glibc random(); we use a seeded splitmix64 so every shard / rank derives the
// same matrix from a seed.  Any such matrix is legal for the file format
// (readers take it from the header, file_header.hpp:35-47).
#pragma once
#include <stdint.h>
#include <vector>

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
