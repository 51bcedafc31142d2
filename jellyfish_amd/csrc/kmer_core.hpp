// jellyfish_amd/csrc/kmer_core.hpp
//
// Pure arithmetic of the hot path, shared by the HIP kernels (device) and the
// host-side geometry/matrix set-up.  No memory traffic here: character classes,
// 2-bit packing, reverse complement, GF(2) hash by byte tables, slot word
// packing.  Reference semantics being reproduced (paths relative to
// /root/reference):
//   include/jellyfish/mer_dna.hpp:38-55        character -> code
//   include/jellyfish/mer_iterator.hpp:53-81   rolling forward / reverse-complement mers
//   include/jellyfish/mer_dna.hpp:227-250      canonical = numerically smaller
//   include/jellyfish/rectangular_binary_matrix.hpp:223-261  pos = M * key over GF(2)
//   include/jellyfish/large_hash_array.hpp:509-597  quotienting: only key bits above lsize stored
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define JF_HD __host__ __device__ __forceinline__
#else
#define JF_HD inline
#endif

namespace jfgpu {

constexpr uint32_t kMaxTileBits = 13;   // probe domain = 8192 slots = 64 KiB: fits LDS, see DESIGN.md
constexpr uint32_t kMinCountBits = 16;  // in-slot count field never narrower than this (64-bit slots)
constexpr uint32_t kMinCountBits32 = 8; // ... and in a 32-bit slot

// ---- character classes ----------------------------------------------------
// A a->0, C c->1, G g->2, T t->3; everything else (N, IUPAC, '\n', ...) -> 4 = reset.
JF_HD uint32_t base_code(uint32_t c) {
  const uint32_t u = c & 0xDFu;                 // fold ASCII case
  const uint32_t x = (u >> 1) & 3u;             // A:0 C:1 G:3 T:2
  const uint32_t code = x ^ (x >> 1);           // A:0 C:1 G:2 T:3
  // letters 0x41 'A', 0x43 'C', 0x47 'G', 0x54 'T' as a 32-entry bit set over 0x40..0x5F
  const uint32_t valid_set = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 20);
  const bool ok = ((u ^ 0x40u) < 32u) && ((valid_set >> (u & 31u)) & 1u);
  return ok ? code : 4u;
}

// Pack 16 characters (little-endian in 4 dwords) into a 32-bit code word and a
// 16-bit invalid mask.  Base j (0 = first character) sits at bits [2(15-j)+1 : 2(15-j)]
// of codes and bit (15-j) of inval, so that earlier bases are MORE significant
// and consecutive words concatenate into one big-endian base stream.
//
// Four characters at a time inside one dword (round 4; the per-character version cost 16 vector instructions per
// character -- a fifth of the partition kernels' issue slots -- this one about 5):
//   u      = w & 0xDFDFDFDF                        ASCII case folded, all four bytes
//   code   = ((u >> 1) ^ (u >> 2)) & 0x03030303    A 0, C 1, G 2, T 3 (bits 1..3 of the letter; stays inside its byte)
//   letter = "ACGT"[code], byte-wise               one v_perm_b32: what u must be for the character to be a base
//   bad    = (u ^ letter) != 0, byte-wise          exact zero-byte test, flag in bit 7 of the byte
// and the 2-bit / 1-bit fields of a dword are gathered with two shift-or steps each (no cross-talk: see the field
// positions in the comments below).
JF_HD uint32_t letters_of_codes(uint32_t code) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(0u, 0x54474341u, code);      // selector bytes 0..3 pick bytes of the second operand
#else
  uint32_t r = 0;
  for(int b = 0; b < 4; ++b) r |= (uint32_t)"ACGT"[(code >> (8 * b)) & 3u] << (8 * b);
  return r;
#endif
}
JF_HD uint32_t bitrev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bitreverse32(x);
#else
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
  x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
  return (x >> 16) | (x << 16);
#endif
}
// One dword = characters 0..3 (byte 0 first).  Returns its four codes as one byte in bits 24..31 (character 0 most
// significant; lower bits are junk) and its four invalid flags in bits 21..24 (character 0 at bit 24; other bits junk).
JF_HD void pack4(uint32_t w, uint32_t& codes_hi8, uint32_t& inv_21_24) {
  const uint32_t u = w & 0xDFDFDFDFu;
  const uint32_t code = ((u >> 1) ^ (u >> 2)) & 0x03030303u;
  const uint32_t t = u ^ letters_of_codes(code);
  const uint32_t bad = (((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;      // bit 7 of a byte: that byte of t is not zero
  // codes: fields at bits 0, 8, 16, 24 -> 30, 28, 26, 24.  y adds copies shifted by 10, z copies shifted by 20: into bits
  // 24..31 fall char 3 (24 + 0), char 2 (16 + 10), char 1 (8 + 20), char 0 (0 + 10 + 20) and nothing else
  const uint32_t y = (code << 10) | code;
  codes_hi8 = (y << 20) | y;
  // flags: bit 7 of byte j -> reversed to bit 24 - 8j; copies shifted by 7 and 14: bits 21..24 get char 3 (0 + 21),
  // char 2 (8 + 14), char 1 (16 + 7), char 0 (24 + 0) and nothing else
  const uint32_t f = bitrev32(bad);
  const uint32_t g = (f << 7) | f;
  inv_21_24 = (g << 14) | g;
}
JF_HD void pack16(const uint32_t w[4], uint32_t& codes, uint32_t& inval) {
  uint32_t c[4], v[4];
#pragma unroll
  for(int i = 0; i < 4; ++i) pack4(w[i], c[i], v[i]);
#if defined(__HIP_DEVICE_COMPILE__)
  // the top bytes of c[0..3], in that order, most significant first
  const uint32_t hi = __builtin_amdgcn_perm(c[0], c[1], 0x07030000u), lo = __builtin_amdgcn_perm(c[2], c[3], 0x00000703u);
  codes = __builtin_amdgcn_perm(hi, lo, 0x07060100u);
#else
  codes = (c[0] & 0xFF000000u) | ((c[1] >> 8) & 0x00FF0000u) | ((c[2] >> 16) & 0x0000FF00u) | (c[3] >> 24);
#endif
  inval = ((v[0] >> 9) & 0xF000u) | ((v[1] >> 13) & 0x0F00u) | ((v[2] >> 17) & 0x00F0u) | ((v[3] >> 21) & 0x000Fu);
}

// (the per-character statement of the same function: what the one above must equal for every input)
JF_HD void pack16_ref(const uint32_t w[4], uint32_t& codes, uint32_t& inval) {
  uint32_t c = 0, v = 0;
#pragma unroll
  for(int i = 0; i < 4; ++i) {
#pragma unroll
    for(int b = 0; b < 4; ++b) {
      const uint32_t code = base_code((w[i] >> (8 * b)) & 0xFFu);
      c = (c << 2) | (code & 3u);
      v = (v << 1) | (code >> 2);
    }
  }
  codes = c;
  inval = v;
}

// Reverse complement of a k-mer held in the low 2k bits of x (k <= 32).
JF_HD uint64_t revcomp64(uint64_t x, uint32_t k) {
  x = ~x;
  x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
  x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
  x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
  x = (x >> 32) | (x << 32);
  return x >> (64 - 2 * k);
}

// ---- table geometry ---------------------------------------------------------
// One slot = one 64-bit word:   [ count : cnt_bits | occupied : 1 | tag : tag_bits ]
// -- or the same three fields in ONE 32-bit word when the tag is short enough (tag_bits <= 23, i.e. at most ten key bits
// left to store: k = 21 at 2^32 slots and up), which halves the table and what every flush of the partitioned path
// streams.  The reference packs entries at bit granularity for the same reason (offsets_key_value.hpp:87-106: 23.4 bits
// per entry at k = 21 / 2^34); here the unit stays a naturally aligned word so that one atomic claims a slot.
//   tag = (idx0 << rem_bits) | rem
//   rem  = key >> lsize_g              (key bits the hash position does not determine)
//   idx0 = position inside the tile    (low tile_bits bits of the hash position)
// The tile index (remaining position bits) is implied by the slot address, the
// shard by the owning GPU.  Probing is triangular and wraps inside the tile, so
// (tile, tag) identifies the key exactly wherever in the tile it finally lands.
// Numeric order of tags inside a tile == the reference's (pos, key) dump order.
// Count at the TOP: an atomic add that overflows the field carries out of the
// word and cannot corrupt the tag; the wrap is detected from the returned old
// value and spilled to the overflow side table (the reference's "large entry"
// idea, offsets_key_value.hpp:35-46).
struct TableGeom {
  uint32_t k, key_bits;
  uint32_t lsize_g, lsize_l;      // log2 global / local (this shard) slots
  uint32_t shard_bits, shard_id;
  uint32_t tile_bits, rem_bits, tag_bits, cnt_bits;
  uint32_t nbytes;                // bytes of a key fed to the hash tables = ceil(2k/8)
  uint32_t canonical;
  uint32_t slot32;                // 1: slots are 32-bit words (same fields, cnt_bits = 31 - tag_bits)
  uint32_t hash_xs;               // 1: the table's matrix is the xor-shift one (xs_hash below), so kernels may compute it in registers
  uint64_t key_mask, tile_mask, rem_mask, local_mask;
  uint64_t occ_bit, low_mask, inc, cnt_max;
};

// Fills every derived field from (k, lsize_g, shard_bits, shard_id, canonical).
// Returns false when the combination cannot be packed into a 64-bit slot.
inline bool geom_init(TableGeom& g, uint32_t k, uint32_t lsize_g, uint32_t shard_bits, uint32_t shard_id,
                      uint32_t canonical, bool allow32 = true) {
  if(k < 1 || k > 32 || lsize_g > 2 * k || shard_bits > lsize_g) return false;
  g.k = k; g.key_bits = 2 * k; g.lsize_g = lsize_g; g.shard_bits = shard_bits; g.shard_id = shard_id;
  g.lsize_l = lsize_g - shard_bits;
  g.tile_bits = g.lsize_l < kMaxTileBits ? g.lsize_l : kMaxTileBits;
  g.rem_bits = g.key_bits - lsize_g;
  g.tag_bits = g.tile_bits + g.rem_bits;
  g.slot32 = allow32 && g.tag_bits + 1 + kMinCountBits32 <= 32; g.hash_xs = 0;
  if(!g.slot32 && g.tag_bits + 1 + kMinCountBits > 64) return false;
  g.cnt_bits = (g.slot32 ? 31 : 63) - g.tag_bits;
  g.nbytes = (g.key_bits + 7) / 8;
  g.canonical = canonical;
  g.key_mask = g.key_bits == 64 ? ~0ull : ((1ull << g.key_bits) - 1);
  g.tile_mask = (1ull << g.tile_bits) - 1;
  g.rem_mask = g.rem_bits == 0 ? 0 : ((1ull << g.rem_bits) - 1);
  g.local_mask = (1ull << g.lsize_l) - 1;
  g.occ_bit = 1ull << g.tag_bits;
  g.low_mask = (g.occ_bit << 1) - 1;
  g.inc = g.occ_bit << 1;
  g.cnt_max = (1ull << g.cnt_bits) - 1;
  return true;
}

// Smallest global lsize the slot format admits for this k (so that cnt_bits >= kMinCountBits).
inline uint32_t geom_min_lsize(uint32_t k, uint32_t shard_bits) {
  // tag_bits = tile_bits + 2k - lsize_g <= 63 - kMinCountBits, tile_bits <= kMaxTileBits
  int need = (int)(2 * k + kMaxTileBits) - (int)(63 - kMinCountBits);
  if(need < 0) need = 0;
  if((uint32_t)need < shard_bits) need = (int)shard_bits;
  if((uint32_t)need > 2 * k) need = (int)(2 * k);
  return (uint32_t)need;
}

// (hi : lo) >> s, low dword, 0 <= s < 32 (v_alignbit_b32)
JF_HD uint32_t funnel_r(uint32_t hi, uint32_t lo, uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(hi, lo, s);
#else
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31u));
#endif
}

// pos = M * key via byte tables: tbl[b * 256 + v] = XOR of the columns selected by
// byte b of the key having value v.  (H is linear over GF(2).)
template <int NB>
JF_HD uint64_t hash_tables_n(const uint64_t* tbl, uint64_t key) {
  // constant byte positions: the key bytes come out of the two dwords with v_bfe_u32 / SDWA, no 64-bit shifts; the two
  // halves of the position are folded separately so that three table words meet in one v_bitop3_b32 (a 64-bit xor is
  // split into two plain v_xor_b32 after instruction selection and never becomes one)
  const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
  uint32_t plo = 0, phi = 0;
#pragma unroll
  for(int b = 0; b < NB; ++b) {
    const uint32_t w = b < 4 ? lo : hi;
    const uint64_t v = tbl[b * 256 + ((w >> (8 * (b & 3))) & 0xFFu)];
    plo ^= (uint32_t)v; phi ^= (uint32_t)(v >> 32);
  }
  return ((uint64_t)phi << 32) | plo;
}

// ---- the xor-shift matrix family (round 6) ----------------------------------------------------------------------
// The file format takes ANY r x 2k matrix whose low r x r block is invertible (readers use the header's columns,
// include/jellyfish/file_header.hpp:35-64; rectangular_binary_matrix.hpp:155-164 is a plain GF(2) product).  The
// reference draws a random one, which a GPU can only apply through table look-ups: six random 8-byte LDS reads per k-mer
// at k = 21, the bank conflicts and the waits behind them (profiles/r05_C2_sq_counters.txt).  This family is linear too --
// every step below is an xor of shifted copies -- but is evaluated with eleven register instructions:
//   r <= 32:  y = (key ^ key >> 9 ^ key >> 21 [^ key >> 32]) mod 2^r   the high key bits (those a slot stores) come down
//             y ^= y << 13 ;  y ^= y << 7  (mod 2^r) ;  y ^= y >> 17
//   r >  32:  lo = the same four steps mod 2^32                  (one dword of work)
//             hi = (key >> 32 ^ lo >> 9) mod 2^(r - 32)           position = hi : lo
// Restricted to the low r key bits, step one is unit upper triangular, the shift steps are invertible, and hi is the
// key's own bits plus a function of lo: the low block is invertible by construction (and checked when a table is made).
// The shifts were chosen on skewed, repeat-rich sequence (order-3 Markov chain + tandem repeats, 10 M distinct 21-mers):
// P1 bucket and tile occupancies as even as under random matrices (structural excess <= 0.4 % at r = 26, 30, 34), where
// the identity low block a CRC-like matrix has piles 36 x the mean into one tile.
// gf2_matrix.hpp builds the matrix column by column from this very function, so tables, headers, look-ups and dumps
// (which all go through the generic byte tables) agree with the kernels that evaluate it directly.
constexpr uint32_t kXsR0a = 9, kXsR0b = 21, kXsL1 = 13, kXsL2 = 7, kXsR3 = 17, kXsFold = 9;
JF_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}
JF_HD uint32_t xor_and(uint32_t a, uint32_t b, uint32_t mask) {        // (a ^ b) & mask
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, mask, b, 0x48);
#else
  return (a ^ b) & mask;
#endif
}
// Keys of more than 53 bits (k >= 27) get one more term in the first step, key >> 32: without it the key bits above
// 32 + 21 never reach the low dword, and k = 31 piles 25 x the mean into one tile on the same test sequence (with it: as
// even as a random matrix).  Two-word keys (33 <= k <= 64) are folded to one word first, x = lo ^ hi ^ rotl(hi, 25), and go
// on as a 64-bit key: the low block is the one-word family's.
JF_HD bool xs_folds(uint32_t key_bits) { return key_bits > 53; }
// 32 < r < 64, on the two dwords of the key: hi_mask = 2^(r - 32) - 1
template <bool FOLD>
JF_HD void xs_hash_halves(uint32_t klo, uint32_t khi, uint32_t hi_mask, uint32_t& ylo, uint32_t& yhi) {
  uint32_t lo = xor3(klo, funnel_r(khi, klo, kXsR0a), funnel_r(khi, klo, kXsR0b));
  if(FOLD) lo ^= khi;
  lo ^= lo << kXsL1;
#ifndef JFGPU_XS_TWO_STEPS                                           /* ablation (round 6): what a shorter mix would buy */
  lo ^= lo << kXsL2;
#endif
  lo ^= lo >> kXsR3;
  ylo = lo; yhi = xor_and(khi, lo >> kXsFold, hi_mask);
}
// key_bits = 2k of a one-word key (it only decides the fold)
JF_HD uint64_t xs_hash(uint64_t key, uint32_t r, uint32_t key_bits) {
  const bool fold = xs_folds(key_bits);
  if(r > 32) {
    uint32_t lo, hi;
    const uint32_t hm = r >= 64 ? 0xFFFFFFFFu : ((1u << (r - 32)) - 1u);
    if(fold) xs_hash_halves<true>((uint32_t)key, (uint32_t)(key >> 32), hm, lo, hi);
    else xs_hash_halves<false>((uint32_t)key, (uint32_t)(key >> 32), hm, lo, hi);
    return ((uint64_t)hi << 32) | lo;
  }
  const uint32_t m = r >= 32 ? 0xFFFFFFFFu : ((1u << r) - 1u);
  uint32_t y = (uint32_t)(key ^ (key >> kXsR0a) ^ (key >> kXsR0b) ^ (fold ? key >> 32 : 0)) & m;
  y ^= (y << kXsL1) & m;
  y ^= (y << kXsL2) & m;
  y ^= y >> kXsR3;
  return y;
}
// two-word keys (lo = bits 0..63 of the key, hi = the bits above)
JF_HD uint64_t xs_hash_wide(uint64_t lo, uint64_t hi, uint32_t r) {
  const uint32_t h0 = (uint32_t)hi, h1 = (uint32_t)(hi >> 32);
  const uint32_t xlo = xor3((uint32_t)lo, h0, funnel_r(h0, h1, 7)), xhi = xor3((uint32_t)(lo >> 32), h1, funnel_r(h1, h0, 7));      // lo ^ hi ^ rotl64(hi, 25)
  return xs_hash(((uint64_t)xhi << 32) | xlo, r, 64);
}
constexpr int kHashXS = -1, kHashXSLow = -2;      // NB of the kernels' hash template: the position comes from xs_hash, no tables

// NB > 0: the number of key bytes is a compile-time constant of the kernel (no per-k-mer switch);
// NB == 0: decided at run time.
template <int NB>
JF_HD uint64_t hash_tables_t(const uint64_t* tbl, uint64_t key, uint32_t nbytes);

JF_HD uint64_t hash_tables(const uint64_t* tbl, uint64_t key, uint32_t nbytes) {
  switch(nbytes) {   // wave-uniform: one scalar branch, then a fully unrolled body
  case 1: return hash_tables_n<1>(tbl, key);
  case 2: return hash_tables_n<2>(tbl, key);
  case 3: return hash_tables_n<3>(tbl, key);
  case 4: return hash_tables_n<4>(tbl, key);
  case 5: return hash_tables_n<5>(tbl, key);
  case 6: return hash_tables_n<6>(tbl, key);
  case 7: return hash_tables_n<7>(tbl, key);
  default: return hash_tables_n<8>(tbl, key);
  }
}

template <int NB>
JF_HD uint64_t hash_tables_t(const uint64_t* tbl, uint64_t key, uint32_t nbytes) {
  if constexpr(NB == 0) return hash_tables(tbl, key, nbytes);
  else return hash_tables_n<NB>(tbl, key);
}

struct SlotAddr {
  uint64_t tile_base;  // first slot of the tile (local slot index)
  uint32_t idx0;       // home position inside the tile
  uint32_t shard;      // owning shard of the global position
};

JF_HD SlotAddr slot_addr(const TableGeom& g, uint64_t pos_g) {
  SlotAddr a;
  a.shard = (uint32_t)(pos_g >> g.lsize_l);
  const uint64_t local = pos_g & g.local_mask;
  a.idx0 = (uint32_t)(local & g.tile_mask);
  a.tile_base = local & ~g.tile_mask;
  return a;
}

JF_HD uint64_t make_tag(const TableGeom& g, uint64_t key, uint32_t idx0) {
  const uint64_t rem = g.lsize_g >= 64 ? 0 : (key >> g.lsize_g);
  return ((uint64_t)idx0 << g.rem_bits) | rem;
}

// Slot word -> key.  inv_tbl are byte tables of the inverse map
// (rem, pos_g) -> low lsize_g key bits.
JF_HD uint64_t slot_key(const TableGeom& g, const uint64_t* inv_tbl, uint64_t word, uint64_t tile_base) {
  const uint64_t tag = word & (g.occ_bit - 1);
  const uint64_t rem = tag & g.rem_mask;
  const uint64_t idx0 = tag >> g.rem_bits;
  const uint64_t pos_g = ((uint64_t)g.shard_id << g.lsize_l) | tile_base | idx0;
  const uint64_t v = (g.lsize_g >= 64 ? 0 : (rem << g.lsize_g)) | pos_g;
  const uint64_t lo = hash_tables(inv_tbl, v, g.nbytes);
  return (g.lsize_g >= 64 ? 0 : (rem << g.lsize_g)) | lo;
}

JF_HD uint64_t slot_count(const TableGeom& g, uint64_t word) { return word >> (g.tag_bits + 1); }

// triangular probe sequence inside a tile: idx0 + p(p+1)/2 (mod tile size) visits every
// slot of a power-of-two tile exactly once for p = 0 .. size-1.
JF_HD uint32_t probe_slot(uint32_t idx0, uint32_t p, uint32_t tile_mask) {
  return (idx0 + ((p * (p + 1)) >> 1)) & tile_mask;
}

// One-word keys (round 3): slots are grouped in buckets of four that fill front to back, and probing is linear from the
// start of the home bucket (wrapping inside the tile).  That is what lets the LDS tile kernel place a flush's items by
// rank instead of by compare-and-swap (kernels_tile.hip.hpp); the global-atomic path, look-ups and growth follow the same
// sequence with one claim per probe as before.  Keys of two and more words keep the triangular sequence above.
// (Round 6 measured buckets of EIGHT -- -DJFGPU_BUCKET_BITS=3, the tile kernel is written over the bucket size --: 1.0 % of
// the items go past their bucket instead of 3.85 %, but looking at every bucket for equal tags compares 28 pairs per
// eight slots instead of 12 and the wider buckets spill registers: T 24.8 -> 32.2 ms on the metric's job, 30.1 -> 43.3 on
// distribution G, k = 31 19.5 -> 24.2.  profiles/r06_bucket8.log)
#ifndef JFGPU_BUCKET_BITS
#define JFGPU_BUCKET_BITS 2
#endif
constexpr uint32_t kBucketBits = JFGPU_BUCKET_BITS;
JF_HD uint32_t probe_lin(uint32_t idx0, uint32_t p, uint32_t tile_mask) {
  return ((idx0 & ~((1u << kBucketBits) - 1u)) + p) & tile_mask;
}

// One aligned 16-byte vector of sequence (global_load_dwordx4 on the device).
JF_HD void load16(const uint8_t* p, uint32_t w[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
#else
  for(int i = 0; i < 4; ++i)
    w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
#endif
}

constexpr int kPerLane = 16;  // sequence positions handled by one lane per tile

// ---- sequence tile -> per-lane packed words -----------------------------------
// Loads 16 sequence bytes for this lane (positions [off, off+16) of the aligned
// buffer), forcing everything outside [lo, hi) invalid.
JF_HD void load_pack16(const uint8_t* __restrict__ base, int64_t off, int64_t lo, int64_t hi,
                                   uint32_t& codes, uint32_t& inval) {
  if(off + 16 <= lo || off >= hi || off < 0) { codes = 0; inval = 0xFFFFu; return; }
  uint32_t w[4];
  if(off + 16 <= hi) {
    load16(base + off, w);
  } else {  // ragged tail: never read past the caller's buffer
    w[0] = w[1] = w[2] = w[3] = 0;
    for(int i = 0; i < 16 && off + i < hi; ++i) w[i >> 2] |= (uint32_t)base[off + i] << (8 * (i & 3));
  }
  pack16(w, codes, inval);
  if(off < lo) inval |= (0xFFFFu << (16 - (int)(lo - off))) & 0xFFFFu;        // first (lo-off) positions
  if(off + 16 > hi) inval |= (1u << (int)(off + 16 - hi)) - 1u;               // last positions
}

// Per-lane k-mer extraction state for one tile.
struct LaneWords {
  uint32_t cur, p1, p2;     // code words: own 16 bases, previous 16, the 16 before those
  uint64_t inv48;           // invalid bits of the same 48 positions (bit 15-j of each 16)
};

// Calls f(j, key) for every valid (canonical) k-mer ending at one of this lane's 16
// positions, in order.  mer_iterator.hpp:67-76 + :51.
template <typename F>
JF_HD void for_each_kmer(const TableGeom& g, const LaneWords& L, F&& f) {
  const uint32_t k = g.k;
  uint64_t fw = (((uint64_t)L.p2 << 32) | L.p1) & g.key_mask;   // k-mer ending just before this lane
  uint64_t rc = revcomp64(fw, k);
  const uint64_t kwin = k >= 64 ? ~0ull : ((1ull << k) - 1);
  const uint32_t rc_shift = 2 * (k - 1);
#pragma unroll
  for(int j = 0; j < kPerLane; ++j) {
    const uint64_t c = (L.cur >> (2 * (15 - j))) & 3u;
    fw = ((fw << 2) | c) & g.key_mask;
    rc = (rc >> 2) | ((3ull - c) << rc_shift);
    const bool valid = ((L.inv48 >> (15 - j)) & kwin) == 0;
    if(valid) f(j, (g.canonical && rc < fw) ? rc : fw);
  }
}

}  // namespace jfgpu
