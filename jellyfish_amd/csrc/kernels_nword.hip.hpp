// jellyfish_amd/csrc/kernels_nword.hip.hpp -- keys of more than two words, 65 <= k <= 128 (gfx950).
//
// The reference's mer_dna is any number of 64-bit words (include/jellyfish/mer_dna.hpp:143-170, 711-717) and its own
// test of the multi-word key path counts 100-mers (tests/large_key.sh:7-18).  Here such keys are 256-bit values in
// four-word slots, the two-word design of kernels_wide.hip.hpp with two more "lo" words:
//
//   slot = { lo0, lo1, lo2, hi }     lo_i = [ tag bits 63i .. 63i+62 | valid ]     hi = [ count | occ | tag >> 189 ]
//
// claim: CAS(hi, 0 -> occ|tag_hi), then CAS(lo_i, 0 -> lo_i) for i = 0, 1, 2: a lane goes on to word i+1 only while
// every word so far equals its own, so the lane that sets the last word matched all the others and the slot holds
// exactly that lane's key; a lane that meets a foreign word probes on, nobody waits (the reference's per-word "set"
// bits, offsets_key_value.hpp:28-31, large_hash_array.hpp:542-579).  Tiles are 2048 slots (64 KiB, what the sorted
// dump sorts in LDS).  Insert path: global atomics only (this range is a correctness feature, not a benchmark).
#pragma once
#include "kernels.hip.hpp"

namespace jfgpu {

constexpr uint32_t kNTileBits = 11;
constexpr uint32_t kNWords = 4;                 // words per key and per slot
constexpr uint32_t kNLoBits = 63 * (kNWords - 1);

struct K256 { uint64_t w[4]; };

JF_HD K256 k256_zero() { K256 r; r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0; return r; }
JF_HD bool k256_eq(const K256& a, const K256& b) { return a.w[0] == b.w[0] && a.w[1] == b.w[1] && a.w[2] == b.w[2] && a.w[3] == b.w[3]; }
JF_HD bool k256_less(const K256& a, const K256& b) {      // numeric, from the top word: mer_dna::operator< (mer_dna.hpp:227-250)
  for(int i = 3; i >= 0; --i) if(a.w[i] != b.w[i]) return a.w[i] < b.w[i];
  return false;
}
JF_HD K256 k256_and(const K256& a, const K256& b) { K256 r; for(int i = 0; i < 4; ++i) r.w[i] = a.w[i] & b.w[i]; return r; }
JF_HD K256 k256_or(const K256& a, const K256& b) { K256 r; for(int i = 0; i < 4; ++i) r.w[i] = a.w[i] | b.w[i]; return r; }
JF_HD K256 k256_shr(const K256& a, uint32_t n) {          // n < 256
  const uint32_t ws = n >> 6, bs = n & 63;
  K256 r;
  for(uint32_t i = 0; i < 4; ++i) {
    const uint64_t lo = i + ws < 4 ? a.w[i + ws] : 0, hi = i + ws + 1 < 4 ? a.w[i + ws + 1] : 0;
    r.w[i] = bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
  }
  return r;
}
JF_HD K256 k256_shl(const K256& a, uint32_t n) {          // n < 256
  const uint32_t ws = n >> 6, bs = n & 63;
  K256 r;
  for(int i = 3; i >= 0; --i) {
    const uint64_t hi = i >= (int)ws ? a.w[i - ws] : 0, lo = i >= (int)ws + 1 ? a.w[i - ws - 1] : 0;
    r.w[i] = bs ? (hi << bs) | (lo >> (64 - bs)) : hi;
  }
  return r;
}
JF_HD K256 k256_low_mask(uint32_t bits) {                  // 2^bits - 1, bits <= 256
  K256 r;
  for(uint32_t i = 0; i < 4; ++i) r.w[i] = bits >= 64 * (i + 1) ? ~0ull : (bits > 64 * i ? ((1ull << (bits - 64 * i)) - 1) : 0ull);
  return r;
}
JF_HD K256 k256_from64(uint64_t v) { K256 r = k256_zero(); r.w[0] = v; return r; }
JF_HD uint64_t k256_chunk63(const K256& a, uint32_t i) { return k256_shr(a, 63 * i).w[0] & 0x7FFFFFFFFFFFFFFFull; }

// the rolling updates of mer_iterator (mer_iterator.hpp:67-76): m = (m << 2 | code) & mask, rcm = rcm >> 2 | (3 - code) << 2(k-1)
JF_HD void k256_roll_fw(K256& x, uint64_t c, const K256& mask) {
  x.w[3] = ((x.w[3] << 2) | (x.w[2] >> 62)) & mask.w[3];
  x.w[2] = ((x.w[2] << 2) | (x.w[1] >> 62)) & mask.w[2];
  x.w[1] = ((x.w[1] << 2) | (x.w[0] >> 62)) & mask.w[1];
  x.w[0] = ((x.w[0] << 2) | c) & mask.w[0];
}
JF_HD void k256_roll_rc(K256& x, uint64_t v, uint32_t bitpos) {
  x.w[0] = (x.w[0] >> 2) | (x.w[1] << 62);
  x.w[1] = (x.w[1] >> 2) | (x.w[2] << 62);
  x.w[2] = (x.w[2] >> 2) | (x.w[3] << 62);
  x.w[3] = x.w[3] >> 2;
  x.w[bitpos >> 6] |= v << (bitpos & 63);
}
JF_HD K256 revcomp256(const K256& x, uint32_t k) {
  K256 r;                                                  // reverse all 128 2-bit groups and complement, then drop the padding
  for(int i = 0; i < 4; ++i) r.w[i] = revcomp64(x.w[3 - i], 32);
  return k256_shr(r, 256 - 2 * k);
}

struct NGeom {
  TableGeom g;          // tag_bits / occ_bit / low_mask / inc / cnt_* describe the HI word
  uint32_t tag_full;    // tile_bits + rem_bits
  uint32_t pad_[3];
  K256 key_mask;
};

inline uint32_t nword_min_lsize(uint32_t k) {
  int need = (int)(2 * k + kNTileBits) - (int)kNLoBits - (int)(63 - kMinCountBits);
  if(need < (int)kNTileBits) need = kNTileBits;
  return (uint32_t)need;
}
inline bool nword_geom_init(NGeom& N, uint32_t k, uint32_t lsize_g, uint32_t canonical) {
  TableGeom& g = N.g;
  if(k < 65 || k > 128 || lsize_g > 63 || lsize_g < kNTileBits) return false;
  memset(&g, 0, sizeof g);
  g.k = k; g.key_bits = 2 * k; g.lsize_g = g.lsize_l = lsize_g;
  g.tile_bits = kNTileBits;
  g.rem_bits = g.key_bits - lsize_g;
  N.tag_full = g.tile_bits + g.rem_bits;
  const uint32_t th = N.tag_full > kNLoBits ? N.tag_full - kNLoBits : 0;
  if(th + 1 + kMinCountBits > 64) return false;
  g.tag_bits = th; g.cnt_bits = 63 - th;
  g.nbytes = (g.key_bits + 7) / 8;
  g.canonical = canonical;
  g.key_mask = ~0ull;
  N.key_mask = k256_low_mask(g.key_bits);
  g.tile_mask = (1ull << g.tile_bits) - 1;
  g.local_mask = (1ull << g.lsize_l) - 1;
  g.occ_bit = 1ull << th;
  g.low_mask = (g.occ_bit << 1) - 1;
  g.inc = g.occ_bit << 1;
  g.cnt_max = (1ull << g.cnt_bits) - 1;
  return true;
}

struct NTable {
  NGeom N;
  uint64_t* slots;            // [4 << lsize]: slot s = slots[4s .. 4s+3] = { lo0, lo1, lo2, hi }
  const uint64_t* fwd_tbl;    // [nbytes * 256]
  const uint64_t* inv_tbl;
  uint64_t* ovf_key; uint64_t* ovf_cnt; uint64_t ovf_mask;
  uint64_t* counters;
  uint32_t max_probe;
};

__device__ inline DevTable ovf_view(const NTable& T) {
  DevTable d; d.g = T.N.g; d.slots = nullptr; d.fwd_tbl = nullptr; d.inv_tbl = nullptr;
  d.ovf_key = T.ovf_key; d.ovf_cnt = T.ovf_cnt; d.ovf_mask = T.ovf_mask; d.counters = T.counters; d.max_probe = T.max_probe;
  d.bloom.data = nullptr; d.dirty = nullptr;
  return d;
}

__device__ inline uint64_t hash_tables_n256(const uint64_t* tbl, const K256& key, uint32_t nbytes) {
  uint64_t pos = 0;
  for(uint32_t b = 0; b < nbytes; ++b) pos ^= tbl[b * 256 + ((key.w[b >> 3] >> (8 * (b & 7))) & 0xFF)];
  return pos;
}

struct NSlot { uint64_t lo[3]; uint64_t hi_low; };

__device__ inline NSlot nword_words(const NGeom& N, const K256& key, uint32_t idx0) {
  const K256 tag = k256_or(k256_shl(k256_from64(idx0), N.g.rem_bits), k256_shr(key, N.g.lsize_g));
  NSlot s;
  for(uint32_t i = 0; i < 3; ++i) s.lo[i] = (k256_chunk63(tag, i) << 1) | 1ull;
  s.hi_low = N.g.occ_bit | k256_shr(tag, kNLoBits).w[0];
  return s;
}

__device__ inline K256 nword_slot_key(const NTable& T, const uint64_t* inv_tbl, const uint64_t* slot, uint64_t tile_base) {
  const NGeom& N = T.N;
  K256 tag = k256_shl(k256_from64(slot[3] & (N.g.occ_bit - 1)), kNLoBits);
  for(uint32_t i = 0; i < 3; ++i) tag = k256_or(tag, k256_shl(k256_from64(slot[i] >> 1), 63 * i));
  const K256 rem = k256_and(tag, k256_low_mask(N.g.rem_bits));
  const uint64_t idx0 = k256_shr(tag, N.g.rem_bits).w[0];
  const K256 hi_part = k256_shl(rem, N.g.lsize_g);
  K256 v = hi_part; v.w[0] |= tile_base | idx0;
  K256 key = hi_part; key.w[0] |= hash_tables_n256(inv_tbl, v, N.g.nbytes);
  return key;
}

// claim-or-increment.  Returns true when the key was new (this lane set the last word).
template <bool RETURNING>
__device__ inline bool nword_add(const NTable& T, const K256& key, uint64_t cnt) {
  const TableGeom& g = T.N.g;
  const uint64_t pos = hash_tables_n256(T.fwd_tbl, key, g.nbytes);
  const SlotAddr a = slot_addr(g, pos);
  const NSlot w = nword_words(T.N, key, a.idx0);
  const uint64_t add = cnt << (g.tag_bits + 1);
  for(uint32_t p = 0; p <= T.max_probe; ++p) {
    const uint64_t slot = a.tile_base + probe_slot(a.idx0, p, (uint32_t)g.tile_mask);
    unsigned long long* sp = (unsigned long long*)&T.slots[4 * slot];
    const unsigned long long old = atomicCAS(sp + 3, 0ull, (unsigned long long)w.hi_low);
    if(old != 0ull && (old & g.low_mask) != w.hi_low) continue;
    bool mine = true, set_last = false;
    for(uint32_t i = 0; i < 3 && mine; ++i) {
      const unsigned long long l = atomicCAS(sp + i, 0ull, (unsigned long long)w.lo[i]);
      if(l != 0ull && l != w.lo[i]) mine = false;
      else if(i == 2) set_last = l == 0ull;
    }
    if(!mine) continue;
    if(add) {
      if(RETURNING) {
        const unsigned long long prev = atomicAdd(sp + 3, (unsigned long long)add);
        if((prev >> (g.tag_bits + 1)) + cnt > g.cnt_max) { const DevTable d = ovf_view(T); ovf_add(d, slot, 1); }
      } else {
        __hip_atomic_fetch_add(sp + 3, (unsigned long long)add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return set_last;
  }
  atomicAdd((unsigned long long*)&T.counters[CTR_FULL], 1ull);
  return false;
}

// Slot holding `key`, or ~0 when it is absent (a look-up: the first never-claimed slot ends the search).
__device__ inline uint64_t nword_find(const NTable& T, const K256& key) {
  const TableGeom& g = T.N.g;
  const uint64_t pos = hash_tables_n256(T.fwd_tbl, key, g.nbytes);
  const SlotAddr a = slot_addr(g, pos);
  const NSlot w = nword_words(T.N, key, a.idx0);
  for(uint32_t p = 0; p <= T.max_probe; ++p) {
    const uint64_t slot = a.tile_base + probe_slot(a.idx0, p, (uint32_t)g.tile_mask);
    const uint64_t* sp = &T.slots[4 * slot];
    const uint64_t hi = __hip_atomic_load(sp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if(hi == 0) return ~0ull;
    if((hi & g.low_mask) != w.hi_low) continue;
    if(__hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == w.lo[0] &&
       __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == w.lo[1] &&
       __hip_atomic_load(sp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == w.lo[2]) return slot;
  }
  return ~0ull;
}

__device__ inline void nword_credit(const NTable& T, uint64_t slot, uint64_t cnt) {    // add to an existing slot
  const TableGeom& g = T.N.g;
  const unsigned long long prev = atomicAdd((unsigned long long*)&T.slots[4 * slot + 3], (unsigned long long)(cnt << (g.tag_bits + 1)));
  if((prev >> (g.tag_bits + 1)) + cnt > g.cnt_max) { const DevTable d = ovf_view(T); ovf_add(d, slot, 1); }
}

// hash_counter::add(key, val) with any 64-bit val
__device__ inline bool nword_add_val(const NTable& T, const K256& key, uint64_t val) {
  const TableGeom& g = T.N.g;
  const uint64_t lowpart = val & g.cnt_max, units = val >> g.cnt_bits;
  const bool is_new = nword_add<true>(T, key, lowpart);
  if(units) { const uint64_t s = nword_find(T, key); if(s != ~0ull) { const DevTable d = ovf_view(T); ovf_add(d, s, units); } }
  return is_new;
}

__device__ inline uint64_t nword_count_at(const NTable& T, const DevTable& d, uint64_t slot, uint64_t hi, int have_ovf) {
  uint64_t c = slot_count(T.N.g, hi);
  if(have_ovf) c += ovf_get(d, slot) << T.N.g.cnt_bits;
  return c;
}
__device__ inline bool nword_complete(const uint64_t* sp) { return sp[3] != 0 && sp[0] != 0 && sp[1] != 0 && sp[2] != 0; }

// ---- sequence -> 256-bit k-mers -----------------------------------------------------------------------------
// Halo: k - 1 <= 127 bases = 8 code words before the lane's own 16.  Validity is tracked the way mer_iterator does
// (`filled`): the number of consecutive valid bases ending at the current position, capped at k.
template <bool RETURNING>
__global__ __launch_bounds__(kBlock) void count_ascii_nword_kernel(NTable T, const uint8_t* __restrict__ base, int64_t lo, int64_t hi, int op) {
  __shared__ uint32_t s_codes[kBlock + 8];
  __shared__ uint32_t s_inv[kBlock + 8];
  __shared__ int s_abort;
  const NGeom& N = T.N;
  const uint32_t k = N.g.k, rc_pos = 2 * (k - 1);
  const int tid = threadIdx.x;
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  uint32_t my_mers = 0;
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    if(tid == 0) s_abort = __hip_atomic_load(&T.counters[CTR_FULL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    __syncthreads();
    if(s_abort) break;
    const int64_t tile_start = tile * kTilePos;
    uint32_t c, v;
    load_pack16(base, tile_start + 16 * tid, lo, hi, c, v);
    s_codes[tid + 8] = c; s_inv[tid + 8] = v;
    if(tid < 8) { uint32_t hc, hv; load_pack16(base, tile_start - 128 + 16 * tid, lo, hi, hc, hv); s_codes[tid] = hc; s_inv[tid] = hv; }
    __syncthreads();
    // the k-mer ending just before this lane (garbage where bases were invalid: `filled` guards it)
    K256 fw;
    for(int i = 0; i < 4; ++i) fw.w[i] = ((uint64_t)s_codes[tid + 7 - 2 * i - 1] << 32) | s_codes[tid + 7 - 2 * i];
    fw = k256_and(fw, N.key_mask);
    K256 rc = revcomp256(fw, k);
    uint32_t filled = 0;
    for(int q = 7; q >= 0; --q) {                          // nearest halo word first
      const uint32_t iv = s_inv[tid + q] & 0xFFFFu;
      if(iv == 0) { filled += 16; continue; }
      filled += (uint32_t)__ffs((int)iv) - 1;              // valid bases after the word's last invalid one
      break;
    }
    if(filled > k) filled = k;
    K256 prev = k256_zero(); uint32_t run = 0;
    auto apply = [&](const K256& key, uint32_t n) {
      if(op == 0) nword_add<RETURNING>(T, key, n);
      else if(op == 1) nword_add<RETURNING>(T, key, 0);
      else { const uint64_t s = nword_find(T, key); if(s != ~0ull) nword_credit(T, s, n); }
    };
#pragma unroll 1
    for(int j = 0; j < kPerLane; ++j) {
      const uint64_t code = (c >> (2 * (15 - j))) & 3u;
      k256_roll_fw(fw, code, N.key_mask);
      k256_roll_rc(rc, 3ull - code, rc_pos);
      if((v >> (15 - j)) & 1u) { filled = 0; continue; }
      if(filled < k) ++filled;
      if(filled < k) continue;
      ++my_mers;
      const K256 key = (N.g.canonical && k256_less(rc, fw)) ? rc : fw;
      if(run && k256_eq(key, prev)) { ++run; continue; }
      if(run) apply(prev, run);
      prev = key; run = 1;
    }
    if(run) apply(prev, run);
  }
  uint64_t w = my_mers;
  for(int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
  if((threadIdx.x & 63) == 0 && w) atomicAdd((unsigned long long*)&T.counters[CTR_MERS], (unsigned long long)w);
}

// keys come as ceil(2k / 64) words each (3 for k <= 96), the reference's mer_dna::data() layout
__device__ inline K256 load_key4(const uint64_t* keys, uint64_t i, uint32_t kw, const K256& mask) {
  K256 r = k256_zero();
  for(uint32_t q = 0; q < kw; ++q) r.w[q] = keys[(uint64_t)kw * i + q];
  return k256_and(r, mask);
}

__global__ __launch_bounds__(kBlock) void add_keys_nword_kernel(NTable T, const uint64_t* __restrict__ keys, uint64_t n, uint32_t kw, uint64_t val,
                                                                uint8_t* __restrict__ is_new) {
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const bool nw = nword_add_val(T, load_key4(keys, i, kw, T.N.key_mask), val);
    if(is_new) is_new[i] = nw ? 1 : 0;
  }
}

__global__ __launch_bounds__(kBlock) void lookup_nword_kernel(NTable T, const uint64_t* __restrict__ keys, uint64_t n, uint32_t kw,
                                                              uint64_t* __restrict__ vals, uint8_t* __restrict__ found, int have_ovf) {
  const DevTable d = ovf_view(T);
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t s = nword_find(T, load_key4(keys, i, kw, T.N.key_mask));
    vals[i] = s == ~0ull ? 0 : nword_count_at(T, d, s, T.slots[4 * s + 3], have_ovf);
    if(found) found[i] = s != ~0ull;
  }
}

// hash_counter::double_size: every complete slot re-inserted with its full count into the doubled table
__global__ __launch_bounds__(kBlock) void rehash_nword_kernel(NTable old, NTable neu, int have_ovf) {
  const DevTable od = ovf_view(old);
  const uint64_t n = 1ull << old.N.g.lsize_l;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t* sp = &old.slots[4 * i];
    if(!nword_complete(sp)) continue;
    const K256 key = nword_slot_key(old, old.inv_tbl, sp, i & ~old.N.g.tile_mask);
    nword_add_val(neu, key, nword_count_at(old, od, i, sp[3], have_ovf));
  }
}

// what: 0 stats (out[0..3] = unique, distinct, total, max), 1 histo, 2 per-tile record counts, 3 content digest
__global__ __launch_bounds__(kBlock) void scan_nword_kernel(NTable T, int what, uint64_t lower, uint64_t upper, int have_ovf,
                                                            uint64_t hbase, uint64_t hceil, uint64_t hinc, uint64_t nb,
                                                            unsigned long long* __restrict__ out, uint32_t* __restrict__ tile_counts) {
  const TableGeom& g = T.N.g;
  const DevTable d = ovf_view(T);
  const uint64_t n = 1ull << g.lsize_l;
  uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t* sp = &T.slots[4 * i];
    if(!nword_complete(sp)) continue;
    const uint64_t c = nword_count_at(T, d, i, sp[3], have_ovf);
    if(what == 1) {
      uint64_t b;
      if(c < hbase) b = 0; else if(c > hceil) b = nb - 1; else b = (c - hbase) / hinc;
      atomicAdd(&out[b], 1ull);
      continue;
    }
    if(c < lower || c > upper) continue;
    if(what == 2) { atomicAdd(&tile_counts[i >> g.tile_bits], 1u); continue; }
    if(what == 3) {
      const K256 key = nword_slot_key(T, T.inv_tbl, sp, i & ~g.tile_mask);
      uint64_t h = kDigestSeed;
      for(uint32_t q = 0; q < (g.k + 31) / 32; ++q) h = digest_mix(h ^ key.w[q]);
      h = digest_mix(h ^ c);
      ++a0; a1 += c; a2 += h; a3 ^= h;
      continue;
    }
    a0 += (c == 1); ++a1; a2 += c; a3 = c > a3 ? c : a3;
  }
  if(what == 3) { digest_reduce(a0, a1, a2, a3, out); return; }
  if(what == 0) {
    for(int o = 32; o > 0; o >>= 1) {
      a0 += __shfl_down(a0, o, 64); a1 += __shfl_down(a1, o, 64); a2 += __shfl_down(a2, o, 64);
      const uint64_t m2 = __shfl_down(a3, o, 64); a3 = m2 > a3 ? m2 : a3;
    }
    if((threadIdx.x & 63) == 0) {
      if(a0) atomicAdd(&out[0], (unsigned long long)a0);
      if(a1) atomicAdd(&out[1], (unsigned long long)a1);
      if(a2) atomicAdd(&out[2], (unsigned long long)a2);
      if(a3) atomicMax(&out[3], (unsigned long long)a3);
    }
  }
}

// Sorted dump: one block per tile, bitonic sort in LDS on the tag (hi tag bits, lo2, lo1, lo0) == (pos, key) order
// (mer_heap.hpp:26-30), keys rebuilt through the inverse tables, records as binary_dumper.hpp:36-40 lays them out.
__global__ __launch_bounds__(kBlock) void dump_tiles_nword_kernel(NTable T, uint64_t lower, uint64_t upper, int have_ovf,
                                                                  uint64_t tile0, uint64_t n_tiles, const uint64_t* __restrict__ tile_offsets,
                                                                  uint8_t* __restrict__ out, uint32_t key_bytes, uint32_t val_bytes) {
  JF_DYN_LDS(s_raw);
  const TableGeom& g = T.N.g;
  const uint32_t tsz = 1u << g.tile_bits;
  uint64_t* s_w = reinterpret_cast<uint64_t*>(s_raw);                    // [4][tsz]: word q of entry i at s_w[q * tsz + i]
  uint16_t* s_idx = reinterpret_cast<uint16_t*>(s_w + 4 * (size_t)tsz);
  const DevTable d = ovf_view(T);
  const uint64_t tagmask = g.occ_bit - 1, SENT = ~g.occ_bit;     // (a stored hi word has the occupied bit: a saturated count field over an all-ones tag is not the sentinel)
  const uint64_t maxval = val_bytes >= 8 ? ~0ull : ((1ull << (8 * val_bytes)) - 1);
  const uint32_t rec = key_bytes + val_bytes;
  for(uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint64_t tb = (tile0 + t) << g.tile_bits;
    __syncthreads();
    for(uint32_t i = threadIdx.x; i < tsz; i += blockDim.x) {
      const uint64_t* sp = &T.slots[4 * (tb + i)];
      uint64_t kh = SENT;
      if(nword_complete(sp)) {
        const uint64_t c = nword_count_at(T, d, tb + i, sp[3], have_ovf);
        if(c >= lower && c <= upper) kh = sp[3];
      }
      s_w[3 * tsz + i] = kh; s_w[i] = sp[0]; s_w[tsz + i] = sp[1]; s_w[2 * tsz + i] = sp[2]; s_idx[i] = (uint16_t)i;
    }
    __syncthreads();
    for(uint32_t size = 2; size <= tsz; size <<= 1)
      for(uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
        for(uint32_t i = threadIdx.x; i < tsz / 2; i += blockDim.x) {
          const uint32_t l = ((i & ~(stride - 1)) << 1) | (i & (stride - 1)), h = l | stride;
          const bool up = (l & size) == 0;
          const uint64_t ah = s_w[3 * tsz + l], bh = s_w[3 * tsz + h];
          const uint64_t ka = ah == SENT ? SENT : (ah & tagmask), kb = bh == SENT ? SENT : (bh & tagmask);
          bool gt = ka > kb;
          if(ka == kb) {
            gt = false;
            for(int q = 2; q >= 0; --q) { const uint64_t x = s_w[q * tsz + l], y = s_w[q * tsz + h]; if(x != y) { gt = x > y; break; } }
          }
          if(gt == up) {
            for(int q = 0; q < 4; ++q) { const uint64_t x = s_w[q * tsz + l]; s_w[q * tsz + l] = s_w[q * tsz + h]; s_w[q * tsz + h] = x; }
            const uint16_t ia = s_idx[l]; s_idx[l] = s_idx[h]; s_idx[h] = ia;
          }
        }
        __syncthreads();
      }
    uint8_t* dst0 = out + tile_offsets[t] * rec;
    for(uint32_t i = threadIdx.x; i < tsz; i += blockDim.x) {
      const uint64_t hi = s_w[3 * tsz + i];
      if(hi == SENT) continue;
      const uint64_t sl[4] = {s_w[i], s_w[tsz + i], s_w[2 * tsz + i], hi};
      const K256 key = nword_slot_key(T, T.inv_tbl, sl, tb);
      uint64_t cnt = nword_count_at(T, d, tb + s_idx[i], hi, have_ovf);
      if(cnt > maxval) cnt = maxval;
      uint8_t* dd = dst0 + (uint64_t)i * rec;
      for(uint32_t b = 0; b < key_bytes; ++b) dd[b] = (uint8_t)(key.w[b >> 3] >> (8 * (b & 7)));
      for(uint32_t b = 0; b < val_bytes; ++b) dd[key_bytes + b] = (uint8_t)(cnt >> (8 * b));
    }
  }
}

}  // namespace jfgpu
