// jellyfish_amd/csrc/kernels_wide.hip.hpp -- two-word keys, 33 <= k <= 64 (BASELINE config 5).
//
// The reference stores a k-mer as ceil(k/32) 64-bit words (include/jellyfish/mer_dna.hpp:143-170,
// 712-717; tests/large_key.sh counts k = 100) and claims multi-word keys with a per-word "set" bit
// (offsets_key_value.hpp:28-31, large_hash_array.hpp:542-579) because x86 has no wide enough CAS.
// gfx950 has 64-bit atomics only, so the same idea here, with 128-bit slots:
//
//   slot = { lo, hi }   hi = [ count | occ | tag >> 63 ]     lo = [ tag & (2^63 - 1) | valid ]
//
// claim: CAS(hi, 0 -> occ|tag_hi) with count 0, then CAS(lo, 0 -> tag_lo|valid): the first lane to set lo
// defines the slot's key among those sharing tag_hi; nobody ever waits.  Then add the count to hi.  Everything
// else (tile-local triangular probing, quotienting, count at the top of the word, overflow side
// table, (pos, key) dump order) is the one-word design of kernels.hip.hpp.
// Insert path: global atomics only (the partitioned path is one-word for now).
#pragma once
#include "kernels.hip.hpp"
#include "kernels_bloom.hip.hpp"

namespace jfgpu {

typedef unsigned __int128 u128;

struct WideGeom {
  TableGeom g;          // tag_bits/occ_bit/low_mask/inc/cnt_* describe the HI word; rem_bits may exceed 64
  uint32_t tag_full;    // tile_bits + rem_bits: bits of the whole tag
  uint32_t pad_[3];
  u128 key_mask;
};

// shard_bits / shard_id: one shard of a table spread over GPUs (the shard owns the global positions whose top shard_bits
// bits are shard_id, like the one-word geometry: kmer_core.hpp).
inline bool wide_geom_init(WideGeom& W, uint32_t k, uint32_t lsize_g, uint32_t canonical, uint32_t shard_bits = 0, uint32_t shard_id = 0) {
  TableGeom& g = W.g;
  if(k < 33 || k > 64 || lsize_g > 63 || shard_bits > lsize_g || lsize_g - shard_bits < kMaxTileBits) return false;
  g.k = k; g.key_bits = 2 * k; g.lsize_g = lsize_g; g.lsize_l = lsize_g - shard_bits; g.shard_bits = shard_bits; g.shard_id = shard_id;
  g.tile_bits = kMaxTileBits; g.slot32 = 0; g.hash_xs = 0;
  g.rem_bits = g.key_bits - lsize_g;
  W.tag_full = g.tile_bits + g.rem_bits;
  const uint32_t th = W.tag_full > 63 ? W.tag_full - 63 : 0;     // tag bits kept in the hi word
  if(th + 1 + kMinCountBits > 64) return false;
  g.tag_bits = th;
  g.cnt_bits = 63 - th;
  g.nbytes = (g.key_bits + 7) / 8;
  g.canonical = canonical;
  g.key_mask = ~0ull;
  W.key_mask = g.key_bits == 128 ? ~(u128)0 : (((u128)1 << g.key_bits) - 1);
  g.tile_mask = (1ull << g.tile_bits) - 1;
  g.rem_mask = 0;                                                // unused (rem is wider than a word)
  g.local_mask = (1ull << g.lsize_l) - 1;
  g.occ_bit = 1ull << th;
  g.low_mask = (g.occ_bit << 1) - 1;
  g.inc = g.occ_bit << 1;
  g.cnt_max = (1ull << g.cnt_bits) - 1;
  return true;
}
inline uint32_t wide_min_lsize(uint32_t k) {
  int need = (int)(2 * k + kMaxTileBits) - 63 - (int)(63 - kMinCountBits);   // tag_full - 63 <= 47
  if(need < (int)kMaxTileBits) need = kMaxTileBits;
  return (uint32_t)need;
}

struct WideTable {
  WideGeom W;
  uint64_t* slots;            // [2 << lsize]: slot s = { slots[2s] = lo, slots[2s+1] = hi }
  const uint64_t* fwd_tbl;    // [nbytes * 256]
  const uint64_t* inv_tbl;
  uint64_t* ovf_key; uint64_t* ovf_cnt; uint64_t ovf_mask;
  uint64_t* counters;
  uint32_t max_probe;
  DevBloom bloom;             // count --bc filter (data == nullptr: none)
  uint8_t* dirty;             // one byte per tile: something was ever inserted (the LDS tile insert skips reading clean tiles)
};

__device__ inline DevTable ovf_view(const WideTable& T) {     // reuse ovf_add / ovf_get of the one-word code
  DevTable d; d.g = T.W.g; d.slots = nullptr; d.fwd_tbl = nullptr; d.inv_tbl = nullptr;
  d.ovf_key = T.ovf_key; d.ovf_cnt = T.ovf_cnt; d.ovf_mask = T.ovf_mask; d.counters = T.counters; d.max_probe = T.max_probe;
  d.bloom.data = nullptr; d.dirty = nullptr;
  return d;
}

__device__ inline u128 revcomp128(u128 x, uint32_t k) {
  // reverse the 64 2-bit groups and complement: both halves through the 64-bit routine, swapped
  const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  const u128 r = ((u128)revcomp64(lo, 32) << 64) | revcomp64(hi, 32);
  return r >> (128 - 2 * k);
}

__device__ inline uint64_t hash_tables_wide(const uint64_t* tbl, u128 key, uint32_t nbytes) {
  uint64_t pos = 0;
  const uint64_t lo = (uint64_t)key, hi = (uint64_t)(key >> 64);
#pragma unroll
  for(uint32_t b = 0; b < 8; ++b) pos ^= tbl[b * 256 + ((lo >> (8 * b)) & 0xFF)];
  for(uint32_t b = 8; b < nbytes; ++b) pos ^= tbl[b * 256 + ((hi >> (8 * (b - 8))) & 0xFF)];
  return pos;
}

struct WideSlot { uint64_t lo, hi_low; };   // lo word (with valid bit) and occ|tag_hi of a key at a position

__device__ inline WideSlot wide_words(const WideGeom& W, u128 key, uint32_t idx0) {
  const u128 tag = ((u128)idx0 << W.g.rem_bits) | (key >> W.g.lsize_g);
  WideSlot s;
  s.lo = (((uint64_t)tag) << 1) | 1ull;                       // low 63 tag bits + valid
  s.hi_low = W.g.occ_bit | (uint64_t)(tag >> 63);
  return s;
}

__device__ inline u128 wide_slot_key(const WideTable& T, const uint64_t* inv_tbl, uint64_t lo, uint64_t hi, uint64_t tile_base) {
  const WideGeom& W = T.W;
  const u128 tag = ((u128)(hi & (W.g.occ_bit - 1)) << 63) | (lo >> 1);
  const u128 rem = tag & ((((u128)1) << W.g.rem_bits) - 1);
  const uint64_t idx0 = (uint64_t)(tag >> W.g.rem_bits);
  const uint64_t pos = ((uint64_t)W.g.shard_id << W.g.lsize_l) | tile_base | idx0;
  const u128 v = (rem << W.g.lsize_g) | pos;
  const uint64_t low_bits = hash_tables_wide(inv_tbl, v, W.g.nbytes);
  return (rem << W.g.lsize_g) | low_bits;
}

// claim-or-increment on a 128-bit slot.  Returns true when the key was new.
template <bool RETURNING>
__device__ inline bool wide_add(const WideTable& T, const uint64_t* fwd_lds, u128 key, uint64_t cnt) {
  const TableGeom& g = T.W.g;
  const uint64_t pos = hash_tables_wide(fwd_lds, key, g.nbytes);
  const SlotAddr a = slot_addr(g, pos);
  if(a.shard != g.shard_id) { atomicAdd((unsigned long long*)&T.counters[CTR_MISROUTED], 1ull); return false; }      // not ours: never silently inserted
  const WideSlot w = wide_words(T.W, key, a.idx0);
  const uint64_t add = cnt << (g.tag_bits + 1);
  const uint32_t tmask = (uint32_t)g.tile_mask;
  if(T.dirty) { uint8_t* d = &T.dirty[a.tile_base >> g.tile_bits]; if(!*d) *d = 1; }
  for(uint32_t p = 0; p <= T.max_probe; ++p) {
    const uint64_t slot = a.tile_base + probe_slot(a.idx0, p, tmask);
    unsigned long long* hi = (unsigned long long*)&T.slots[2 * slot + 1];
    unsigned long long* lo = (unsigned long long*)&T.slots[2 * slot];
    // 1. make sure the hi word carries occ | tag_hi (count 0 if we are first)
    const unsigned long long old = atomicCAS(hi, 0ull, (unsigned long long)w.hi_low);
    if(old != 0ull && (old & g.low_mask) != w.hi_low) continue;               // another tag_hi lives here
    // 2. whoever sets lo first defines which key (among those sharing tag_hi) owns the slot.  No lane
    //    ever waits for another one: a lane that claimed hi but loses lo simply probes on, leaving a
    //    consistent (tag_hi, lo) pair behind.  (A wait-for-valid-bit protocol deadlocks lanes of one
    //    wave against each other under SIMT.)
    const unsigned long long l = atomicCAS(lo, 0ull, (unsigned long long)w.lo);
    if(l != 0ull && l != w.lo) continue;                                        // same tag_hi, different key
    if(add) {
      if(RETURNING) {
        const unsigned long long prev = atomicAdd(hi, (unsigned long long)add);
        if((prev >> (g.tag_bits + 1)) + cnt > g.cnt_max) { const DevTable d = ovf_view(T); ovf_add(d, slot, 1); }
      } else {
        __hip_atomic_fetch_add(hi, (unsigned long long)add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return l == 0ull;                                                           // new key iff we set lo
  }
  atomicAdd((unsigned long long*)&T.counters[CTR_FULL], 1ull);
  return false;
}

// update_add on a 128-bit slot: increment only if the key is there (the UPDATE pass of `count --if`).
template <bool RETURNING>
__device__ inline bool wide_update_add(const WideTable& T, const uint64_t* fwd_lds, u128 key, uint64_t cnt) {
  const TableGeom& g = T.W.g;
  const uint64_t pos = hash_tables_wide(fwd_lds, key, g.nbytes);
  const SlotAddr a = slot_addr(g, pos);
  const WideSlot w = wide_words(T.W, key, a.idx0);
  const uint64_t add = cnt << (g.tag_bits + 1);
  const uint32_t tmask = (uint32_t)g.tile_mask;
  for(uint32_t p = 0; p <= T.max_probe; ++p) {
    const uint64_t slot = a.tile_base + probe_slot(a.idx0, p, tmask);
    const uint64_t hi = __hip_atomic_load(&T.slots[2 * slot + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if(hi == 0) return false;
    if((hi & g.low_mask) != w.hi_low) continue;
    if(__hip_atomic_load(&T.slots[2 * slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != w.lo) continue;
    unsigned long long* hp = (unsigned long long*)&T.slots[2 * slot + 1];
    if(RETURNING) {
      const unsigned long long prev = atomicAdd(hp, (unsigned long long)add);
      if((prev >> (g.tag_bits + 1)) + cnt > g.cnt_max) { const DevTable d = ovf_view(T); ovf_add(d, slot, 1); }
    } else {
      __hip_atomic_fetch_add(hp, (unsigned long long)add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
  }
  return false;
}

// hash_counter::add(key, val) with an arbitrary 64-bit val: low part in the slot, the rest in the side table
__device__ inline bool wide_add_val(const WideTable& T, const uint64_t* fwd_lds, u128 key, uint64_t val) {
  const TableGeom& g = T.W.g;
  const uint64_t lowpart = val & g.cnt_max, units = val >> g.cnt_bits;
  const bool is_new = wide_add<true>(T, fwd_lds, key, lowpart);
  if(units) {   // find the slot again (cheap: rare) and credit the overflow units
    const uint64_t pos = hash_tables_wide(fwd_lds, key, g.nbytes);
    const SlotAddr a = slot_addr(g, pos);
    const WideSlot w = wide_words(T.W, key, a.idx0);
    for(uint32_t p = 0; p <= T.max_probe; ++p) {
      const uint64_t slot = a.tile_base + probe_slot(a.idx0, p, (uint32_t)g.tile_mask);
      const uint64_t hi = __hip_atomic_load(&T.slots[2 * slot + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if(hi == 0) break;
      if((hi & g.low_mask) == w.hi_low && __hip_atomic_load(&T.slots[2 * slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == w.lo) {
        const DevTable d = ovf_view(T); ovf_add(d, slot, units); break;
      }
    }
  }
  return is_new;
}

// ---- sequence -> 128-bit k-mers ----------------------------------------------------------------
struct LaneWordsW { uint32_t cur, p1, p2, p3, p4; u128 inv80; };

__device__ inline LaneWordsW stage_tile_wide(const uint8_t* __restrict__ base, int64_t tile_start, int64_t lo, int64_t hi,
                                             uint32_t* s_codes, uint32_t* s_inv) {
  const int tid = threadIdx.x;
  uint32_t c, v;
  load_pack16(base, tile_start + 16 * tid, lo, hi, c, v);
  s_codes[tid + 4] = c; s_inv[tid + 4] = v;
  if(tid < 4) {
    uint32_t hc, hv;
    load_pack16(base, tile_start - 64 + 16 * tid, lo, hi, hc, hv);
    s_codes[tid] = hc; s_inv[tid] = hv;
  }
  __syncthreads();
  LaneWordsW L;
  L.cur = c; L.p1 = s_codes[tid + 3]; L.p2 = s_codes[tid + 2]; L.p3 = s_codes[tid + 1]; L.p4 = s_codes[tid];
  L.inv80 = ((u128)s_inv[tid] << 64) | ((u128)s_inv[tid + 1] << 48) | ((u128)s_inv[tid + 2] << 32) | ((u128)s_inv[tid + 3] << 16) | v;
  return L;
}

template <typename F>
__device__ inline void for_each_kmer_wide(const WideGeom& W, const LaneWordsW& L, F&& f) {
  const uint32_t k = W.g.k;
  u128 fw = ((((u128)L.p4 << 96) | ((u128)L.p3 << 64) | ((u128)L.p2 << 32) | L.p1)) & W.key_mask;
  u128 rc = revcomp128(fw, k);
  const u128 kwin = (((u128)1) << k) - 1;                 // k <= 64
  const uint32_t rc_shift = 2 * (k - 1);
#pragma unroll
  for(int j = 0; j < kPerLane; ++j) {
    const uint64_t c = (L.cur >> (2 * (15 - j))) & 3u;
    fw = ((fw << 2) | c) & W.key_mask;
    rc = (rc >> 2) | ((u128)(3ull - c) << rc_shift);
    const bool valid = ((L.inv80 >> (15 - j)) & kwin) == 0;
    if(valid) f(j, (W.g.canonical && rc < fw) ? rc : fw);
  }
}

// Bloom counter on two-word keys: h0 = M1 * key, h1 = M2 * key with 64 x 2k matrices (mer_dna_bloom_counter.hpp:19-34);
// the byte tables (16 x 256 entries each) are read through the caches.
__device__ inline bool bloom_admits_wide(const DevBloom& B, u128 key) {
  const uint64_t h0 = hash_tables_wide(B.tbl1, key, B.nbytes), h1 = hash_tables_wide(B.tbl2, key, B.nbytes);
  return B.kind == 1 ? bloom_filter_insert(B, h0, h1) : bloom_all_two(B, h0, h1);
}

__global__ __launch_bounds__(kBlock) void bloom_insert_ascii_wide_kernel(DevBloom B, WideGeom W, const uint8_t* __restrict__ base,
                                                                         int64_t lo, int64_t hi, unsigned long long* __restrict__ mers) {
  __shared__ uint32_t s_codes[kBlock + 4];
  __shared__ uint32_t s_inv[kBlock + 4];
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  uint32_t my = 0;
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();
    const LaneWordsW L = stage_tile_wide(base, tile * kTilePos, lo, hi, s_codes, s_inv);
    for_each_kmer_wide(W, L, [&](int, u128 key) {
      ++my;
      bloom_insert(B, hash_tables_wide(B.tbl1, key, B.nbytes), hash_tables_wide(B.tbl2, key, B.nbytes));
    });
  }
  uint64_t w = my;
  for(int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
  if((threadIdx.x & 63) == 0 && w) atomicAdd(mers, (unsigned long long)w);
}

__global__ __launch_bounds__(kBlock) void bloom_keys_wide_kernel(DevBloom B, u128 key_mask, const uint64_t* __restrict__ keys, uint64_t n,
                                                                 uint8_t* __restrict__ out, int do_insert) {
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const u128 key = ((((u128)keys[2 * i + 1]) << 64) | keys[2 * i]) & key_mask;
    const uint64_t h0 = hash_tables_wide(B.tbl1, key, B.nbytes), h1 = hash_tables_wide(B.tbl2, key, B.nbytes);
    const uint32_t r = do_insert ? bloom_insert(B, h0, h1) : bloom_check(B, h0, h1);
    if(out) out[i] = (uint8_t)r;
  }
}

template <bool RETURNING>
__global__ __launch_bounds__(kBlock) void count_ascii_wide_kernel(WideTable T, const uint8_t* __restrict__ base, int64_t lo, int64_t hi, int op) {
  __shared__ uint64_t s_fwd[16 * 256];
  __shared__ uint32_t s_codes[kBlock + 4];
  __shared__ uint32_t s_inv[kBlock + 4];
  __shared__ int s_abort;
  load_tables_lds(s_fwd, T.fwd_tbl, T.W.g.nbytes);
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  uint32_t my_mers = 0;
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    if(threadIdx.x == 0) s_abort = __hip_atomic_load(&T.counters[CTR_FULL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    __syncthreads();
    if(s_abort) break;
    const LaneWordsW L = stage_tile_wide(base, tile * kTilePos, lo, hi, s_codes, s_inv);
    u128 prev = 0; uint32_t run = 0;
    auto apply = [&](u128 key, uint32_t n) {               // op: 0 add, 1 set (prime), 2 update_add (count_main.cc:152-184)
      if(op == 0) wide_add<RETURNING>(T, s_fwd, key, n);
      else if(op == 1) wide_add<RETURNING>(T, s_fwd, key, 0);
      else wide_update_add<RETURNING>(T, s_fwd, key, n);
    };
    const bool filtered = T.bloom.data != nullptr;          // count --bc (count_main.cc:115-118)
    for_each_kmer_wide(T.W, L, [&](int, u128 key) {
      ++my_mers;
      if(filtered && !bloom_admits_wide(T.bloom, key)) return;
      if(run && key == prev) { ++run; return; }
      if(run) apply(prev, run);
      prev = key; run = 1;
    });
    if(run) apply(prev, run);
  }
  uint64_t w = my_mers;
  for(int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
  if((threadIdx.x & 63) == 0 && w) atomicAdd((unsigned long long*)&T.counters[CTR_MERS], (unsigned long long)w);
}

__device__ inline u128 load_key2(const uint64_t* keys, uint64_t i, u128 mask) {
  return (((u128)keys[2 * i + 1] << 64) | keys[2 * i]) & mask;
}

__global__ __launch_bounds__(kBlock) void add_keys_wide_kernel(WideTable T, const uint64_t* __restrict__ keys, uint64_t n, uint64_t val,
                                                               uint8_t* __restrict__ is_new) {
  __shared__ uint64_t s_fwd[16 * 256];
  load_tables_lds(s_fwd, T.fwd_tbl, T.W.g.nbytes);
  __syncthreads();
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const bool nw = wide_add_val(T, s_fwd, load_key2(keys, i, T.W.key_mask), val);
    if(is_new) is_new[i] = nw ? 1 : 0;
  }
}

// hash_counter::update_add on encoded two-word keys, val == 1: the receive side of the exchange in the UPDATE pass of
// count --if over shards (count_main.cc:152-184 with --gpus; the one-word twin is update_keys_one_kernel)
template <bool RETURNING>
__global__ __launch_bounds__(kBlock) void update_keys_wide_kernel(WideTable T, const uint64_t* __restrict__ keys, uint64_t n) {
  __shared__ uint64_t s_fwd[16 * 256];
  load_tables_lds(s_fwd, T.fwd_tbl, T.W.g.nbytes);
  __syncthreads();
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    wide_update_add<RETURNING>(T, s_fwd, load_key2(keys, i, T.W.key_mask), 1);
}

__global__ __launch_bounds__(kBlock) void lookup_wide_kernel(WideTable T, const uint64_t* __restrict__ keys, uint64_t n,
                                                             uint64_t* __restrict__ vals, uint8_t* __restrict__ found, int have_ovf) {
  __shared__ uint64_t s_fwd[16 * 256];
  load_tables_lds(s_fwd, T.fwd_tbl, T.W.g.nbytes);
  __syncthreads();
  const TableGeom& g = T.W.g;
  const DevTable d = ovf_view(T);
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const u128 key = load_key2(keys, i, T.W.key_mask);
    const uint64_t pos = hash_tables_wide(s_fwd, key, g.nbytes);
    const SlotAddr a = slot_addr(g, pos);
    const WideSlot w = wide_words(T.W, key, a.idx0);
    uint64_t val = 0; uint8_t fnd = 0;
    for(uint32_t p = 0; p <= T.max_probe; ++p) {
      const uint64_t slot = a.tile_base + probe_slot(a.idx0, p, (uint32_t)g.tile_mask);
      const uint64_t hi = T.slots[2 * slot + 1];
      if(hi == 0) break;
      if((hi & g.low_mask) == w.hi_low && T.slots[2 * slot] == w.lo) {
        val = slot_count(g, hi);
        if(have_ovf) val += ovf_get(d, slot) << g.cnt_bits;
        fnd = 1; break;
      }
    }
    vals[i] = val;
    if(found) found[i] = fnd;
  }
}

// hash_counter::double_size for two-word keys: every occupied slot is decoded and re-inserted with its full
// count into the doubled table (new matrix, one more row).  Hash tables are read through the caches.
__global__ __launch_bounds__(kBlock) void rehash_wide_kernel(WideTable old, WideTable neu, int have_ovf) {
  const TableGeom& g = old.W.g;
  const DevTable od = ovf_view(old);
  const uint64_t n = 1ull << g.lsize_l;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t hi = old.slots[2 * i + 1];
    if(!hi) continue;
    const uint64_t lo = old.slots[2 * i];
    if(!lo) continue;                                    // hi claimed, never completed: holds no key
    const u128 key = wide_slot_key(old, old.inv_tbl, lo, hi, i & ~g.tile_mask);
    uint64_t c = slot_count(g, hi);
    if(have_ovf) c += ovf_get(od, i) << g.cnt_bits;
    wide_add_val(neu, neu.fwd_tbl, key, c);
  }
}

// stats / histo / tile_count: only the hi word (count + occupancy) matters -> one strided scan.
// what: 0 stats (out[0..3] = unique, distinct, total, max), 1 histo, 2 per-tile record counts
__global__ __launch_bounds__(kBlock) void scan_wide_kernel(WideTable T, int what, uint64_t lower, uint64_t upper, int have_ovf,
                                                           uint64_t hbase, uint64_t hceil, uint64_t hinc, uint64_t nb,
                                                           unsigned long long* __restrict__ out, uint32_t* __restrict__ tile_counts) {
  const TableGeom& g = T.W.g;
  const DevTable d = ovf_view(T);
  const uint64_t n = 1ull << g.lsize_l;
  uint64_t uniq = 0, dist = 0, tot = 0, mx = 0;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t hi = T.slots[2 * i + 1];
    if(!hi) continue;
    uint64_t c = slot_count(g, hi);
    if(have_ovf) c += ovf_get(d, i) << g.cnt_bits;
    if(what == 1) {
      uint64_t b;
      if(c < hbase) b = 0; else if(c > hceil) b = nb - 1; else b = (c - hbase) / hinc;
      atomicAdd(&out[b], 1ull);
      continue;
    }
    if(c < lower || c > upper) continue;
    if(what == 2) { atomicAdd(&tile_counts[i >> g.tile_bits], 1u); continue; }
    uniq += (c == 1); ++dist; tot += c; mx = c > mx ? c : mx;
  }
  if(what == 0) {
    for(int o = 32; o > 0; o >>= 1) {
      uniq += __shfl_down(uniq, o, 64); dist += __shfl_down(dist, o, 64); tot += __shfl_down(tot, o, 64);
      const uint64_t m2 = __shfl_down(mx, o, 64); mx = m2 > mx ? m2 : mx;
    }
    if((threadIdx.x & 63) == 0) {
      if(uniq) atomicAdd(&out[0], (unsigned long long)uniq);
      if(dist) atomicAdd(&out[1], (unsigned long long)dist);
      if(tot) atomicAdd(&out[2], (unsigned long long)tot);
      if(mx) atomicMax(&out[3], (unsigned long long)mx);
    }
  }
}

// Content digest of a two-word-key table (see digest_kernel in kernels.hip.hpp).
__global__ __launch_bounds__(kBlock) void digest_wide_kernel(WideTable T, uint64_t lower, uint64_t upper, int have_ovf,
                                                             unsigned long long* __restrict__ out) {
  const TableGeom& g = T.W.g;
  const DevTable d = ovf_view(T);
  const uint64_t n = 1ull << g.lsize_l;
  uint64_t cnt = 0, tot = 0, sum = 0, x = 0;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t hi = T.slots[2 * i + 1];
    if(!hi) continue;
    const uint64_t lo = T.slots[2 * i];
    if(!lo) continue;
    uint64_t c = slot_count(g, hi);
    if(have_ovf) c += ovf_get(d, i) << g.cnt_bits;
    if(c < lower || c > upper) continue;
    const u128 key = wide_slot_key(T, T.inv_tbl, lo, hi, i & ~g.tile_mask);
    const uint64_t h = digest_mix(digest_mix(digest_mix(kDigestSeed ^ (uint64_t)key) ^ (uint64_t)(key >> 64)) ^ c);
    ++cnt; tot += c; sum += h; x ^= h;
  }
  digest_reduce(cnt, tot, sum, x, out);
}

// Sorted dump of 128-bit slots: one block per tile, bitonic sort on (tag_hi, tag_lo) in LDS
// (8192 x 16 B = 128 KiB + 16 KiB of slot indices), inverse tables through the caches.
__global__ __launch_bounds__(kBlock) void dump_tiles_wide_kernel(WideTable T, uint64_t lower, uint64_t upper, int have_ovf,
                                                                 uint64_t tile0, uint64_t n_tiles,
                                                                 const uint64_t* __restrict__ tile_offsets,
                                                                 uint8_t* __restrict__ out, uint32_t key_bytes, uint32_t val_bytes) {
  JF_DYN_LDS(s_raw);
  const TableGeom& g = T.W.g;
  const uint32_t tsz = 1u << g.tile_bits;
  uint64_t* s_hi = reinterpret_cast<uint64_t*>(s_raw);
  uint64_t* s_lo = s_hi + tsz;
  uint16_t* s_idx = reinterpret_cast<uint16_t*>(s_lo + tsz);
  const DevTable d = ovf_view(T);
  const uint64_t tagmask = g.occ_bit - 1, SENT = ~g.occ_bit;     // (a stored hi word has the occupied bit: a saturated count field over an all-ones tag is not the sentinel)
  const uint64_t maxval = val_bytes >= 8 ? ~0ull : ((1ull << (8 * val_bytes)) - 1);
  const uint32_t rec = key_bytes + val_bytes;
  for(uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint64_t tb = (tile0 + t) << g.tile_bits;
    __syncthreads();
    for(uint32_t i = threadIdx.x; i < tsz; i += blockDim.x) {
      uint64_t hi = T.slots[2 * (tb + i) + 1], lo = T.slots[2 * (tb + i)];
      uint64_t kh = SENT;
      if(hi) {
        uint64_t c = slot_count(g, hi);
        if(have_ovf) c += ovf_get(d, tb + i) << g.cnt_bits;
        if(c >= lower && c <= upper) kh = hi;
      }
      s_hi[i] = kh; s_lo[i] = lo; s_idx[i] = (uint16_t)i;
    }
    __syncthreads();
    for(uint32_t size = 2; size <= tsz; size <<= 1) {
      for(uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
        for(uint32_t i = threadIdx.x; i < tsz / 2; i += blockDim.x) {
          const uint32_t l = ((i & ~(stride - 1)) << 1) | (i & (stride - 1)), h = l | stride;
          const bool up = (l & size) == 0;
          const uint64_t ah = s_hi[l], bh = s_hi[h], al = s_lo[l], bl = s_lo[h];
          const uint64_t ka = ah == SENT ? SENT : (ah & tagmask), kb = bh == SENT ? SENT : (bh & tagmask);
          const bool gt = ka > kb || (ka == kb && al > bl);
          if(gt == up) {
            s_hi[l] = bh; s_hi[h] = ah; s_lo[l] = bl; s_lo[h] = al;
            const uint16_t ia = s_idx[l]; s_idx[l] = s_idx[h]; s_idx[h] = ia;
          }
        }
        __syncthreads();
      }
    }
    uint8_t* dst0 = out + tile_offsets[t] * rec;
    for(uint32_t i = threadIdx.x; i < tsz; i += blockDim.x) {
      const uint64_t hi = s_hi[i];
      if(hi == SENT) continue;
      const u128 key = wide_slot_key(T, T.inv_tbl, s_lo[i], hi, tb);
      uint64_t cnt = slot_count(g, hi);
      if(have_ovf) cnt += ovf_get(d, tb + s_idx[i]) << g.cnt_bits;
      if(cnt > maxval) cnt = maxval;
      uint8_t* dd = dst0 + (uint64_t)i * rec;
      for(uint32_t b = 0; b < key_bytes; ++b) dd[b] = (uint8_t)(key >> (8 * b));
      for(uint32_t b = 0; b < val_bytes; ++b) dd[key_bytes + b] = (uint8_t)(cnt >> (8 * b));
    }
  }
}

// ---- multi-GPU: a contract buffer's two-word k-mers grouped by owner (abi_comm.inl, key path) -------------------------
// The two passes of kernels.hip.hpp's partition_count / partition_scatter kernels for 128-bit keys: out receives two
// 64-bit words per k-mer (low word first: what add_keys takes), counts and cursors are in k-mers.
// BLOOM: count --bc with --gpus -- the sender asks its copy of the (read-only) Bloom counter, what it does not admit never travels
template <bool BLOOM = false>
__global__ __launch_bounds__(kBlock) void partition_count_wide_kernel(WideTable T, const uint8_t* __restrict__ base, int64_t lo, int64_t hi,
                                                                      unsigned long long* __restrict__ shard_counts) {
  __shared__ uint64_t s_fwd[16 * 256];
  __shared__ uint32_t s_codes[kBlock + 4];
  __shared__ uint32_t s_inv[kBlock + 4];
  __shared__ uint32_t s_hist[256];
  load_tables_lds(s_fwd, T.fwd_tbl, T.W.g.nbytes);
  const uint32_t n_shards = 1u << T.W.g.shard_bits;
  for(uint32_t i = threadIdx.x; i < n_shards; i += blockDim.x) s_hist[i] = 0;
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();
    const LaneWordsW L = stage_tile_wide(base, tile * kTilePos, lo, hi, s_codes, s_inv);
    for_each_kmer_wide(T.W, L, [&](int, u128 key) {
      if(BLOOM && !bloom_admits_wide(T.bloom, key)) return;
      const uint64_t pos = hash_tables_wide(s_fwd, key, T.W.g.nbytes);
      atomicAdd(&s_hist[(uint32_t)(pos >> T.W.g.lsize_l)], 1u);
    });
  }
  __syncthreads();
  for(uint32_t i = threadIdx.x; i < n_shards; i += blockDim.x)
    if(s_hist[i]) atomicAdd(&shard_counts[i], (unsigned long long)s_hist[i]);
}

template <bool BLOOM = false>
__global__ __launch_bounds__(kBlock) void partition_scatter_wide_kernel(WideTable T, const uint8_t* __restrict__ base, int64_t lo, int64_t hi,
                                                                        unsigned long long* __restrict__ cursors, uint64_t* __restrict__ out) {
  __shared__ uint64_t s_fwd[16 * 256];
  __shared__ uint32_t s_codes[kBlock + 4];
  __shared__ uint32_t s_inv[kBlock + 4];
  __shared__ uint32_t s_hist[256];
  __shared__ unsigned long long s_base[256];
  load_tables_lds(s_fwd, T.fwd_tbl, T.W.g.nbytes);
  const uint32_t n_shards = 1u << T.W.g.shard_bits;
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();
    for(uint32_t i = threadIdx.x; i < n_shards; i += blockDim.x) s_hist[i] = 0;
    const LaneWordsW L = stage_tile_wide(base, tile * kTilePos, lo, hi, s_codes, s_inv);     // contains a barrier
    // two sweeps over the lane's windows (the keys are wide: they are not kept, they are rolled again): ranks, then stores
    uint32_t rank[kPerLane]; uint32_t vmask = 0, sh[kPerLane];
    for_each_kmer_wide(T.W, L, [&](int j, u128 key) {
      if(BLOOM && !bloom_admits_wide(T.bloom, key)) return;      // (the same answers as the count pass: the counter is read-only here)
      const uint32_t s = (uint32_t)(hash_tables_wide(s_fwd, key, T.W.g.nbytes) >> T.W.g.lsize_l);
      sh[j] = s; rank[j] = atomicAdd(&s_hist[s], 1u); vmask |= 1u << j;
    });
    __syncthreads();
    for(uint32_t i = threadIdx.x; i < n_shards; i += blockDim.x)
      s_base[i] = s_hist[i] ? atomicAdd(&cursors[i], (unsigned long long)s_hist[i]) : 0ull;
    __syncthreads();
    for_each_kmer_wide(T.W, L, [&](int j, u128 key) {
      if(!((vmask >> j) & 1u)) return;
      const unsigned long long at = s_base[sh[j]] + rank[j];
      out[2 * at] = (uint64_t)key; out[2 * at + 1] = (uint64_t)(key >> 64);
    });
  }
}

}  // namespace jfgpu
