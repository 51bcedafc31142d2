// jellyfish_amd/csrc/kernels.hip.hpp -- gfx950 (CDNA4) device code of the hot path.
//
// What each kernel replaces in the reference (paths relative to /root/reference):
//   count_ascii_kernel   the COUNT loop of mer_counter_base::start
//                        (sub_commands/count_main.cc:152-163): mer_iterator
//                        (mer_iterator.hpp:53-81) + RectangularBinaryMatrix::times
//                        (rectangular_binary_matrix.hpp:155-164) + array_base::add
//                        (large_hash_array.hpp:291-295,509-597,741-752)
//   add_keys_kernel      hash_counter::add on encoded mers (hash_counter.hpp:91-126)
//   lookup_kernel        array_base::get_val_for_key (large_hash_array.hpp:354-372)
//   partition_*_kernel   new: hash-prefix routing of k-mers to their owning GPU
//   stats/histo/dump     region iterators + sorted_dumper + binary_writer
//                        (large_hash_iterator.hpp, sorted_dumper.hpp:57-101,
//                        binary_dumper.hpp:36-40)
//
// Design notes (see DESIGN.md for the numbers):
//   * wave64 everywhere; blocks of 256 threads = 4 waves, one per SIMD.
//   * sequence bytes are read once, 16 B per lane (1 KiB per wave-instruction),
//     converted to 2-bit codes + an invalid mask in registers and exchanged with
//     the neighbouring lanes through LDS (k-1 <= 31 bases of halo = two words).
//   * the GF(2) hash is 2k/8 LDS table look-ups (the matrix is linear), not
//     2k select-XORs.
//   * the table lives in HBM as 64-bit slots; claim = one 64-bit atomicCAS,
//     increment = one 64-bit atomicAdd.  Integer / indexing work: no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kmer_core.hpp"

namespace jfgpu {

constexpr int kBlock = 256;
constexpr int kPerThread = kPerLane;           // positions per lane per tile (16)
constexpr int kTilePos = kBlock * kPerThread;  // 4096 sequence positions per block iteration

enum Counter : int { CTR_FULL = 0, CTR_MERS = 1, CTR_OVF_FULL = 2, CTR_OVF_USED = 3, CTR_MISROUTED = 4, CTR_DIRECT = 5,
                     CTR_T_ITEMS = 6, CTR_T_QUEUED = 7 /* tile stage: items placed / items past rank 3 (the sampling launch of a flush) */,
                     CTR_PROF0 = 8 /* .. 11: phase clocks of a -DJFGPU_TILE_PROF build */, CTR_COUNT = 12 };

// -DJFGPU_PHASE_PROF builds: shader clocks per phase of the partition kernels, as wave 0 of every block sees them,
// summed over blocks into a device array the host prints at jfgpu_sync (tools/build_libs.sh builds it, JFGPU_LIB selects it).  Slots 0-7 P1, 8-15 P2, 16-23 T.
#ifdef JFGPU_PHASE_PROF
__device__ unsigned long long g_phase_prof[24];
struct PhaseClk {
  long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long t;
  __device__ PhaseClk() { t = clock64(); }
  __device__ void mark(int i) { const long long n = clock64(); acc[i] += n - t; t = n; }
  __device__ void flush(int base) { if(threadIdx.x == 0) for(int i = 0; i < 8; ++i) if(acc[i]) atomicAdd(&g_phase_prof[base + i], (unsigned long long)acc[i]); }
};
#define JF_PHASE(pc, i) (pc).mark(i)
#define JF_PHASE_FLUSH(pc, base) (pc).flush(base)
#else
struct PhaseClk {};
#define JF_PHASE(pc, i) do {} while(0)
#define JF_PHASE_FLUSH(pc, base) do {} while(0)
#endif

// Bloom counter view (kernels_bloom.hip.hpp); data == nullptr: no filter attached.
struct DevBloom {
  uint32_t* data;             // ceil(m/5) bytes, addressed as dwords
  uint64_t m;                 // number of base-3 cells
  uint64_t recip;             // floor(2^64 / m): x % m without a 64-bit divide (the reference's divisor64, divisor.hpp:64-109)
  uint32_t nh;                // hash functions per key
  uint32_t nbytes;            // key bytes fed to the tables
  uint32_t kind;              // 0: Bloom counter, admit iff check() > 1 (count --bc); 1: one-pass Bloom filter, insert and admit iff
  uint32_t pad_;              //    every bit was already set (count --bf-size, count_main.cc:121-131)
  const uint64_t* tbl1;       // byte tables of the two 64-row matrices
  const uint64_t* tbl2;
  // count --bc on high-coverage input: k-mers the counter has ADMITTED before (bloom_admit_mask), two-way sets of
  // key + 1 (0: empty); nullptr: off.  A hit answers for the ten cell reads; only admitted keys are ever stored, and an
  // admitted key stays admitted (the cells saturate), so the answers are the counter's own (kernels_bloom.hip.hpp)
  uint64_t* cache;
  uint64_t cache_mask;        // sets - 1
};

struct DevTable {
  TableGeom g;
  uint64_t* slots;            // [1 << lsize_l]
  const uint64_t* fwd_tbl;    // [nbytes * 256]   key -> pos
  const uint64_t* inv_tbl;    // [nbytes * 256]   (rem, pos) -> low key bits
  uint64_t* ovf_key;          // overflow side table, keyed by slot index + 1 (0 = empty)
  uint64_t* ovf_cnt;          // units of 2^cnt_bits
  uint64_t ovf_mask;          // capacity - 1
  uint64_t* counters;         // [CTR_COUNT]
  uint32_t max_probe;         // last probe index tried before declaring the tile full
  DevBloom bloom;             // count --bc filter (data == nullptr: none)
  uint8_t* dirty;             // one byte per tile: something was ever inserted (tile_insert may skip reading clean tiles)
};

// Workgroup barrier that orders LDS only.  __syncthreads() also waits for every outstanding
// global store of the wave (s_waitcnt vmcnt(0)), which serialises "write a chunk to HBM" with
// "start the next chunk" in the streaming kernels; here the only cross-wave traffic is LDS.
// (JFGPU_EMU: the same sources compiled for the host by tests/host/hip_emu, where a barrier is a fiber rendezvous.)
#if defined(JFGPU_EMU)
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
#define JF_DYN_LDS(name) unsigned char* name = ::hip_emu::dyn_lds()
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#define JF_DYN_LDS(name) extern __shared__ __align__(16) unsigned char name[]
#endif

__device__ inline bool bloom_admits(const DevBloom& B, uint64_t key);   // kernels_bloom.hip.hpp
__device__ inline uint32_t bloom_admit_mask(const DevBloom& B, const TableGeom& g, const LaneWords& L);   // kernels_bloom.hip.hpp

// ---- overflow side table ----------------------------------------------------
// A slot's count field wrapped: remember `units` x 2^cnt_bits for that slot.  Keyed by
// the slot index, which identifies the key (keys never move while the table lives).
__device__ inline void ovf_add(const DevTable& T, uint64_t slot, uint64_t units) {
  const uint64_t want = slot + 1;
  uint64_t h = (slot * 0x9E3779B97F4A7C15ull) >> 20;
  for(uint64_t p = 0; p <= T.ovf_mask; ++p) {
    const uint64_t s = (h + p) & T.ovf_mask;
    unsigned long long old = atomicCAS((unsigned long long*)&T.ovf_key[s], 0ull, (unsigned long long)want);
    if(old == 0ull) atomicAdd((unsigned long long*)&T.counters[CTR_OVF_USED], 1ull);
    if(old == 0ull || old == want) {
      atomicAdd((unsigned long long*)&T.ovf_cnt[s], (unsigned long long)units);
      return;
    }
  }
  atomicAdd((unsigned long long*)&T.counters[CTR_OVF_FULL], 1ull);
}

__device__ inline uint64_t ovf_get(const DevTable& T, uint64_t slot) {
  const uint64_t want = slot + 1;
  uint64_t h = (slot * 0x9E3779B97F4A7C15ull) >> 20;
  for(uint64_t p = 0; p <= T.ovf_mask; ++p) {
    const uint64_t s = (h + p) & T.ovf_mask;
    const uint64_t k = T.ovf_key[s];
    if(k == 0) return 0;
    if(k == want) return T.ovf_cnt[s];
  }
  return 0;
}

__device__ inline uint64_t full_count(const DevTable& T, uint64_t word, uint64_t slot, bool have_ovf) {
  uint64_t c = slot_count(T.g, word);
  if(have_ovf) c += ovf_get(T, slot) << T.g.cnt_bits;
  return c;
}

// ---- slot access ------------------------------------------------------------------
// 64-bit slots, or 32-bit slots holding the same [count | occ | tag] fields (TableGeom::slot32).  The branch is
// uniform over the whole grid (a scalar branch); slot words travel as uint64_t either way.
__device__ __forceinline__ uint64_t slot_ld(const DevTable& T, uint64_t i) {
  return T.g.slot32 ? (uint64_t)reinterpret_cast<const uint32_t*>(T.slots)[i] : T.slots[i];
}
__device__ __forceinline__ uint64_t slot_ld_relaxed(const DevTable& T, uint64_t i) {
  if(T.g.slot32) return __hip_atomic_load(reinterpret_cast<const uint32_t*>(T.slots) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_load(&T.slots[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t slot_cas(const DevTable& T, uint64_t i, uint64_t cmp, uint64_t val) {
  if(T.g.slot32) return atomicCAS(reinterpret_cast<unsigned int*>(T.slots) + i, (unsigned int)cmp, (unsigned int)val);
  return atomicCAS((unsigned long long*)&T.slots[i], (unsigned long long)cmp, (unsigned long long)val);
}
__device__ __forceinline__ uint64_t slot_add_rtn(const DevTable& T, uint64_t i, uint64_t add) {
  if(T.g.slot32) return atomicAdd(reinterpret_cast<unsigned int*>(T.slots) + i, (unsigned int)add);
  return atomicAdd((unsigned long long*)&T.slots[i], (unsigned long long)add);
}
__device__ __forceinline__ void slot_add(const DevTable& T, uint64_t i, uint64_t add) {
  if(T.g.slot32) __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(T.slots) + i, (unsigned int)add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_fetch_add((unsigned long long*)&T.slots[i], (unsigned long long)add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- insert / increment -------------------------------------------------------
// large_hash_array.hpp:509-597 (claim_key) + :741-752 (add_val), restated for a
// 64-bit [count|occ|tag] slot: CAS the whole word from 0 to claim, atomicAdd on
// the top field to increment.  Returns true when the key was new.
// RETURNING selects the add that reads back the old value (needed only when the
// count field can wrap; a fire-and-forget add otherwise).
template <bool RETURNING>
__device__ inline bool table_add(const DevTable& T, const uint64_t* fwd_lds, uint64_t key, uint64_t cnt) {
  const TableGeom& g = T.g;
  const uint64_t pos = hash_tables(fwd_lds, key, g.nbytes);
  const SlotAddr a = slot_addr(g, pos);
  if(a.shard != g.shard_id) {  // a key that belongs to another GPU's shard must never land here
    atomicAdd((unsigned long long*)&T.counters[CTR_MISROUTED], 1ull);
    return false;
  }
  { uint8_t* d = &T.dirty[a.tile_base >> g.tile_bits]; if(!*d) *d = 1; }
  const uint64_t tag = make_tag(g, key, a.idx0);
  const uint64_t low = g.occ_bit | tag;
  const uint64_t add = cnt << (g.tag_bits + 1);
  const uint64_t neww = add | low;
  const uint32_t tmask = (uint32_t)g.tile_mask;
  for(uint32_t p = 0; p <= T.max_probe; ++p) {
    const uint64_t slot = a.tile_base + probe_lin(a.idx0, p, tmask);
    const uint64_t old = slot_cas(T, slot, 0, neww);
    if(old == 0ull) return true;
    if((old & g.low_mask) == low) {
      if(RETURNING) {
        const uint64_t prev = slot_add_rtn(T, slot, add);
        if((prev >> (g.tag_bits + 1)) + cnt > g.cnt_max) ovf_add(T, slot, 1);
      } else {
        slot_add(T, slot, add);
      }
      return false;
    }
  }
  atomicAdd((unsigned long long*)&T.counters[CTR_FULL], 1ull);  // tile exhausted: "Hash full"
  return false;
}

// hash_counter::update_add (hash_counter.hpp:150-166, large_hash_array.hpp update_add): increment only
// if the key is already there -- the UPDATE pass of `count --if` (count_main.cc:173-181).  Probes like a
// look-up: the first empty slot ends the search.
template <bool RETURNING>
__device__ inline bool table_update_add(const DevTable& T, const uint64_t* fwd_lds, uint64_t key, uint64_t cnt) {
  const TableGeom& g = T.g;
  const uint64_t pos = hash_tables(fwd_lds, key, g.nbytes);
  const SlotAddr a = slot_addr(g, pos);
  if(a.shard != g.shard_id) { atomicAdd((unsigned long long*)&T.counters[CTR_MISROUTED], 1ull); return false; }
  const uint64_t low = g.occ_bit | make_tag(g, key, a.idx0);
  const uint64_t add = cnt << (g.tag_bits + 1);
  const uint32_t tmask = (uint32_t)g.tile_mask;
  for(uint32_t p = 0; p <= T.max_probe; ++p) {
    const uint64_t slot = a.tile_base + probe_lin(a.idx0, p, tmask);
    const uint64_t old = slot_ld_relaxed(T, slot);
    if(old == 0ull) return false;
    if((old & g.low_mask) == low) {
      if(RETURNING) {
        const uint64_t prev = slot_add_rtn(T, slot, add);
        if((prev >> (g.tag_bits + 1)) + cnt > g.cnt_max) ovf_add(T, slot, 1);
      } else {
        slot_add(T, slot, add);
      }
      return true;
    }
  }
  return false;
}

// Arbitrary 64-bit increment (hash_counter::add(key, val)): split into field-sized pieces.
__device__ inline bool table_add_val(const DevTable& T, const uint64_t* fwd_lds, uint64_t key, uint64_t val) {
  const TableGeom& g = T.g;
  const uint64_t lowpart = val & g.cnt_max;
  const uint64_t units = g.cnt_bits >= 64 ? 0 : (val >> g.cnt_bits);
  // one claim-or-find with the low part (possibly 0: still claims the key, like set())
  const uint64_t pos = hash_tables(fwd_lds, key, g.nbytes);
  const SlotAddr a = slot_addr(g, pos);
  if(a.shard != g.shard_id) {
    atomicAdd((unsigned long long*)&T.counters[CTR_MISROUTED], 1ull);
    return false;
  }
  { uint8_t* d = &T.dirty[a.tile_base >> g.tile_bits]; if(!*d) *d = 1; }
  const uint64_t tag = make_tag(g, key, a.idx0);
  const uint64_t low = g.occ_bit | tag;
  const uint64_t add = lowpart << (g.tag_bits + 1);
  const uint32_t tmask = (uint32_t)g.tile_mask;
  for(uint32_t p = 0; p <= T.max_probe; ++p) {
    const uint64_t slot = a.tile_base + probe_lin(a.idx0, p, tmask);
    const uint64_t old = slot_cas(T, slot, 0, add | low);
    bool mine = false, is_new = false;
    if(old == 0ull) { mine = true; is_new = true; }
    else if((old & g.low_mask) == low) {
      mine = true;
      if(add) {
        const uint64_t prev = slot_add_rtn(T, slot, add);
        if((prev >> (g.tag_bits + 1)) + lowpart > g.cnt_max) ovf_add(T, slot, 1);
      }
    }
    if(mine) {
      if(units) ovf_add(T, slot, units);
      return is_new;
    }
  }
  atomicAdd((unsigned long long*)&T.counters[CTR_FULL], 1ull);
  return false;
}

__device__ inline LaneWords stage_tile(const uint8_t* __restrict__ base, int64_t tile_start, int64_t lo, int64_t hi,
                                       uint32_t* s_codes, uint32_t* s_inv) {
  const int tid = threadIdx.x;
  uint32_t c, v;
  load_pack16(base, tile_start + 16 * tid, lo, hi, c, v);
  s_codes[tid + 2] = c; s_inv[tid + 2] = v;
  if(tid < 2) {
    uint32_t hc, hv;
    load_pack16(base, tile_start - 32 + 16 * tid, lo, hi, hc, hv);
    s_codes[tid] = hc; s_inv[tid] = hv;
  }
  lds_barrier();
  LaneWords L;
  L.cur = c;
  L.p1 = s_codes[tid + 1];
  L.p2 = s_codes[tid];
  L.inv48 = ((uint64_t)s_inv[tid] << 32) | ((uint64_t)s_inv[tid + 1] << 16) | v;
  return L;
}

// The same in two halves, so a kernel can issue the loads of its NEXT tile before it works on the
// current one: tile_fetch only loads (4 + 4 registers), tile_stage packs and publishes to LDS.
struct TileRaw { uint32_t w[4]; uint32_t h[4]; };
__device__ inline void raw16(const uint8_t* __restrict__ base, int64_t off, int64_t lo, int64_t hi, uint32_t w[4]) {
  w[0] = w[1] = w[2] = w[3] = 0;
  if(off + 16 <= lo || off >= hi || off < 0) return;
  if(off + 16 <= hi) load16(base + off, w);
  else for(int i = 0; i < 16 && off + i < hi; ++i) w[i >> 2] |= (uint32_t)base[off + i] << (8 * (i & 3));
}
__device__ inline void edges16(const uint32_t w[4], int64_t off, int64_t lo, int64_t hi, uint32_t& codes, uint32_t& inval) {
  if(off + 16 <= lo || off >= hi || off < 0) { codes = 0; inval = 0xFFFFu; return; }
  pack16(w, codes, inval);
  if(off < lo) inval |= (0xFFFFu << (16 - (int)(lo - off))) & 0xFFFFu;
  if(off + 16 > hi) inval |= (1u << (int)(off + 16 - hi)) - 1u;
}
__device__ inline TileRaw tile_fetch(const uint8_t* __restrict__ base, int64_t tile_start, int64_t lo, int64_t hi) {
  TileRaw R;
  raw16(base, tile_start + 16 * (int64_t)threadIdx.x, lo, hi, R.w);
  R.h[0] = R.h[1] = R.h[2] = R.h[3] = 0;
  if(threadIdx.x < 2) raw16(base, tile_start - 32 + 16 * (int64_t)threadIdx.x, lo, hi, R.h);
  return R;
}
__device__ inline LaneWords tile_stage(const TileRaw& R, int64_t tile_start, int64_t lo, int64_t hi, uint32_t* s_codes, uint32_t* s_inv) {
  const int tid = threadIdx.x;
  uint32_t c, v;
  edges16(R.w, tile_start + 16 * (int64_t)tid, lo, hi, c, v);
  s_codes[tid + 2] = c; s_inv[tid + 2] = v;
  if(tid < 2) {
    uint32_t hc, hv;
    edges16(R.h, tile_start - 32 + 16 * (int64_t)tid, lo, hi, hc, hv);
    s_codes[tid] = hc; s_inv[tid] = hv;
  }
  lds_barrier();
  LaneWords L;
  L.cur = c;
  L.p1 = s_codes[tid + 1];
  L.p2 = s_codes[tid];
  L.inv48 = ((uint64_t)s_inv[tid] << 32) | ((uint64_t)s_inv[tid + 1] << 16) | v;
  return L;
}

__device__ inline void load_tables_lds(uint64_t* dst, const uint64_t* src, uint32_t nbytes) {
  for(uint32_t i = threadIdx.x; i < nbytes * 256; i += blockDim.x) dst[i] = src[i];
}

// ---- K2+K3 fused: count every k-mer of a contract buffer ----------------------
// base: 16-byte aligned; valid bytes are [lo, hi).
// op: 0 COUNT add(m, 1); 1 PRIME set(m) = claim with count 0; 2 UPDATE update_add(m, 1) (count_main.cc:152-184).
template <bool RETURNING, bool BLOOM>
__global__ __launch_bounds__(kBlock) void count_ascii_kernel(DevTable T, const uint8_t* __restrict__ base,
                                                             int64_t lo, int64_t hi, int op) {
  __shared__ uint64_t s_fwd[8 * 256];
  __shared__ uint32_t s_codes[kBlock + 2];
  __shared__ uint32_t s_inv[kBlock + 2];
  __shared__ int s_abort;
  load_tables_lds(s_fwd, T.fwd_tbl, T.g.nbytes);
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  uint32_t my_mers = 0;
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // "Hash full" already raised by some block: the result is void, stop burning probes (the
    // host reports the error at jfgpu_sync).  One lane polls, the barrier makes it block-uniform
    // and also fences the previous iteration's LDS reads.
    if(threadIdx.x == 0)
      s_abort = __hip_atomic_load(&T.counters[CTR_FULL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    __syncthreads();
    if(s_abort) break;
    const LaneWords L = stage_tile(base, tile * kTilePos, lo, hi, s_codes, s_inv);
    // run-length merge of consecutive identical k-mers (homopolymers / short tandem
    // repeats are the heavy hitters of real data): one atomic per run, not per k-mer.
    uint64_t prev = 0; uint32_t run = 0;
    auto apply = [&](uint64_t key, uint32_t n) {
      if(op == 0) table_add<RETURNING>(T, s_fwd, key, n);
      else if(op == 1) table_add<RETURNING>(T, s_fwd, key, 0);
      else table_update_add<RETURNING>(T, s_fwd, key, n);
    };
    const uint32_t adm = BLOOM ? bloom_admit_mask(T.bloom, T.g, L) : 0xFFFFu;   // count --bc (count_main.cc:115-118); compiled out otherwise
    for_each_kmer(T.g, L, [&](int j, uint64_t key) {
      ++my_mers;
      if(BLOOM && !((adm >> j) & 1u)) return;
      if(run && key == prev) { ++run; return; }
      if(run) apply(prev, run);
      prev = key; run = 1;
    });
    if(run) apply(prev, run);
  }
  // one counter update per wave
  uint64_t w = my_mers;
  for(int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
  if((threadIdx.x & 63) == 0 && w) atomicAdd((unsigned long long*)&T.counters[CTR_MERS], (unsigned long long)w);
}

// ---- hash_counter::add on encoded keys -----------------------------------------
__global__ __launch_bounds__(kBlock) void add_keys_kernel(DevTable T, const uint64_t* __restrict__ keys, uint64_t n,
                                                          uint64_t val, uint8_t* __restrict__ is_new) {
  __shared__ uint64_t s_fwd[8 * 256];
  load_tables_lds(s_fwd, T.fwd_tbl, T.g.nbytes);
  __syncthreads();
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t key = keys[i] & T.g.key_mask;
    const bool nw = table_add_val(T, s_fwd, key, val);
    if(is_new) is_new[i] = nw ? 1 : 0;
  }
}

// Same, but val == 1 and no is_new: the receive side of the multi-GPU exchange.
template <bool RETURNING>
__global__ __launch_bounds__(kBlock) void add_keys_one_kernel(DevTable T, const uint64_t* __restrict__ keys, uint64_t n) {
  __shared__ uint64_t s_fwd[8 * 256];
  load_tables_lds(s_fwd, T.fwd_tbl, T.g.nbytes);
  __syncthreads();
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    table_add<RETURNING>(T, s_fwd, keys[i] & T.g.key_mask, 1);
}

// hash_counter::update_add on encoded keys, val == 1: the receive side of the exchange in the UPDATE pass of count --if
// (count_main.cc:152-184 with --gpus): only keys that are present are counted.
template <bool RETURNING>
__global__ __launch_bounds__(kBlock) void update_keys_one_kernel(DevTable T, const uint64_t* __restrict__ keys, uint64_t n) {
  __shared__ uint64_t s_fwd[8 * 256];
  load_tables_lds(s_fwd, T.fwd_tbl, T.g.nbytes);
  __syncthreads();
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    table_update_add<RETURNING>(T, s_fwd, keys[i] & T.g.key_mask, 1);
}

// ---- get_val_for_key -------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void lookup_kernel(DevTable T, const uint64_t* __restrict__ keys, uint64_t n,
                                                        uint64_t* __restrict__ vals, uint8_t* __restrict__ found,
                                                        int have_ovf) {
  __shared__ uint64_t s_fwd[8 * 256];
  load_tables_lds(s_fwd, T.fwd_tbl, T.g.nbytes);
  __syncthreads();
  const TableGeom& g = T.g;
  const uint32_t tmask = (uint32_t)g.tile_mask;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t key = keys[i] & g.key_mask;
    const uint64_t pos = hash_tables(s_fwd, key, g.nbytes);
    const SlotAddr a = slot_addr(g, pos);
    uint64_t val = 0; uint8_t fnd = 0;
    if(a.shard == g.shard_id) {
      const uint64_t low = g.occ_bit | make_tag(g, key, a.idx0);
      for(uint32_t p = 0; p <= T.max_probe; ++p) {
        const uint64_t slot = a.tile_base + probe_lin(a.idx0, p, tmask);
        const uint64_t w = slot_ld(T, slot);
        if(w == 0) break;
        if((w & g.low_mask) == low) { val = full_count(T, w, slot, have_ovf); fnd = 1; break; }
      }
    }
    vals[i] = val;
    if(found) found[i] = fnd;
  }
}

// ---- multi-GPU routing: count per shard, then scatter ----------------------------
// Pass A: how many k-mers of this buffer belong to each shard.
// BLOOM: count --bc with --gpus -- the sender asks its copy of the Bloom counter, what it does not admit never travels.
template <bool BLOOM = false>
__global__ __launch_bounds__(kBlock) void partition_count_kernel(DevTable T, const uint8_t* __restrict__ base,
                                                                 int64_t lo, int64_t hi,
                                                                 unsigned long long* __restrict__ shard_counts) {
  __shared__ uint64_t s_fwd[8 * 256];
  __shared__ uint32_t s_codes[kBlock + 2];
  __shared__ uint32_t s_inv[kBlock + 2];
  __shared__ uint32_t s_hist[256];
  load_tables_lds(s_fwd, T.fwd_tbl, T.g.nbytes);
  const uint32_t n_shards = 1u << T.g.shard_bits;
  for(uint32_t i = threadIdx.x; i < n_shards; i += blockDim.x) s_hist[i] = 0;
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();
    const LaneWords L = stage_tile(base, tile * kTilePos, lo, hi, s_codes, s_inv);
    const uint32_t adm = BLOOM ? bloom_admit_mask(T.bloom, T.g, L) : 0xFFFFu;
    for_each_kmer(T.g, L, [&](int j, uint64_t key) {
      if(BLOOM && !((adm >> j) & 1u)) return;
      const uint64_t pos = hash_tables(s_fwd, key, T.g.nbytes);
      atomicAdd(&s_hist[(uint32_t)(pos >> T.g.lsize_l)], 1u);
    });
  }
  __syncthreads();
  for(uint32_t i = threadIdx.x; i < n_shards; i += blockDim.x)
    if(s_hist[i]) atomicAdd(&shard_counts[i], (unsigned long long)s_hist[i]);
}

// Pass B: write each k-mer into its shard's region.  cursors[s] starts at the
// region's offset; a block reserves its share with one atomic per (tile, shard).
template <bool BLOOM = false>
__global__ __launch_bounds__(kBlock) void partition_scatter_kernel(DevTable T, const uint8_t* __restrict__ base,
                                                                   int64_t lo, int64_t hi,
                                                                   unsigned long long* __restrict__ cursors,
                                                                   uint64_t* __restrict__ out) {
  __shared__ uint64_t s_fwd[8 * 256];
  __shared__ uint32_t s_codes[kBlock + 2];
  __shared__ uint32_t s_inv[kBlock + 2];
  __shared__ uint32_t s_hist[256];
  __shared__ unsigned long long s_base[256];
  load_tables_lds(s_fwd, T.fwd_tbl, T.g.nbytes);
  const uint32_t n_shards = 1u << T.g.shard_bits;
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();
    for(uint32_t i = threadIdx.x; i < n_shards; i += blockDim.x) s_hist[i] = 0;
    const LaneWords L = stage_tile(base, tile * kTilePos, lo, hi, s_codes, s_inv);  // contains a barrier
    // indexed by the unrolled position j (compile-time after unrolling) so these stay in VGPRs
    uint64_t keys[kPerThread]; uint32_t shard[kPerThread]; uint32_t rank[kPerThread]; uint32_t vmask = 0;
    const uint32_t adm = BLOOM ? bloom_admit_mask(T.bloom, T.g, L) : 0xFFFFu;      // (the same answers as the count pass: the counter is read-only here)
    for_each_kmer(T.g, L, [&](int j, uint64_t key) {
      if(BLOOM && !((adm >> j) & 1u)) return;
      const uint64_t pos = hash_tables(s_fwd, key, T.g.nbytes);
      const uint32_t s = (uint32_t)(pos >> T.g.lsize_l);
      keys[j] = key; shard[j] = s;
      rank[j] = atomicAdd(&s_hist[s], 1u);
      vmask |= 1u << j;
    });
    __syncthreads();
    for(uint32_t i = threadIdx.x; i < n_shards; i += blockDim.x)
      s_base[i] = s_hist[i] ? atomicAdd(&cursors[i], (unsigned long long)s_hist[i]) : 0ull;
    __syncthreads();
#pragma unroll
    for(int j = 0; j < kPerThread; ++j)
      if((vmask >> j) & 1u) out[s_base[shard[j]] + rank[j]] = keys[j];
  }
}

// ---- cooperative size doubling (hash_counter::double_size, hash_counter.hpp:200-238) -------------
// Every entry of the old table is re-derived (slot -> key by the inverse tables) and inserted with its
// full count into the new, twice as large table (one more matrix row).  Hash tables are read through
// the caches here: growth is rare and the kernel is bound by the random inserts anyway.
__global__ __launch_bounds__(kBlock) void rehash_kernel(DevTable old, DevTable neu, int have_ovf) {
  const uint64_t n = 1ull << old.g.lsize_l;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t w = slot_ld(old, i);
    if(!w) continue;
    const uint64_t key = slot_key(old.g, old.inv_tbl, w, i & ~old.g.tile_mask);
    table_add_val(neu, neu.fwd_tbl, key, full_count(old, w, i, have_ovf));
  }
}

// ---- stats (stats_main.cc:33-46) ---------------------------------------------------
// out: [0] unique [1] distinct [2] total [3] max
__global__ __launch_bounds__(kBlock) void stats_kernel(DevTable T, uint64_t lower, uint64_t upper, int have_ovf,
                                                       unsigned long long* __restrict__ out) {
  const uint64_t n = 1ull << T.g.lsize_l;
  uint64_t uniq = 0, dist = 0, tot = 0, mx = 0;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t w = slot_ld(T, i);
    if(!w) continue;
    const uint64_t c = full_count(T, w, i, have_ovf);
    if(c < lower || c > upper) continue;
    uniq += (c == 1); ++dist; tot += c; mx = c > mx ? c : mx;
  }
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

// ---- content digest (at-scale parity) -----------------------------------------
// An order-independent checksum of the {k-mer -> count} multiset: per entry h = mix(..mix(mix(seed ^ w0) ^ w1).. ^ count)
// over the key's little-endian 64-bit words, then out[0] += 1, out[1] += count, out[2] += h, out[3] ^= h (all mod 2^64).
// The reference driver (oracle/ref_drivers/ref_jf.cc, `count --digest` / `digest`) computes the same four numbers from the
// reference's own table, so two runs over 10 Gbp are compared without writing or sorting 86 GB of records.
__device__ __host__ inline uint64_t digest_mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
constexpr uint64_t kDigestSeed = 0x9E3779B97F4A7C15ull;

__device__ inline void digest_reduce(uint64_t n, uint64_t tot, uint64_t sum, uint64_t x, unsigned long long* __restrict__ out) {
  for(int o = 32; o > 0; o >>= 1) {
    n += __shfl_down(n, o, 64); tot += __shfl_down(tot, o, 64); sum += __shfl_down(sum, o, 64); x ^= __shfl_down(x, o, 64);
  }
  if((threadIdx.x & 63) == 0 && n) {
    atomicAdd(&out[0], (unsigned long long)n); atomicAdd(&out[1], (unsigned long long)tot);
    atomicAdd(&out[2], (unsigned long long)sum); atomicXor(&out[3], (unsigned long long)x);
  }
}

__global__ __launch_bounds__(kBlock) void digest_kernel(DevTable T, uint64_t lower, uint64_t upper, int have_ovf,
                                                        unsigned long long* __restrict__ out) {
  __shared__ uint64_t s_inv[8 * 256];
  load_tables_lds(s_inv, T.inv_tbl, T.g.nbytes);
  __syncthreads();
  const uint64_t n = 1ull << T.g.lsize_l;
  uint64_t cnt = 0, tot = 0, sum = 0, x = 0;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t w = slot_ld(T, i);
    if(!w) continue;
    const uint64_t c = full_count(T, w, i, have_ovf);
    if(c < lower || c > upper) continue;
    const uint64_t key = slot_key(T.g, s_inv, w, i & ~T.g.tile_mask);
    const uint64_t h = digest_mix(digest_mix(kDigestSeed ^ key) ^ c);
    ++cnt; tot += c; sum += h; x ^= h;
  }
  digest_reduce(cnt, tot, sum, x, out);
}

// ---- histo (histo_main.cc:34-45) -----------------------------------------------------
constexpr uint32_t kHistoLds = 8192;  // buckets privatised per block
__global__ __launch_bounds__(kBlock) void histo_kernel(DevTable T, uint64_t hbase, uint64_t hceil, uint64_t inc,
                                                       uint64_t nb, int have_ovf, unsigned long long* __restrict__ histo) {
  __shared__ uint32_t s_h[kHistoLds];
  const uint32_t nl = nb < kHistoLds ? (uint32_t)nb : kHistoLds;
  for(uint32_t i = threadIdx.x; i < nl; i += blockDim.x) s_h[i] = 0;
  __syncthreads();
  const uint64_t n = 1ull << T.g.lsize_l;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t w = slot_ld(T, i);
    if(!w) continue;
    const uint64_t c = full_count(T, w, i, have_ovf);
    uint64_t b;
    if(c < hbase) b = 0; else if(c > hceil) b = nb - 1; else b = (c - hbase) / inc;
    if(b < nl) atomicAdd(&s_h[(uint32_t)b], 1u); else atomicAdd(&histo[b], 1ull);
  }
  __syncthreads();
  for(uint32_t i = threadIdx.x; i < nl; i += blockDim.x)
    if(s_h[i]) atomicAdd(&histo[i], (unsigned long long)s_h[i]);
}

// ---- sorted dump ------------------------------------------------------------------------
// Pass 1: records per tile after the [lower, upper] filter.
__global__ __launch_bounds__(kBlock) void tile_count_kernel(DevTable T, uint64_t lower, uint64_t upper, int have_ovf,
                                                            uint64_t n_tiles, uint32_t* __restrict__ tile_counts) {
  __shared__ uint32_t s_sum[kBlock / 64];
  const uint32_t tsz = 1u << T.g.tile_bits;
  for(uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t tb = tile << T.g.tile_bits;
    uint32_t c = 0;
    for(uint32_t i = threadIdx.x; i < tsz; i += blockDim.x) {
      const uint64_t w = slot_ld(T, tb + i);
      if(!w) continue;
      const uint64_t cnt = full_count(T, w, tb + i, have_ovf);
      c += (cnt >= lower && cnt <= upper);
    }
    for(int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if(threadIdx.x == 0) { uint32_t s = 0; for(int i = 0; i < kBlock / 64; ++i) s += s_sum[i]; tile_counts[tile] = s; }
    __syncthreads();
  }
}

// Pass 2: one block per tile.  Load the tile into LDS, bitonic-sort by tag (== (pos, key)
// order, mer_heap.hpp:26-30), rebuild each key with the inverse tables and emit
// fixed-width records (binary_dumper.hpp:36-40) at the tile's record offset.
// Dynamic LDS: tsz * 8 (words) + tsz * 2 (slot index) + nbytes * 2048 (inverse tables).
__global__ __launch_bounds__(kBlock) void dump_tiles_kernel(DevTable T, uint64_t lower, uint64_t upper, int have_ovf,
                                                            uint64_t tile0, uint64_t n_tiles,
                                                            const uint64_t* __restrict__ tile_offsets,  // record offset of tile (relative to tile0's)
                                                            uint8_t* __restrict__ out, uint32_t key_bytes, uint32_t val_bytes) {
  JF_DYN_LDS(s_raw);
  const uint32_t tsz = 1u << T.g.tile_bits;
  uint64_t* s_w = reinterpret_cast<uint64_t*>(s_raw);
  uint64_t* s_invt = s_w + tsz;
  uint16_t* s_idx = reinterpret_cast<uint16_t*>(s_invt + T.g.nbytes * 256);
  load_tables_lds(s_invt, T.inv_tbl, T.g.nbytes);
  const uint64_t tagmask = T.g.occ_bit - 1;
  const uint64_t SENT = ~T.g.occ_bit;      // no stored slot equals it (they all have the occupied bit), and it sorts after every tag
  const uint64_t maxval = val_bytes >= 8 ? ~0ull : ((1ull << (8 * val_bytes)) - 1);
  const uint32_t rec = key_bytes + val_bytes;
  for(uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const uint64_t tile = tile0 + t;
    const uint64_t tb = tile << T.g.tile_bits;
    __syncthreads();
    for(uint32_t i = threadIdx.x; i < tsz; i += blockDim.x) {
      uint64_t w = slot_ld(T, tb + i);
      uint64_t sk = SENT;
      if(w) {
        const uint64_t cnt = full_count(T, w, tb + i, have_ovf);
        if(cnt >= lower && cnt <= upper) sk = w;
      }
      s_w[i] = sk; s_idx[i] = (uint16_t)i;
    }
    __syncthreads();
    // bitonic sort ascending on (word & tagmask), sentinels last
    for(uint32_t size = 2; size <= tsz; size <<= 1) {
      for(uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
        for(uint32_t i = threadIdx.x; i < tsz / 2; i += blockDim.x) {
          const uint32_t lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));
          const uint32_t hi = lo | stride;
          const bool up = (lo & size) == 0;
          const uint64_t a = s_w[lo], b = s_w[hi];
          const uint64_t ka = a == SENT ? SENT : (a & tagmask), kb = b == SENT ? SENT : (b & tagmask);
          if((ka > kb) == up) {
            s_w[lo] = b; s_w[hi] = a;
            const uint16_t ia = s_idx[lo]; s_idx[lo] = s_idx[hi]; s_idx[hi] = ia;
          }
        }
        __syncthreads();
      }
    }
    uint8_t* dst0 = out + tile_offsets[t] * rec;
    for(uint32_t i = threadIdx.x; i < tsz; i += blockDim.x) {
      const uint64_t w = s_w[i];
      if(w == SENT) continue;
      const uint64_t key = slot_key(T.g, s_invt, w, tb);
      uint64_t cnt = slot_count(T.g, w);
      if(have_ovf) cnt += ovf_get(T, tb + s_idx[i]) << T.g.cnt_bits;
      if(cnt > maxval) cnt = maxval;
      uint8_t* d = dst0 + (uint64_t)i * rec;
      for(uint32_t b = 0; b < key_bytes; ++b) d[b] = (uint8_t)(key >> (8 * b));
      for(uint32_t b = 0; b < val_bytes; ++b) d[key_bytes + b] = (uint8_t)(cnt >> (8 * b));
    }
  }
}

// ---- synthetic reads (generate_sequence-like: iid uniform bases) ---------------------------
__device__ inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(kBlock) void gen_reads_kernel(uint8_t* __restrict__ out, uint64_t first_read, uint64_t n_reads,
                                                           uint32_t read_len, uint64_t seed) {
  const uint64_t stride = (uint64_t)read_len + 1;
  const uint64_t total = n_reads * stride;
  for(uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v * 16 < total; v += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p0 = v * 16;
    uint64_t r = p0 / stride; uint32_t off = (uint32_t)(p0 - r * stride);
    uint32_t w[4] = {0, 0, 0, 0};
    uint64_t draw = 0; uint64_t draw_id = ~0ull;
    for(int i = 0; i < 16; ++i) {
      uint32_t ch = 0;
      if(p0 + i < total) {
        if(off == read_len) ch = 'N';
        else {
          const uint64_t id = (first_read + r) * 64 + (off >> 5);   // 32 bases per draw
          if(id != draw_id) { draw_id = id; draw = mix64(mix64(seed + 0x9E3779B97F4A7C15ull * (id + 1))); }
          ch = (uint32_t)("ACGT"[(draw >> (2 * (off & 31))) & 3]);
        }
      }
      w[i >> 2] |= ch << (8 * (i & 3));
      if(++off == stride) { off = 0; ++r; }
    }
    if(p0 + 16 <= total) *reinterpret_cast<uint4*>(out + p0) = make_uint4(w[0], w[1], w[2], w[3]);
    else for(int i = 0; i < 16 && p0 + i < total; ++i) out[p0 + i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
  }
}

// Distribution "G" of BASELINE.md section 3: reads sampled at uniform positions and strands from a uniform random
// genome (never materialised: base g of the genome is a pure function of (seed, g)), with iid substitutions.
// Same output layout as gen_reads_kernel (read_len bases + one 'N' per read), counter-based, any slice reproducible.
__device__ inline uint32_t genome_base(uint64_t gseed, uint64_t g) {
  const uint64_t d = mix64(mix64(gseed + 0x9E3779B97F4A7C15ull * ((g >> 5) + 1)));
  return (uint32_t)(d >> (2 * (g & 31))) & 3u;
}
__global__ __launch_bounds__(kBlock) void gen_genome_reads_kernel(uint8_t* __restrict__ out, uint64_t first_read, uint64_t n_reads,
                                                                  uint32_t read_len, uint64_t genome_len, uint32_t sub_per_64k,
                                                                  uint64_t seed) {
  const uint64_t stride = (uint64_t)read_len + 1;
  const uint64_t total = n_reads * stride;
  const uint64_t gseed = mix64(seed ^ 0x67656E6F6D65ull);
  for(uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v * 16 < total; v += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p0 = v * 16;
    uint64_t r = p0 / stride; uint32_t off = (uint32_t)(p0 - r * stride);
    uint32_t w[4] = {0, 0, 0, 0};
    uint64_t rid = ~0ull, pos = 0; bool rev = false;
    for(int i = 0; i < 16; ++i) {
      uint32_t ch = 0;
      if(p0 + i < total) {
        if(off == read_len) ch = 'N';
        else {
          if(r != rid) {
            rid = r;
            const uint64_t h = mix64(mix64(seed + 0xD1B54A32D192ED03ull * (first_read + r + 1)));
            pos = (h >> 1) % (genome_len - read_len + 1); rev = (h & 1) != 0;
          }
          uint32_t b = rev ? 3u - genome_base(gseed, pos + (read_len - 1 - off)) : genome_base(gseed, pos + off);
          const uint64_t e = mix64(seed + 0xA24BAED4963EE407ull * ((first_read + r) * 4096 + off + 1));   // substitution draw of this base
          if((uint32_t)(e & 0xFFFFu) < sub_per_64k) b = (b + 1 + (uint32_t)((e >> 16) % 3)) & 3u;
          ch = (uint32_t)("ACGT"[b]);
        }
      }
      w[i >> 2] |= ch << (8 * (i & 3));
      if(++off == stride) { off = 0; ++r; }
    }
    if(p0 + 16 <= total) *reinterpret_cast<uint4*>(out + p0) = make_uint4(w[0], w[1], w[2], w[3]);
    else for(int i = 0; i < 16 && p0 + i < total; ++i) out[p0 + i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
  }
}

// ---- random-access roofline probes (R_gups) ------------------------------------
// mode 0: fire-and-forget atomicAdd   mode 1: returning atomicAdd   mode 2: atomicCAS(0 -> x)
// mode 3: plain load + dependent fire-and-forget atomicAdd
__global__ __launch_bounds__(kBlock) void gups_kernel(uint64_t* __restrict__ tab, uint64_t mask, uint64_t n, int mode,
                                                      uint64_t seed, unsigned long long* __restrict__ sink) {
  uint64_t acc = 0;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t s = mix64(seed + i * 0x9E3779B97F4A7C15ull) & mask;
    unsigned long long* a = (unsigned long long*)&tab[s];
    if(mode == 0) __hip_atomic_fetch_add(a, 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if(mode == 1) acc += atomicAdd(a, 1ull << 32);
    else if(mode == 2) acc += atomicCAS(a, 0ull, (unsigned long long)(i | 1));
    else { const uint64_t v = tab[s]; __hip_atomic_fetch_add(a, (1ull << 32) + (v & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  }
  if(acc == 0x123456789ull) atomicAdd(sink, 1ull);
}

}  // namespace jfgpu
