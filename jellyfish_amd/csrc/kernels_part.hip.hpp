// jellyfish_amd/csrc/kernels_part.hip.hpp -- the partitioned insert path (gfx950).
//
// Why: device-scope 64-bit atomics on MI355X retire at ~21 G/s whatever the table size
// (profiles/r01_gups_probe.txt), so the fused kernel in kernels.hip.hpp cannot pass ~20 G
// k-mers/s.  Sequential HBM traffic is two orders of magnitude cheaper per byte, so large
// batches are inserted WITHOUT global atomics:
//
//   P1  encode+canonical+hash, then radix-partition the hashed k-mers by the upper b1 bits of
//       their table position into "items" (p1_kernel, count pass + scatter pass)
//   P2  (tables with more than 2^11 tiles) partition one P1 bucket by the next b2 bits so that
//       one sub-bucket = one 64 KiB tile (p2_kernel, count + scatter)
//   T   one workgroup owns one tile: tile -> LDS, LDS ds_cmpst/ds_add inserts (duplicates
//       aggregate here for free), LDS -> tile (tile_insert_kernel)
//
// The table format is untouched (same slots, same tile-local triangular probing as
// large_hash_array.hpp:509-597 restated in kernels.hip.hpp::table_add), so lookups, stats and
// the sorted dump do not care which path inserted a key; results are bit-identical.
//
// item = (tile_rest << tag_bits) | tag : everything about a k-mer except the P1 bucket.
//   tag       = (idx0 << rem_bits) | (key >> lsize_g)   exactly the slot's tag field
//   tile_rest = the b2 tile-index bits P2 still has to resolve
// 32-bit items whenever lsize_l - b1 + rem_bits <= 32 (k=21 at 2^34 slots: 24 + 8 = 32).
//
// No global atomics in the passes either: the count pass leaves a [block][bucket] histogram
// matrix, a scan turns it into each block's private write cursor per bucket, and the scatter
// pass replays the same block->chunk assignment.
#pragma once
#include "kernels.hip.hpp"

namespace jfgpu {

// Hides a value from the optimiser (the value itself is unchanged): what was derived from it before this point is not
// kept alive across it, it is derived again.  The unrolled per-item code otherwise carries bucket numbers, shift counts
// and addresses of all of a lane's items from the rank requests to the placement, and spills.
#if defined(JFGPU_EMU)
#define JF_OPAQUE(x) do {} while(0)
#else
#define JF_OPAQUE(x) asm volatile("" : "+v"(x))
#endif

constexpr int kPBlock = 1024;                 // threads per block in the partition passes
constexpr int kPTilePos = kPBlock * kPerLane; // 16384 sequence positions per block iteration
constexpr int kMaxBuckets = 2048;             // per pass
constexpr int kMaxSeg = 128;                  // pending batches per flush

struct PartGeom {
  uint32_t b1, b2;          // bits consumed by P1 / P2 (b2 == 0: P1 buckets are tiles)
  uint32_t item_bits;       // lsize_l - b1 + rem_bits
  uint32_t rest_shift;      // lsize_l - b1: low local-position bits kept in the item
};

// Pending batches: batch s holds items grouped by P1 bucket, off[s][j] .. off[s][j+1].
// sh[s] == 0: buckets are packed, bucket j = [off[j], off[j+1]).
// sh[s] == 1: "granule" batches of the single-pass P1: bucket j = [off[2j], off[2j+1]) inside its own
//             fixed-capacity region, and entries equal to the all-ones item are holes to skip.
struct SegList {
  const void* items[kMaxSeg];
  const uint64_t* off[kMaxSeg];
  uint32_t sh[kMaxSeg];
  uint32_t n;
};
__device__ inline uint64_t seg_lo(const SegList& S, uint32_t s, uint32_t j) { return S.off[s][(size_t)j << S.sh[s]]; }
__device__ inline uint64_t seg_hi(const SegList& S, uint32_t s, uint32_t j) { return S.off[s][((size_t)j << S.sh[s]) + 1]; }

#ifndef JFGPU_KGRAN
#define JFGPU_KGRAN 64
#endif
constexpr uint32_t kGran = JFGPU_KGRAN;       // items per reservation of the single-pass P1

template <typename ITEM>
__device__ inline ITEM make_item(const TableGeom& g, const PartGeom& P, uint64_t key, uint64_t local) {
  const uint64_t rest = local & ((1ull << P.rest_shift) - 1);
  const uint64_t rem = g.lsize_g >= 64 ? 0 : (key >> g.lsize_g);
  return (ITEM)((rest << g.rem_bits) | rem);
}

// Runs of identical consecutive k-mers (homopolymers, tandem repeats) bypass the partition: one
// table_add per run.  They are rare, so the hot unrolled loops only note that a lane saw one and
// this rolled replay of the lane's 16 positions applies them -- one copy of the probing code per
// kernel instead of seventeen.
template <bool RETURNING, bool BLOOM>
__device__ inline uint32_t apply_runs(const DevTable& T, const uint64_t* s_fwd, const LaneWords& L) {
  const TableGeom& g = T.g;
  const uint32_t k = g.k;
  uint64_t fw = (((uint64_t)L.p2 << 32) | L.p1) & g.key_mask;
  uint64_t rc = revcomp64(fw, k);
  const uint64_t kwin = k >= 64 ? ~0ull : ((1ull << k) - 1);
  const uint32_t rc_shift = 2 * (k - 1);
  uint64_t prev = 0; uint32_t run = 0, applied = 0;
#pragma unroll 1
  for(int j = 0; j <= kPerLane; ++j) {          // the extra iteration only closes the last run
    uint64_t key = 0;
    bool have = false;
    if(j < kPerLane) {
      const uint64_t c = (L.cur >> (2 * (15 - j))) & 3u;
      fw = ((fw << 2) | c) & g.key_mask;
      rc = (rc >> 2) | ((3ull - c) << rc_shift);
      if(((L.inv48 >> (15 - j)) & kwin) == 0) {
        key = (g.canonical && rc < fw) ? rc : fw;
        have = !BLOOM || bloom_admits(T.bloom, key);
      }
      if(!have) continue;
      if(run && key == prev) { ++run; continue; }
    }
    if(run > 1) { table_add<RETURNING>(T, s_fwd, prev, run); ++applied; }
    prev = key; run = have ? 1 : 0;
  }
  return applied;
}

// ---- P1 ------------------------------------------------------------------------------
// SCATTER == false: histogram of this block's k-mers per bucket -> M[blockIdx][*]
// SCATTER == true : replay, items written at bucket_off[j] + M[blockIdx][j] + running count
// FROM_KEYS       : input is an array of encoded k-mers (hash_counter::add batches / the
//                   receive side of the multi-GPU exchange) instead of a contract buffer
// Runs of >= 2 identical consecutive k-mers in one lane (homopolymers, tandem repeats) bypass
// the partition and go straight to the table with one atomic per run (scatter pass only).
template <typename ITEM, bool SCATTER, bool FROM_KEYS, bool RETURNING, bool BLOOM, int NB = 0>
__global__ __launch_bounds__(kPBlock) void p1_kernel(DevTable T, PartGeom P, const uint8_t* __restrict__ base, int64_t lo,
                                                     int64_t hi, uint32_t* __restrict__ M,
                                                     const uint64_t* __restrict__ bucket_off, ITEM* __restrict__ out) {
  __shared__ uint64_t s_fwd[8 * 256];
  __shared__ uint32_t s_codes[kPBlock + 2];
  __shared__ uint32_t s_inv[kPBlock + 2];
  __shared__ unsigned long long s_cur[kMaxBuckets];
  const uint32_t nb = 1u << P.b1;
  load_tables_lds(s_fwd, T.fwd_tbl, T.g.nbytes);
  for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x)
    s_cur[j] = SCATTER ? (bucket_off[j] + M[(size_t)blockIdx.x * nb + j]) : 0ull;
  uint32_t my_mers = 0, my_direct = 0;
  const uint32_t bshift = T.g.lsize_l - P.b1;

  auto emit = [&](uint64_t key) {
    const uint64_t pos = hash_tables_t<NB>(s_fwd, key, T.g.nbytes);
    const uint64_t local = pos & T.g.local_mask;
    if((uint32_t)(pos >> T.g.lsize_l) != T.g.shard_id) {   // not ours: never silently inserted
      if(SCATTER) atomicAdd((unsigned long long*)&T.counters[CTR_MISROUTED], 1ull);
      return;
    }
    const uint32_t b = P.b1 ? (uint32_t)(local >> bshift) : 0u;
    const unsigned long long at = atomicAdd(&s_cur[b], 1ull);
    if(SCATTER) out[at] = make_item<ITEM>(T.g, P, key, local);
  };

  if(FROM_KEYS) {
    const uint64_t* keys = reinterpret_cast<const uint64_t*>(base);
    const int64_t n = hi;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = (int64_t)blockIdx.x * per, b1e = b0 + per < n ? b0 + per : n;
    lds_barrier();
    for(int64_t i = b0 + threadIdx.x; i < b1e; i += blockDim.x) { emit(keys[i] & T.g.key_mask); ++my_mers; }
  } else {
    const int64_t n_tiles = (hi + kPTilePos - 1) / kPTilePos;
    for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
      lds_barrier();
      const LaneWords L = stage_tile(base, tile * kPTilePos, lo, hi, s_codes, s_inv);
      uint64_t prev = 0; uint32_t run = 0; bool long_runs = false;
      auto flush_run = [&]() {
        if(run == 1) emit(prev);
        else if(run > 1) long_runs = true;
      };
      const uint32_t adm = BLOOM ? bloom_admit_mask(T.bloom, T.g, L) : 0xFFFFu;   // count --bc; read-only, so both passes agree
      for_each_kmer(T.g, L, [&](int j, uint64_t key) {
        ++my_mers;
        if(BLOOM && !((adm >> j) & 1u)) return;
        if(run && key == prev) { ++run; return; }
        flush_run();
        prev = key; run = 1;
      });
      flush_run();
      if(SCATTER && long_runs) my_direct += apply_runs<RETURNING, BLOOM>(T, s_fwd, L);
    }
  }
  lds_barrier();
  if(!SCATTER) {
    for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) M[(size_t)blockIdx.x * nb + j] = (uint32_t)s_cur[j];
    uint64_t w = my_mers;
    for(int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
    if((threadIdx.x & 63) == 0 && w) atomicAdd((unsigned long long*)&T.counters[CTR_MERS], (unsigned long long)w);
  } else if(my_direct) {
    atomicAdd((unsigned long long*)&T.counters[CTR_DIRECT], (unsigned long long)my_direct);
  }
}

// ---- histogram matrix -> private cursors ------------------------------------------------
// One block per group (P1: the single group; P2: one group per P1 bucket).  Group q owns the
// sub-matrix M[q][nblk][nb] (row = block).  Every bucket column becomes its exclusive prefix
// over the blocks, and off[q * nb + j] = base[q] + exclusive prefix of the column totals, so
// `off` is one global offset array with n_groups * nb + 1 entries (the last written by the
// last group).  base == nullptr means 0.
__global__ __launch_bounds__(1024) void scan_matrix_kernel(uint32_t* __restrict__ M, uint32_t nblk, uint32_t nb,
                                                           const uint64_t* __restrict__ base, uint64_t* __restrict__ off, uint32_t q0) {
  __shared__ unsigned long long s_tot[kMaxBuckets];
  __shared__ unsigned long long s_wave[16];
  const uint32_t q = q0 + blockIdx.x;
  uint32_t* Mq = M + (size_t)q * nblk * nb;
  for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) {
    uint64_t run = 0;
#pragma unroll 8
    for(uint32_t b = 0; b < nblk; ++b) {
      const uint32_t v = Mq[(size_t)b * nb + j];
      Mq[(size_t)b * nb + j] = (uint32_t)run;
      run += v;
    }
    s_tot[j] = run;
  }
  for(uint32_t j = nb + threadIdx.x; j < kMaxBuckets; j += blockDim.x) s_tot[j] = 0;
  lds_barrier();
  // exclusive scan of s_tot[0 .. 2048): two adjacent entries per thread, wave scan, then wave totals
  const uint32_t t = threadIdx.x;
  const unsigned long long a = s_tot[2 * t], b = s_tot[2 * t + 1];
  unsigned long long incl = a + b;
  for(int o = 1; o < 64; o <<= 1) {
    const unsigned long long up = __shfl_up(incl, o, 64);
    if((int)(t & 63) >= o) incl += up;
  }
  if((t & 63) == 63) s_wave[t >> 6] = incl;
  lds_barrier();
  unsigned long long wbase = 0;
  for(uint32_t w = 0; w < (t >> 6); ++w) wbase += s_wave[w];
  const unsigned long long excl = wbase + incl - (a + b);
  const unsigned long long g0 = base ? base[q] : 0ull;
  if(2 * t < nb) off[(size_t)q * nb + 2 * t] = g0 + excl;
  if(2 * t + 1 < nb) off[(size_t)q * nb + 2 * t + 1] = g0 + excl + a;
  // the entry after this group's last bucket: the end of the whole array, or the same value the next group
  // will write as its first entry (base[q+1] = base[q] + this total), so a consumer of group q alone is complete
  if(t == 1023) off[(size_t)(q + 1) * nb] = g0 + excl + a + b;
}

// ---- P2: every P1 bucket -> its tiles, one launch -------------------------------------------
// grid = (G2, 2^b1): blockIdx.y is the P1 bucket, blockIdx.x owns a contiguous slice of that
// bucket's items (the concatenation over pending batches of
// items[s][off[s][bucket] .. off[s][bucket+1]); same slice in both passes).
// M is [bucket][G2][2^b2]; goff the global per-tile offsets produced by scan_matrix_kernel.
template <typename ITEM, bool SCATTER>
__global__ __launch_bounds__(kPBlock) void p2_kernel(PartGeom P, uint32_t tag_bits, SegList S, uint32_t* __restrict__ M,
                                                     const uint64_t* __restrict__ goff, ITEM* __restrict__ out, uint32_t bucket0) {
  __shared__ unsigned long long s_cur[kMaxBuckets];
  __shared__ unsigned long long s_seg_lo[kMaxSeg + 1];
  const uint32_t nb = 1u << P.b2;
  const uint32_t bucket = bucket0 + blockIdx.y;
  uint32_t* Mq = M + ((size_t)bucket * gridDim.x + blockIdx.x) * nb;
  for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x)
    s_cur[j] = SCATTER ? (goff[(size_t)bucket * nb + j] + Mq[j]) : 0ull;
  if(threadIdx.x == 0) {
    unsigned long long c = 0;
    for(uint32_t s = 0; s < S.n; ++s) { s_seg_lo[s] = c; c += seg_hi(S, s, bucket) - seg_lo(S, s, bucket); }
    s_seg_lo[S.n] = c;
  }
  lds_barrier();
  const uint64_t n = s_seg_lo[S.n];
  const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint64_t my_lo = (uint64_t)blockIdx.x * per, my_hi = my_lo + per < n ? my_lo + per : n;
  for(uint32_t s = 0; s < S.n; ++s) {
    const uint64_t slo = s_seg_lo[s], shi = s_seg_lo[s + 1];
    const uint64_t a = my_lo > slo ? my_lo : slo, b = my_hi < shi ? my_hi : shi;
    if(a >= b) continue;
    const ITEM* src = reinterpret_cast<const ITEM*>(S.items[s]) + seg_lo(S, s, bucket) - slo;
    const bool holes = S.sh[s] != 0;                      // block-uniform
    constexpr int U = sizeof(ITEM) >= 16 ? 4 : 8;          // loads in flight per lane
    for(uint64_t v0 = a + threadIdx.x; v0 < b; v0 += (uint64_t)blockDim.x * U) {
      ITEM x[U];
#pragma unroll
      for(int r = 0; r < U; ++r) { const uint64_t v = v0 + (uint64_t)r * blockDim.x; x[r] = v < b ? src[v] : (ITEM)0; }
#pragma unroll
      for(int r = 0; r < U; ++r) {
        if(v0 + (uint64_t)r * blockDim.x >= b) break;
        const ITEM it = x[r];
        if(holes && it == (ITEM)~(ITEM)0) continue;
        const uint32_t d = (uint32_t)(it >> tag_bits) & (nb - 1);
        const unsigned long long at = atomicAdd(&s_cur[d], 1ull);
        if(SCATTER) out[at] = it;
      }
    }
  }
  if(!SCATTER) {
    lds_barrier();
    for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) Mq[j] = (uint32_t)s_cur[j];
  }
}

// Block-wide exclusive scan of nb (<= 2048) uint32 counters in LDS: in[] -> out[].
// 1024 threads, two adjacent entries per thread, wave prefix by __shfl_up, 16 wave totals.
__device__ inline void block_excl_scan_2048(const uint32_t* in, uint32_t* out, uint32_t nb, uint32_t* s_wave /*[16]*/) {
  const uint32_t t = threadIdx.x;
  const uint32_t a = 2 * t < nb ? in[2 * t] : 0u, b = 2 * t + 1 < nb ? in[2 * t + 1] : 0u;
  uint32_t incl = a + b;
  for(int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(incl, o, 64);
    if((int)(t & 63) >= o) incl += up;
  }
  if((t & 63) == 63) s_wave[t >> 6] = incl;
  lds_barrier();
  uint32_t wbase = 0;
  for(uint32_t w = 0; w < (t >> 6); ++w) wbase += s_wave[w];
  const uint32_t excl = wbase + incl - (a + b);
  if(2 * t < nb) out[2 * t] = excl;
  if(2 * t + 1 < nb) out[2 * t + 1] = excl + a;
}

// P1 scatter with write combining (32-bit items, contract-buffer input).  One block iteration =
// 16384 sequence positions = one chunk: every lane keeps its <= 17 emitted items in registers,
// the block counting-sorts them by bucket in LDS and writes whole runs.  Same tile->block
// assignment as the count pass, so the per-(block, bucket) cursors derived from M are exact.
template <bool RETURNING, bool BLOOM, int NB = 0>
__global__ __launch_bounds__(kPBlock) void p1_scatter_sorted_kernel(DevTable T, PartGeom P, const uint8_t* __restrict__ base,
                                                                    int64_t lo, int64_t hi, const uint32_t* __restrict__ M,
                                                                    const uint64_t* __restrict__ bucket_off,
                                                                    uint32_t* __restrict__ out) {
  JF_DYN_LDS(s_dyn);
  uint32_t* s_item = reinterpret_cast<uint32_t*>(s_dyn);                          // [kPTilePos]
  uint16_t* s_bkt = reinterpret_cast<uint16_t*>(s_dyn + (size_t)kPTilePos * 4);   // [kPTilePos]
  __shared__ uint64_t s_fwd[8 * 256];
  __shared__ uint32_t s_codes[kPBlock + 2];
  __shared__ uint32_t s_inv[kPBlock + 2];
  __shared__ unsigned long long s_gcur[kMaxBuckets];
  __shared__ uint32_t s_hist[kMaxBuckets];
  __shared__ uint32_t s_lstart[kMaxBuckets];
  __shared__ uint32_t s_wave[16];
  const uint32_t nb = 1u << P.b1;
  load_tables_lds(s_fwd, T.fwd_tbl, T.g.nbytes);
  for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) s_gcur[j] = bucket_off[j] + M[(size_t)blockIdx.x * nb + j];
  const uint32_t bshift = T.g.lsize_l - P.b1;
  uint32_t my_direct = 0;
  const int64_t n_tiles = (hi + kPTilePos - 1) / kPTilePos;
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    lds_barrier();
    for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) s_hist[j] = 0;
    const LaneWords L = stage_tile(base, tile * kPTilePos, lo, hi, s_codes, s_inv);   // barrier inside
    uint32_t it[kPerLane + 1], dr[kPerLane + 1];
#pragma unroll
    for(int e = 0; e <= kPerLane; ++e) dr[e] = 0xFFFFFFFFu;
    uint64_t prev = 0; uint32_t run = 0; bool long_runs = false;
    auto flush_run = [&](int site) {
      if(run == 1) {
        const uint64_t pos = hash_tables_t<NB>(s_fwd, prev, T.g.nbytes);
        const uint64_t local = pos & T.g.local_mask;
        const uint32_t b = P.b1 ? (uint32_t)(local >> bshift) : 0u;
        it[site] = make_item<uint32_t>(T.g, P, prev, local);
        dr[site] = (b << 16) | atomicAdd(&s_hist[b], 1u);
      } else if(run > 1) long_runs = true;
    };
    const uint32_t adm = BLOOM ? bloom_admit_mask(T.bloom, T.g, L) : 0xFFFFu;
    for_each_kmer(T.g, L, [&](int j, uint64_t key) {
      if(BLOOM && !((adm >> j) & 1u)) return;
      if(run && key == prev) { ++run; return; }
      flush_run(j);
      prev = key; run = 1;
    });
    flush_run(kPerLane);
    if(long_runs) my_direct += apply_runs<RETURNING, BLOOM>(T, s_fwd, L);
    lds_barrier();
    block_excl_scan_2048(s_hist, s_lstart, nb, s_wave);
    lds_barrier();
#pragma unroll
    for(int e = 0; e <= kPerLane; ++e)
      if(dr[e] != 0xFFFFFFFFu) {
        const uint32_t at = s_lstart[dr[e] >> 16] + (dr[e] & 0xFFFFu);
        s_item[at] = it[e]; s_bkt[at] = (uint16_t)(dr[e] >> 16);
      }
    lds_barrier();
    const uint32_t cn = s_lstart[nb - 1] + s_hist[nb - 1];
    for(uint32_t i = threadIdx.x; i < cn; i += blockDim.x) {
      const uint32_t b = s_bkt[i];
      out[s_gcur[b] + (i - s_lstart[b])] = s_item[i];
    }
    lds_barrier();
    for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) s_gcur[j] += s_hist[j];
  }
  if(my_direct) atomicAdd((unsigned long long*)&T.counters[CTR_DIRECT], (unsigned long long)my_direct);
}

// The same write-combining scatter for a batch of encoded k-mers (hash_counter::add batches and the
// receive side of the multi-GPU exchange).  Block b owns the contiguous slice the count pass
// (p1_kernel<.., false, true, ..>) gave it and walks it in chunks of 16384 keys.
template <int NB = 0>
__global__ __launch_bounds__(kPBlock) void p1_keys_scatter_sorted_kernel(DevTable T, PartGeom P, const uint64_t* __restrict__ keys,
                                                                         int64_t n, const uint32_t* __restrict__ M,
                                                                         const uint64_t* __restrict__ bucket_off,
                                                                         uint32_t* __restrict__ out) {
  JF_DYN_LDS(s_dyn);
  uint32_t* s_item = reinterpret_cast<uint32_t*>(s_dyn);                          // [kPTilePos]
  uint16_t* s_bkt = reinterpret_cast<uint16_t*>(s_dyn + (size_t)kPTilePos * 4);   // [kPTilePos]
  __shared__ uint64_t s_fwd[8 * 256];
  __shared__ unsigned long long s_gcur[kMaxBuckets];
  __shared__ uint32_t s_hist[kMaxBuckets];
  __shared__ uint32_t s_lstart[kMaxBuckets];
  __shared__ uint32_t s_wave[16];
  const uint32_t nb = 1u << P.b1;
  load_tables_lds(s_fwd, T.fwd_tbl, T.g.nbytes);
  for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) s_gcur[j] = bucket_off[j] + M[(size_t)blockIdx.x * nb + j];
  const uint32_t bshift = T.g.lsize_l - P.b1;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = (int64_t)blockIdx.x * per, b1e = b0 + per < n ? b0 + per : n;
  for(int64_t c0 = b0; c0 < b1e; c0 += kPTilePos) {
    lds_barrier();
    for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) s_hist[j] = 0;
    uint64_t kk[kPerLane];
#pragma unroll
    for(int e = 0; e < kPerLane; ++e) {
      const int64_t i = c0 + (int64_t)e * kPBlock + threadIdx.x;
      kk[e] = i < b1e ? keys[i] : 0;
    }
    lds_barrier();
    uint32_t it[kPerLane], dr[kPerLane];
#pragma unroll
    for(int e = 0; e < kPerLane; ++e) {
      dr[e] = 0xFFFFFFFFu;
      if(c0 + (int64_t)e * kPBlock + threadIdx.x < b1e) {
        const uint64_t key = kk[e] & T.g.key_mask;
        const uint64_t pos = hash_tables_t<NB>(s_fwd, key, T.g.nbytes);
        if((uint32_t)(pos >> T.g.lsize_l) != T.g.shard_id) { atomicAdd((unsigned long long*)&T.counters[CTR_MISROUTED], 1ull); continue; }
        const uint64_t local = pos & T.g.local_mask;
        const uint32_t b = P.b1 ? (uint32_t)(local >> bshift) : 0u;
        it[e] = make_item<uint32_t>(T.g, P, key, local);
        dr[e] = (b << 16) | atomicAdd(&s_hist[b], 1u);
      }
    }
    lds_barrier();
    block_excl_scan_2048(s_hist, s_lstart, nb, s_wave);
    lds_barrier();
#pragma unroll
    for(int e = 0; e < kPerLane; ++e)
      if(dr[e] != 0xFFFFFFFFu) {
        const uint32_t at = s_lstart[dr[e] >> 16] + (dr[e] & 0xFFFFu);
        s_item[at] = it[e]; s_bkt[at] = (uint16_t)(dr[e] >> 16);
      }
    lds_barrier();
    const uint32_t cn = s_lstart[nb - 1] + s_hist[nb - 1];
    for(uint32_t i = threadIdx.x; i < cn; i += blockDim.x) {
      const uint32_t b = s_bkt[i];
      out[s_gcur[b] + (i - s_lstart[b])] = s_item[i];
    }
    lds_barrier();
    for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) s_gcur[j] += s_hist[j];
  }
}

// P2 scatter with write combining: scattered 4-byte stores top out at ~50-100 G items/s on
// MI355X (one L2 transaction each), so every block first counting-sorts a chunk of kChunk
// items by destination bucket in LDS and then writes whole runs (consecutive lanes ->
// consecutive addresses).  Same block->slice assignment and per-(block, bucket) cursors as
// p2_kernel<.., false> counted, so positions are exact and no global atomic is needed.
constexpr int kP2MidPer = 14;       // ... of 8-byte items (single tiles): 14 Ki items = 112 KiB
constexpr int kP2PairPer = 28;      // items per lane and chunk when P2 routes to pairs of tiles (28 Ki items = 112 KiB of LDS; 32 would spill registers)
template <typename ITEM, int PER_THREAD>
__global__ __launch_bounds__(kPBlock) void p2_scatter_sorted_kernel(PartGeom P, uint32_t tag_bits, SegList S,
                                                                    const uint32_t* __restrict__ M,
                                                                    const uint64_t* __restrict__ goff, ITEM* __restrict__ out, uint32_t bucket0) {
  constexpr int kChunk = kPBlock * PER_THREAD;
  JF_DYN_LDS(s_dyn);
  ITEM* s_item = reinterpret_cast<ITEM*>(s_dyn);                       // [kChunk]
  __shared__ uint32_t s_delta[kMaxBuckets];     // write cursor of sub-bucket d (relative to the bucket's first item) minus its start in the sorted chunk, mod 2^32
  __shared__ uint32_t s_hist[kMaxBuckets];
  __shared__ uint32_t s_lstart[kMaxBuckets];
  __shared__ uint32_t s_wave[16];
  const uint32_t nb = 1u << P.b2;
  const uint32_t bucket = bucket0 + blockIdx.y;
  const uint32_t* Mq = M + ((size_t)bucket * gridDim.x + blockIdx.x) * nb;
  const uint64_t base0 = goff[(size_t)bucket * nb];     // a bucket holds < 2^32 items (checked by the host)
  for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) { s_delta[j] = (uint32_t)(goff[(size_t)bucket * nb + j] - base0) + Mq[j]; s_lstart[j] = 0; s_hist[j] = 0; }
  // this bucket's items = concatenation over the pending batches (block-uniform scalars)
  uint64_t n = 0;
  for(uint32_t s = 0; s < S.n; ++s) n += seg_hi(S, s, bucket) - seg_lo(S, s, bucket);
  const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint64_t my_lo = (uint64_t)blockIdx.x * per, my_hi = my_lo + per < n ? my_lo + per : n;
  [[maybe_unused]] PhaseClk pc;
  // One chunk's items into registers (vm: which of the lane's PER_THREAD positions hold an item, hm: which of those came
  // from a batch with holes).  No use of the loaded values here, so all the loads of a chunk are in flight together --
  // and the next chunk's are issued before the current one is sorted, so their latency hides behind the LDS work.
  // Addresses are a block-uniform base (the chunk's first item inside the batch) plus a 32-bit lane offset, so the loads
  // of a chunk share one offset register instead of carrying a 64-bit address each.
  auto load_chunk = [&](uint64_t c0, ITEM (&it)[PER_THREAD], uint32_t& vm, uint32_t& hm) {
    vm = 0; hm = 0;
#pragma unroll
    for(int r = 0; r < PER_THREAD; ++r) it[r] = 0;
    if(c0 >= my_hi) return;                             // block-uniform
    const uint32_t cn = my_hi - c0 < (uint64_t)kChunk ? (uint32_t)(my_hi - c0) : (uint32_t)kChunk;
    uint64_t slo = 0;
    for(uint32_t s = 0; s < S.n; ++s) {                 // uniform loop: usually one or two batches overlap a chunk
      const uint64_t o0 = seg_lo(S, s, bucket), len = seg_hi(S, s, bucket) - o0, shi = slo + len;
      if(shi > c0 && slo < c0 + cn) {
        const uint32_t lo_rel = slo > c0 ? (uint32_t)(slo - c0) : 0u;                 // the batch's part of this chunk,
        const uint32_t hi_rel = shi < c0 + cn ? (uint32_t)(shi - c0) : cn;            // in chunk-relative positions
        const ITEM* src = reinterpret_cast<const ITEM*>(S.items[s]) + (int64_t)o0 + ((int64_t)c0 - (int64_t)slo);
        const bool holes = S.sh[s] != 0;
#pragma unroll
        for(int r = 0; r < PER_THREAD; ++r) {
          const uint32_t rel = (uint32_t)r * kPBlock + threadIdx.x;
          if(rel >= lo_rel && rel < hi_rel) { it[r] = src[rel]; vm |= 1u << r; if(holes) hm |= 1u << r; }
        }
      }
      slo = shi;
    }
  };
  ITEM nxt[PER_THREAD]; uint32_t nvm = 0, nhm = 0;
  load_chunk(my_lo, nxt, nvm, nhm);
  for(uint64_t c0 = my_lo; c0 < my_hi; c0 += kChunk) {
    lds_barrier();                                      // previous chunk's readers of s_hist/s_lstart/s_item are done
    JF_PHASE(pc, 0);
    // the cursor of bucket d advances by what the previous chunk wrote; delta is re-based on the new lstart below
    for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) { s_delta[j] += s_hist[j] + s_lstart[j]; s_hist[j] = 0; }
    ITEM it[PER_THREAD];
    uint32_t rk[(PER_THREAD + 1) / 2];                  // rank inside the chunk's bucket, 16 bits each (a chunk holds <= 32768 items)
    const uint32_t hm = nhm;
    uint32_t vm = nvm;
#pragma unroll
    for(int r = 0; r < PER_THREAD; ++r) it[r] = nxt[r];
    load_chunk(c0 + kChunk, nxt, nvm, nhm);
    lds_barrier();
    JF_PHASE(pc, 1);
#pragma unroll
    for(int r = 0; r < PER_THREAD; ++r) {
      uint32_t rank = 0;
      if((vm >> r) & 1) {
        if(((hm >> r) & 1) && it[r] == (ITEM)~(ITEM)0) vm &= ~(1u << r);   // a hole
        else rank = atomicAdd(&s_hist[(uint32_t)(it[r] >> tag_bits) & (nb - 1)], 1u);
      }
      if(r & 1) rk[r >> 1] |= rank << 16; else rk[r >> 1] = rank;
    }
    lds_barrier();
    JF_PHASE(pc, 2);
    block_excl_scan_2048(s_hist, s_lstart, nb, s_wave);
    lds_barrier();
    JF_PHASE(pc, 3);
    for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) s_delta[j] -= s_lstart[j];
#pragma unroll
    for(int r = 0; r < PER_THREAD; ++r)
      if((vm >> r) & 1) s_item[s_lstart[(uint32_t)(it[r] >> tag_bits) & (nb - 1)] + ((rk[r >> 1] >> ((r & 1) * 16)) & 0xFFFFu)] = it[r];
    lds_barrier();
    JF_PHASE(pc, 4);
    const uint32_t cn = s_lstart[nb - 1] + s_hist[nb - 1];       // items of this chunk that are not holes
    for(uint32_t i = threadIdx.x; i < cn; i += blockDim.x) {
      const ITEM v = s_item[i];
      const uint32_t d = (uint32_t)(v >> tag_bits) & (nb - 1);
      out[base0 + (uint32_t)(s_delta[d] + i)] = v;     // run of bucket d: consecutive lanes, consecutive addresses
    }
    JF_PHASE(pc, 5);
  }
  JF_PHASE_FLUSH(pc, 8);
}

// One item of P1 bucket `bucket` straight into the table with global atomics (same protocol as
// table_add; tag and tile are already in the item).
template <bool RETURNING>
__device__ inline void item_direct_insert(const DevTable& T, const PartGeom& P, uint32_t bucket, uint64_t it) {
  const TableGeom& g = T.g;
  const uint32_t tmask = (uint32_t)g.tile_mask;
  const uint64_t tag = it & (g.occ_bit - 1);
  const uint64_t tile = ((uint64_t)bucket << P.b2) | ((it >> g.tag_bits) & ((1ull << P.b2) - 1));
  const uint64_t tile_base = tile << g.tile_bits;
  const uint32_t idx0 = (uint32_t)(tag >> g.rem_bits);
  { uint8_t* d = &T.dirty[tile]; if(!*d) *d = 1; }
  const uint64_t low = g.occ_bit | tag, neww = g.inc | low;
  for(uint32_t p = 0; p <= T.max_probe; ++p) {
    const uint64_t slot = tile_base + probe_lin(idx0, p, tmask);
    const uint64_t old = slot_cas(T, slot, 0, neww);
    if(old == 0ull) return;
    if((old & g.low_mask) == low) {
      if(RETURNING) {
        const uint64_t prev = slot_add_rtn(T, slot, g.inc);
        if((prev >> (g.tag_bits + 1)) + 1 > g.cnt_max) ovf_add(T, slot, 1);
      } else {
        slot_add(T, slot, g.inc);
      }
      return;
    }
  }
  atomicAdd((unsigned long long*)&T.counters[CTR_FULL], 1ull);
}

// Fallback when a flush holds too few items to be worth streaming the tiles: insert the
// pending items with global atomics.  cap == 0: packed batch (off[nb] items, bucket by binary
// search); cap > 0: granule batch, bucket j occupies [j * cap, off[2j+1]) and holes are skipped.
template <typename ITEM, bool RETURNING>
__global__ __launch_bounds__(kBlock) void items_direct_kernel(DevTable T, PartGeom P, const ITEM* __restrict__ items,
                                                              const uint64_t* __restrict__ off, uint64_t cap) {
  const uint32_t nb = 1u << P.b1;
  const uint64_t n = cap ? (uint64_t)nb * cap : off[nb];
  for(uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t lo = 0;
    if(cap) {
      lo = (uint32_t)(v / cap);
      if(v >= off[2 * (size_t)lo + 1]) continue;
    } else {                                  // bucket of item v: largest j with off[j] <= v
      uint32_t hi = nb;
      while(hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if(off[mid] <= v) lo = mid; else hi = mid; }
    }
    const ITEM it = items[v];
    if(cap && it == (ITEM)~(ITEM)0) continue;
    item_direct_insert<RETURNING>(T, P, lo, (uint64_t)it);
  }
}

// ---- P1 in one pass (32-bit items, contract-buffer input, two-level tables) -------------------
// The count pass of the exact scheme costs a second encode+hash of the whole batch.  Here bucket j
// of a batch owns a fixed region of `cap` items and blocks reserve space in it kGran items at a
// time (one global atomicAdd per reservation: ~1 per 64 items), so placement needs no histogram
// matrix.  What a block leaves unused in its last reservation of a bucket is filled with the
// all-ones item (a hole the consumers skip); a real all-ones item, and anything that does not fit
// the region any more (skewed input), goes straight to the table with global atomics.
//   gcur[j]  reservations handed out in bucket j (in items, multiples of kGran), zeroed per batch
//   tot[j]   exact number of items stored for bucket j (the flush sizes P2's output from it)
constexpr int kGranMaxB = 1024;             // buckets of the single-pass kernels (one per thread)

// Per-block state of the single-pass scatter, in LDS, shared by the contract-buffer and the key-array
// kernels.  emit(): given this chunk's per-bucket histogram (hist) and every lane's items (it) with
// bucket << 16 | rank (dr; 0xFFFFFFFF = none), place the chunk in the bucket regions and write it out.
struct GranuleLds {
  uint32_t cur[kGranMaxB];      // next free position of this block in bucket b's region (relative)
  uint16_t room[kGranMaxB];     // items left in the current reservation
  uint32_t hist[kGranMaxB];
  uint32_t lstart[kGranMaxB];
  uint4 place[kGranMaxB];       // this chunk's run of bucket b, in terms of the item's index i in the sorted chunk:
                                //   x: i < x goes to region position i + y (the current reservation), the rest to i + z (a new one)
                                //   w != 0: there is no new one (region exhausted), the rest is inserted directly
  uint32_t cnt[kGranMaxB];      // items this block stored per bucket
  uint32_t wave[16];
};
constexpr uint32_t kNoRoom = 0xFFFFFFFFu;

__device__ inline void granule_init(GranuleLds& G, uint32_t nb) {
  for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) { G.cur[j] = 0; G.room[j] = 0; G.cnt[j] = 0; }
}

// direct(b, item): what to do with an item that cannot be stored in bucket b's region (region exhausted, or the item
// equals the hole marker) -- the count path inserts it with global atomics, the Bloom path bumps its cell.
// ITEM: uint32_t (one-word keys, Bloom cell updates) or unsigned __int128 (two-word keys); the hole marker is all ones.
// gshort[b]: the overflow note of bucket b (see granule_finish_kernel).  bkt_of(i, v): bucket of the i-th item of the
// sorted chunk -- read back from s_bkt (P1: the bucket is not part of the item) or recomputed from the item (P2).
// scatter_fn(): the caller's items into the sorted chunk s_item (and s_bkt), item of bucket b with rank r at G.lstart[b] + r.
template <typename ITEM, typename DIRECT, typename BKTOF, typename SCATTER>
__device__ inline uint32_t granule_emit_x(GranuleLds& G, uint32_t nb, uint32_t cap, unsigned int* __restrict__ gcur, unsigned int* __restrict__ gshort,
                                          ITEM* __restrict__ out, const ITEM* s_item,
                                          SCATTER&& scatter_fn, DIRECT&& direct_fn, BKTOF&& bkt_of, PhaseClk* pc = nullptr) {
  lds_barrier();
  if(pc) JF_PHASE(*pc, 2);
  block_excl_scan_2048(G.hist, G.lstart, nb, G.wave);
  lds_barrier();
  if(pc) JF_PHASE(*pc, 3);
  // Placement of every bucket's run (nb <= blockDim: one bucket per thread): what fits the current
  // reservation stays there, the rest goes to a new one.  The reservation (a global atomic) is issued
  // first, its round trip overlaps the LDS scatter, its answer is used afterwards.
  const uint32_t pb = threadIdx.x;
  uint32_t ph = 0, proom = 0, pcur = 0, need = 0, take = 0, g0 = 0;
  if(pb < nb) {
    ph = G.hist[pb]; proom = G.room[pb]; pcur = G.cur[pb];
    if(ph > proom) { need = ph - proom; take = (need + kGran - 1) / kGran * kGran; g0 = atomicAdd(&gcur[pb], take); }
  }
  scatter_fn();
  if(pb < nb) {
    const uint32_t ls = G.lstart[pb];
    if(!take) { G.place[pb] = make_uint4(ls + ph, pcur - ls, 0u, 0u); G.cur[pb] = pcur + ph; G.room[pb] = (uint16_t)(proom - ph); G.cnt[pb] += ph; }
    else if((uint64_t)g0 + take <= cap) {
      G.place[pb] = make_uint4(ls + proom, pcur - ls, g0 - ls - proom, 0u);
      G.cur[pb] = g0 + need; G.room[pb] = (uint16_t)(take - need); G.cnt[pb] += ph;
    } else {                                                                                 // region exhausted
      G.place[pb] = make_uint4(ls + proom, pcur - ls, 0u, 1u);
      G.cur[pb] = pcur + proom; G.room[pb] = 0; G.cnt[pb] += proom;
      if(g0 < cap) atomicMax(&gshort[pb], cap - g0);          // everything below g0 was handed out successfully
    }
  }
  lds_barrier();
  if(pc) JF_PHASE(*pc, 4);
  uint32_t direct_n = 0;
  const uint32_t cn = G.lstart[nb - 1] + G.hist[nb - 1];
  const ITEM hole = (ITEM)~(ITEM)0;
  for(uint32_t i = threadIdx.x; i < cn; i += blockDim.x) {
    const ITEM v = s_item[i];
    const uint32_t b = bkt_of(i, v);
    const uint4 w = G.place[b];
    uint32_t rel = i + w.y;
    bool direct = false;
    if(i >= w.x) { if(w.w) direct = true; else rel = i + w.z; }
    if(!direct) {
      out[(uint64_t)b * cap + rel] = v;
      if(v == hole) { direct = true; atomicSub(&G.cnt[b], 1u); }   // its slot now reads as a hole
    }
    if(direct) { direct_fn(b, v); ++direct_n; }
  }
  if(pc) JF_PHASE(*pc, 5);
  return direct_n;
}

template <typename ITEM, int N, typename DIRECT>
__device__ inline uint32_t granule_emit(GranuleLds& G, uint32_t nb, uint32_t cap,
                                        unsigned int* __restrict__ gcur, ITEM* __restrict__ out, ITEM* s_item, uint16_t* s_bkt,
                                        const ITEM (&it)[N], const uint32_t (&dr)[N], DIRECT&& direct_fn, PhaseClk* pc = nullptr) {
  return granule_emit_x<ITEM>(G, nb, cap, gcur, gcur + nb, out, s_item,
                              [&]() {
#pragma unroll
                                for(int e = 0; e < N; ++e)
                                  if(dr[e] != 0xFFFFFFFFu) {
                                    const uint32_t at = G.lstart[dr[e] >> 16] + (dr[e] & 0xFFFFu);
                                    s_item[at] = it[e]; s_bkt[at] = (uint16_t)(dr[e] >> 16);
                                  }
                              },
                              direct_fn, [&](uint32_t i, ITEM) -> uint32_t { return s_bkt[i]; }, pc);
}

// Kernel end: unused tails of the last reservations become holes, exact per-bucket counts go to tot.
template <typename ITEM>
__device__ inline void granule_finish(GranuleLds& G, uint32_t nb, uint32_t cap, unsigned long long* __restrict__ tot, ITEM* __restrict__ out) {
  lds_barrier();
  for(uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
    const uint32_t room = G.room[b], cur = G.cur[b];
    for(uint32_t r = 0; r < room; ++r) out[(uint64_t)b * cap + cur + r] = (ITEM)~(ITEM)0;
    if(tot && G.cnt[b]) atomicAdd(&tot[b], (unsigned long long)G.cnt[b]);
  }
}

// (32-bit items from sequence: kernels_p1ring.hip.hpp -- per-bucket rings in LDS instead of a sort per chunk.)

// Single-pass P1 with 64-bit items (one-word keys whose item does not fit 32 bits, e.g. k = 31 at 2^33 slots): the
// block's 16384 positions go through the sort in two rounds of 8 positions per lane (8192 items of 8 bytes = 64 KiB of
// LDS per round).  Consecutive identical k-mers are not merged here: they meet in the LDS tile like any other duplicate.
constexpr int kG64Per = 8;
constexpr int kG64Chunk = kPBlock * kG64Per;
template <bool RETURNING, bool BLOOM, int NB>
__global__ __launch_bounds__(kPBlock) void p1_granule64_kernel(DevTable T, PartGeom P, const uint8_t* __restrict__ base,
                                                               int64_t lo, int64_t hi, uint32_t cap,
                                                               unsigned int* __restrict__ gcur, unsigned long long* __restrict__ tot,
                                                               uint64_t* __restrict__ out) {
  JF_DYN_LDS(s_dyn);
  uint64_t* s_item = reinterpret_cast<uint64_t*>(s_dyn);                               // [kG64Chunk]
  uint16_t* s_bkt = reinterpret_cast<uint16_t*>(s_dyn + (size_t)kG64Chunk * 8);        // [kG64Chunk]
  __shared__ uint64_t s_fwd[NB < 0 ? 1 : 8 * 256];                                      // (NB < 0: the xor-shift matrix, evaluated in registers)
  __shared__ uint32_t s_codes[kPBlock + 2];
  __shared__ uint32_t s_inv[kPBlock + 2];
  __shared__ GranuleLds G;
  const TableGeom& g = T.g;
  const uint32_t nb = 1u << P.b1;
  if constexpr(NB >= 0) load_tables_lds(s_fwd, T.fwd_tbl, g.nbytes);
  granule_init(G, nb);
  const uint32_t k = g.k, bshift = g.lsize_l - P.b1;
  const uint64_t kwin = k >= 64 ? ~0ull : ((1ull << k) - 1);
  const uint32_t rc_shift = 2 * (k - 1);
  uint32_t my_direct = 0, my_mers = 0;
  const int64_t n_tiles = (hi + kPTilePos - 1) / kPTilePos;
  TileRaw R = tile_fetch(base, (int64_t)blockIdx.x * kPTilePos, lo, hi);
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    lds_barrier();
    const LaneWords L = tile_stage(R, tile * kPTilePos, lo, hi, s_codes, s_inv);      // barrier inside
    R = tile_fetch(base, (tile + gridDim.x) * kPTilePos, lo, hi);
    const uint32_t adm = BLOOM ? bloom_admit_mask(T.bloom, g, L) : 0xFFFFu;
    uint64_t fw = (((uint64_t)L.p2 << 32) | L.p1) & g.key_mask;
    uint64_t rc = revcomp64(fw, k);
#pragma unroll 1
    for(int j0 = 0; j0 < kPerLane; j0 += kG64Per) {
      lds_barrier();
      for(uint32_t q = threadIdx.x; q < nb; q += blockDim.x) G.hist[q] = 0;
      lds_barrier();
      uint64_t it[kG64Per]; uint32_t dr[kG64Per];
#pragma unroll
      for(int e = 0; e < kG64Per; ++e) {
        const int j = j0 + e;
        dr[e] = 0xFFFFFFFFu; it[e] = 0;
        const uint64_t c = (L.cur >> (2 * (15 - j))) & 3u;
        fw = ((fw << 2) | c) & g.key_mask;
        rc = (rc >> 2) | ((3ull - c) << rc_shift);
        if(((L.inv48 >> (15 - j)) & kwin) == 0) {
          ++my_mers;
          if(!BLOOM || ((adm >> j) & 1u)) {
            const uint64_t key = (g.canonical && rc < fw) ? rc : fw;
            const uint64_t pos = NB < 0 ? xs_hash(key, g.lsize_g, g.key_bits) : hash_tables_t<(NB < 0 ? 0 : NB)>(s_fwd, key, g.nbytes);
            const uint64_t local = pos & g.local_mask;
            it[e] = make_item<uint64_t>(g, P, key, local);
            const uint32_t b = (uint32_t)(local >> bshift);
            dr[e] = (b << 16) | atomicAdd(&G.hist[b], 1u);
          }
        }
      }
      my_direct += granule_emit(G, nb, cap, gcur, out, s_item, s_bkt, it, dr,
                                [&](uint32_t b, uint64_t v) { item_direct_insert<RETURNING>(T, P, b, v); });
    }
  }
  granule_finish(G, nb, cap, tot, out);
  if(my_direct) atomicAdd((unsigned long long*)&T.counters[CTR_DIRECT], (unsigned long long)my_direct);
  uint64_t w = my_mers;
  for(int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
  if((threadIdx.x & 63) == 0 && w) atomicAdd((unsigned long long*)&T.counters[CTR_MERS], (unsigned long long)w);
}

// The same for a batch of encoded k-mers (hash_counter::add batches, the receive side of the multi-GPU
// exchange): block b walks its contiguous slice in chunks of 16384 keys.
template <bool RETURNING, int NB>
__global__ __launch_bounds__(kPBlock) void p1_keys_granule_kernel(DevTable T, PartGeom P, const uint64_t* __restrict__ keys, int64_t n,
                                                                  uint32_t cap, unsigned int* __restrict__ gcur,
                                                                  unsigned long long* __restrict__ tot, uint32_t* __restrict__ out) {
  JF_DYN_LDS(s_dyn);
  uint32_t* s_item = reinterpret_cast<uint32_t*>(s_dyn);
  uint16_t* s_bkt = reinterpret_cast<uint16_t*>(s_dyn + (size_t)kPTilePos * 4);
  __shared__ uint64_t s_fwd[8 * 256];
  __shared__ GranuleLds G;
  const uint32_t nb = 1u << P.b1;
  load_tables_lds(s_fwd, T.fwd_tbl, T.g.nbytes);
  granule_init(G, nb);
  const uint32_t bshift = T.g.lsize_l - P.b1;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = (int64_t)blockIdx.x * per, b1e = b0 + per < n ? b0 + per : n;
  uint32_t my_direct = 0, misrouted = 0;
  if(threadIdx.x == 0 && b1e > b0) atomicAdd((unsigned long long*)&T.counters[CTR_MERS], (unsigned long long)(b1e - b0));
  uint64_t kk[kPerLane];
  auto fetch = [&](int64_t c0) {
#pragma unroll
    for(int e = 0; e < kPerLane; ++e) {
      const int64_t i = c0 + (int64_t)e * kPBlock + threadIdx.x;
      kk[e] = i < b1e ? keys[i] : 0;
    }
  };
  fetch(b0);
  for(int64_t c0 = b0; c0 < b1e; c0 += kPTilePos) {
    lds_barrier();
    for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) G.hist[j] = 0;
    lds_barrier();
    uint32_t it[kPerLane], dr[kPerLane];
#pragma unroll
    for(int e = 0; e < kPerLane; ++e) {
      dr[e] = 0xFFFFFFFFu;
      if(c0 + (int64_t)e * kPBlock + threadIdx.x < b1e) {
        const uint64_t key = kk[e] & T.g.key_mask;
        const uint64_t pos = hash_tables_t<NB>(s_fwd, key, T.g.nbytes);
        if((uint32_t)(pos >> T.g.lsize_l) != T.g.shard_id) { ++misrouted; continue; }
        const uint64_t local = pos & T.g.local_mask;
        const uint32_t b = (uint32_t)(local >> bshift);
        it[e] = make_item<uint32_t>(T.g, P, key, local);
        dr[e] = (b << 16) | atomicAdd(&G.hist[b], 1u);
      }
    }
    fetch(c0 + kPTilePos);                                 // next chunk's keys travel during the sort and the write-out
    my_direct += granule_emit(G, nb, cap, gcur, out, s_item, s_bkt, it, dr,
                              [&](uint32_t b, uint32_t v) { item_direct_insert<RETURNING>(T, P, b, (uint64_t)v); });
  }
  granule_finish(G, nb, cap, tot, out);
  if(my_direct) atomicAdd((unsigned long long*)&T.counters[CTR_DIRECT], (unsigned long long)my_direct);
  if(misrouted) atomicAdd((unsigned long long*)&T.counters[CTR_MISROUTED], (unsigned long long)misrouted);
}

// ---- P2 in one pass (32-bit items routed to pairs of tiles) ------------------------------------------------------
// The count pass of the exact P2 reads every item once more just to size the destinations.  Here every destination (a pair
// of tiles) owns a fixed region of `cap` items and blocks reserve space in it kGran items at a time, exactly like the
// single-pass P1: same LDS counting sort per chunk, same holes, and what does not fit a region goes straight to the table.
// grid = (G2, buckets): blockIdx.y is the P1 bucket, blockIdx.x a contiguous slice of its items.  gcur / gshort / out are
// indexed by destination = bucket * 2^b2e + sub-bucket.
// What to do with an item that cannot be stored in its region: one-word keys (32-bit items) -- the table's global claim.
// The rare ways out of the partition (an item that would read as a hole, a ring or a region that is full, a run of
// identical consecutive k-mers) as ONE real function: inlined at every site, item_direct_insert keeps a few dozen scalars
// of the table alive through the whole hot loop (the kernel then spills scalar registers) and multiplies the code.
// T: the table's descriptor in device memory (taking the address of a kernel's by-value copy would put it in scratch).
__device__ __attribute__((noinline)) void item_direct_call(const DevTable* T, uint32_t b2, uint32_t b, uint64_t item, uint32_t cnt, int returning) {
  const TableGeom& g = T->g;
  const uint32_t tmask = (uint32_t)g.tile_mask;
  const uint64_t it = item;
  const uint64_t tag = it & (g.occ_bit - 1);
  const uint64_t tile = ((uint64_t)b << b2) | ((it >> g.tag_bits) & ((1ull << b2) - 1));
  const uint64_t tile_base = tile << g.tile_bits;
  const uint32_t idx0 = (uint32_t)(tag >> g.rem_bits);
  { uint8_t* d = &T->dirty[tile]; if(!*d) *d = 1; }
  const uint64_t low = g.occ_bit | tag, add = (uint64_t)cnt << (g.tag_bits + 1), neww = add | low;
  for(uint32_t p = 0; p <= T->max_probe; ++p) {
    const uint64_t slot = tile_base + probe_lin(idx0, p, tmask);
    const uint64_t old = slot_cas(*T, slot, 0, neww);
    if(old == 0ull) return;
    if((old & g.low_mask) == low) {
      if(returning) {
        const uint64_t prev = slot_add_rtn(*T, slot, add);
        if((prev >> (g.tag_bits + 1)) + cnt > g.cnt_max) ovf_add(*T, slot, 1);
      } else slot_add(*T, slot, add);
      return;
    }
  }
  atomicAdd((unsigned long long*)&T->counters[CTR_FULL], 1ull);
}

// What to do with an item that cannot be stored in its region: one-word keys -- the table's global claim, out of line,
// reading the table's descriptor from device memory (Tm).
template <bool RETURNING>
struct TableDirect {
  const DevTable* Tm; PartGeom P; unsigned long long* ctr_direct;
  __device__ void operator()(uint32_t bucket, uint64_t item) const { item_direct_call(Tm, P.b2, bucket, item, 1, RETURNING ? 1 : 0); }
  __device__ unsigned long long* direct_counter() const { return ctr_direct; }
};

// ITEM: uint32_t or unsigned __int128; DIRECT: see TableDirect (kernels_wide_part.hip.hpp has the two-word one).
// (Requesting the next chunk into a second set of registers before the current one is sorted was measured in round 4: no
// gain, 29.5 ms either way on the metric's job -- the five barriers of a chunk are the cost, not the loads' latency.)
template <typename ITEM, typename DIRECT, int PER_THREAD, int SMALL = 0>
__global__ __launch_bounds__(kPBlock) void p2_granule_kernel(DIRECT D, uint32_t b2e, uint32_t tag_bits, SegList S, uint32_t cap,
                                                             unsigned int* __restrict__ gcur, unsigned int* __restrict__ gshort,
                                                             ITEM* __restrict__ out, uint32_t bucket0,
                                                             unsigned long long* __restrict__ tot = nullptr, uint32_t bucket_mask = 0xFFFFFFFFu) {
  // tot (optional): exact items stored per destination.  bucket_mask: the part of the bucket index that P, the geometry
  // of the direct inserts, knows about (the receive side of the multi-GPU exchange splits buckets numbered globally).
  constexpr int kChunk = kPBlock * PER_THREAD;
  JF_DYN_LDS(s_dyn);
  ITEM* s_item = reinterpret_cast<ITEM*>(s_dyn);                       // [kChunk]
  __shared__ GranuleLds G;
  const uint32_t nb = 1u << b2e;
  const uint32_t bucket = bucket0 + blockIdx.y;
  unsigned int* gc = gcur + (size_t)bucket * nb;
  unsigned int* gs = gshort + (size_t)bucket * nb;
  ITEM* o = out + (size_t)bucket * nb * cap;
  granule_init(G, nb);
  uint64_t n = 0;
  for(uint32_t s = 0; s < S.n; ++s) n += seg_hi(S, s, bucket) - seg_lo(S, s, bucket);
  const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint64_t my_lo = (uint64_t)blockIdx.x * per, my_hi = my_lo + per < n ? my_lo + per : n;
  uint32_t my_direct = 0;
  [[maybe_unused]] PhaseClk pc;
  // one chunk's items into registers (vm: which of the lane's PER_THREAD positions hold an item, hm: which of those came
  // from a batch with holes); no use of the loaded values here, so all the loads of a chunk are in flight together
  auto load_chunk = [&](uint64_t c0, ITEM (&it)[PER_THREAD], uint32_t& vm, uint32_t& hm) {
    vm = 0; hm = 0;
#pragma unroll
    for(int r = 0; r < PER_THREAD; ++r) it[r] = 0;
    if(c0 >= my_hi) return;                             // block-uniform
    const uint32_t cn = my_hi - c0 < (uint64_t)kChunk ? (uint32_t)(my_hi - c0) : (uint32_t)kChunk;
    uint64_t slo = 0;
    for(uint32_t s = 0; s < S.n; ++s) {                 // uniform loop: usually one or two batches overlap a chunk
      const uint64_t o0 = seg_lo(S, s, bucket), len = seg_hi(S, s, bucket) - o0, shi = slo + len;
      if(shi > c0 && slo < c0 + cn) {
        const uint32_t lo_rel = slo > c0 ? (uint32_t)(slo - c0) : 0u;
        const uint32_t hi_rel = shi < c0 + cn ? (uint32_t)(shi - c0) : cn;
        const ITEM* src = reinterpret_cast<const ITEM*>(S.items[s]) + (int64_t)o0 + ((int64_t)c0 - (int64_t)slo);
        const bool holes = S.sh[s] != 0;
#pragma unroll
        for(int r = 0; r < PER_THREAD; ++r) {
          const uint32_t rel = (uint32_t)r * kPBlock + threadIdx.x;
          if(rel >= lo_rel && rel < hi_rel) { it[r] = src[rel]; vm |= 1u << r; if(holes) hm |= 1u << r; }
        }
      }
      slo = shi;
    }
  };
  // 16-byte items (two-word keys): the phase clocks of round 5 showed this kernel waiting for its chunk's loads 65 % of the
  // time (seven 16-byte loads a lane, nothing else to do meanwhile).  The NEXT chunk is therefore requested into a second
  // set of registers before the current one is ranked -- with unconditional loads when the chunk lies inside one batch's
  // region (nearly always), so that the compiler can count them instead of waiting on the spot.  (For 4-byte items the
  // same was measured in round 4 without gain: there the five barriers of a chunk are the cost.)
#ifndef JFGPU_PF_MIN
#define JFGPU_PF_MIN 16
#endif
  constexpr bool PF = sizeof(ITEM) >= JFGPU_PF_MIN;
  ITEM nx[PF ? PER_THREAD : 1];
  [[maybe_unused]] uint32_t nvm = 0, nhm = 0;
  auto load_chunk_pf = [&](uint64_t c0, ITEM (&x)[PER_THREAD], uint32_t& vm, uint32_t& hm) {
    if(c0 < my_hi && my_hi - c0 >= (uint64_t)kChunk) {
      uint64_t slo = 0;
      for(uint32_t s = 0; s < S.n; ++s) {
        const uint64_t o0 = seg_lo(S, s, bucket), shi = slo + (seg_hi(S, s, bucket) - o0);
        if(c0 >= slo && c0 + kChunk <= shi) {                  // (block-uniform)
          const ITEM* src = reinterpret_cast<const ITEM*>(S.items[s]) + (int64_t)o0 + ((int64_t)c0 - (int64_t)slo);
#pragma unroll
          for(int r = 0; r < PER_THREAD; ++r) x[r] = src[(uint32_t)r * kPBlock + threadIdx.x];
          vm = (1u << PER_THREAD) - 1u; hm = S.sh[s] != 0 ? vm : 0u;
          return;
        }
        slo = shi;
      }
    }
    load_chunk(c0, x, vm, hm);
  };
  if constexpr(PF) load_chunk_pf(my_lo, nx, nvm, nhm);
  for(uint64_t c0 = my_lo; c0 < my_hi; c0 += kChunk) {
    lds_barrier();                                      // previous chunk's readers are done
    JF_PHASE(pc, 0);
    for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) G.hist[j] = 0;
    ITEM it[PER_THREAD];
    uint32_t rk[(PER_THREAD + 1) / 2];                  // rank inside the chunk's bucket, 16 bits each
    uint32_t hm = 0, vm = 0;
    if constexpr(PF) {
#pragma unroll
      for(int r = 0; r < PER_THREAD; ++r) it[r] = nx[r];
      vm = nvm; hm = nhm;
      load_chunk_pf(c0 + kChunk, nx, nvm, nhm);
    } else load_chunk(c0, it, vm, hm);
    lds_barrier();
    JF_PHASE(pc, 1);
    if constexpr(SMALL != 0) {
      // (SMALL: nb <= 4 * SMALL <= 16) a fan-out of a few destinations (the receive side of the multi-GPU exchange).  Thousands
      // of lanes on a handful of LDS counters would serialise, and ranking lanes per destination with ballots costs
      // 10 instructions per item AND destination.  Instead every lane counts its own items per destination in packed
      // 16-bit fields (four per 64-bit word), one block-wide prefix sum of the packed words gives each lane its first
      // rank per destination, and a second sweep hands the ranks out: a dozen instructions per item, whatever nb.
      constexpr int SW = SMALL;
      __shared__ unsigned long long s_wtot[kPBlock / 64][4];
      const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      unsigned long long cnt[SW];
#pragma unroll
      for(int w = 0; w < SW; ++w) cnt[w] = 0;
#pragma unroll
      for(int r = 0; r < PER_THREAD; ++r) {
        if(((vm >> r) & 1) && ((hm >> r) & 1) && it[r] == (ITEM)~(ITEM)0) vm &= ~(1u << r);   // a hole
        if((vm >> r) & 1) {
          const uint32_t d = (uint32_t)(it[r] >> tag_bits) & (nb - 1);
          const unsigned long long one = 1ull << ((d & 3) * 16);
#pragma unroll
          for(int w = 0; w < SW; ++w) cnt[w] += (SW == 1 || (int)(d >> 2) == w) ? one : 0ull;
        }
      }
      unsigned long long inc[SW];
#pragma unroll
      for(int w = 0; w < SW; ++w) inc[w] = cnt[w];
      for(int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for(int w = 0; w < SW; ++w) { const unsigned long long up = __shfl_up(inc[w], o, 64); if((int)lane >= o) inc[w] += up; }
      }
      if(lane == 63) {
#pragma unroll
        for(int w = 0; w < SW; ++w) s_wtot[wave][w] = inc[w];
      }
      lds_barrier();
      unsigned long long run[SW];
#pragma unroll
      for(int w = 0; w < SW; ++w) {
        unsigned long long before = 0;
        for(uint32_t v = 0; v < wave; ++v) before += s_wtot[v][w];
        run[w] = before + inc[w] - cnt[w];                   // this lane's first rank per destination
        if(threadIdx.x == kPBlock - 1) {                      // (the last lane's inclusive sums are the chunk's totals)
          const unsigned long long tot_w = before + inc[w];
#pragma unroll
          for(int f = 0; f < 4; ++f) if((uint32_t)(4 * w + f) < nb) G.hist[4 * w + f] = (uint32_t)(tot_w >> (16 * f)) & 0xFFFFu;
        }
      }
#pragma unroll
      for(int r = 0; r < PER_THREAD; ++r) {
        uint32_t rank = 0;
        if((vm >> r) & 1) {
          const uint32_t d = (uint32_t)(it[r] >> tag_bits) & (nb - 1);
          const uint32_t sh = (d & 3) * 16;
#pragma unroll
          for(int w = 0; w < SW; ++w)
            if(SW == 1 || (int)(d >> 2) == w) { rank = (uint32_t)(run[w] >> sh) & 0xFFFFu; run[w] += 1ull << sh; }
        }
        if(r & 1) rk[r >> 1] |= rank << 16; else rk[r >> 1] = rank;
      }
    } else {
#pragma unroll
      for(int r = 0; r < PER_THREAD; ++r) {
        uint32_t rank = 0;
        if((vm >> r) & 1) {
          if(((hm >> r) & 1) && it[r] == (ITEM)~(ITEM)0) vm &= ~(1u << r);   // a hole
          else rank = atomicAdd(&G.hist[(uint32_t)(it[r] >> tag_bits) & (nb - 1)], 1u);
        }
        if(r & 1) rk[r >> 1] |= rank << 16; else rk[r >> 1] = rank;
      }
    }
    my_direct += granule_emit_x<ITEM>(G, nb, cap, gc, gs, o, s_item,
                   [&]() {
#pragma unroll
                     for(int r = 0; r < PER_THREAD; ++r)
                       if((vm >> r) & 1) s_item[G.lstart[(uint32_t)(it[r] >> tag_bits) & (nb - 1)] + ((rk[r >> 1] >> ((r & 1) * 16)) & 0xFFFFu)] = it[r];
                   },
                   [&](uint32_t, ITEM v) { D(bucket & bucket_mask, v); },
                   [&](uint32_t, ITEM v) -> uint32_t { return (uint32_t)(v >> tag_bits) & (nb - 1); }, &pc);
  }
  granule_finish<ITEM>(G, nb, cap, tot ? tot + (size_t)bucket * nb : nullptr, o);
  JF_PHASE(pc, 6);
  JF_PHASE_FLUSH(pc, 8);
  if(my_direct) atomicAdd(D.direct_counter(), (unsigned long long)my_direct);
}

// ---- multi-GPU: the receive split (abi_comm.inl: comm_insert_prev_items) ------------------------------------------------------
// A coarse bucket of a shard arrives as W regions (one per sender) and is split into the shard's own P1 buckets: a
// fan-out of F = 2^LF <= 16.  The sort-based p2_granule_kernel<.., SMALL> did that in round 4 at 25.5 ms per 10 Gbp (five
// barriers a chunk, two workgroups a CU: 2.7 TB/s for a pass that only moves 35 GB in and 35 GB out).  Here every WAVE is
// on its own -- no workgroup barrier anywhere: it streams its share of the regions 256 items at a time (16 bytes a lane,
// three requests ahead), appends each item to its destination's ring in LDS (8 KB of rings per wave; the rank is a
// ballot count for F <= 2, an LDS add for more) and, whenever a ring holds a row of kSplitRow(LF) items, writes the row
// to the destination's region (up to 16 bytes a lane).  The kernel is bound by its instructions, not by HBM (a first
// version with rows of 64 items for every F: 24.2 ms, with or without the requests ahead), hence the long rows.  Space in a
// region is reserved `res` items at a time (one global add per reservation, by the lane that keeps the destination's
// state, asked for when the previous reservation runs out and used a row or more later), so the output is what the
// granule kernels' is: regions of cap items, gcur = what was handed out, gshort = the overflow note, holes behind what
// was not filled, tot = items stored.  What does not fit its region goes to the table directly (D).
constexpr int kSplitWaves = 4;                // waves per workgroup
constexpr uint32_t split_row(uint32_t lf) { return (1024u >> lf) < 256u ? (1024u >> lf) : 256u; }      // items per row written: 256, 256, 256, 128, 64
template <typename DIRECT, int LF>
__global__ __launch_bounds__(64 * kSplitWaves) void recv_split_kernel(DIRECT D, uint32_t split_at, SegList S, uint32_t cap, uint32_t res,
                                                                      unsigned int* __restrict__ gcur, unsigned int* __restrict__ gshort,
                                                                      uint32_t* __restrict__ out, uint32_t bucket0,
                                                                      unsigned long long* __restrict__ tot, uint32_t bucket_mask) {
  constexpr uint32_t F = 1u << LF, RW = split_row(LF), RS = 2 * RW, V = RW / 64;      // ring of two rows per destination; V items a lane and row
  constexpr int kCheck = (int)(RW / 64);                         // rounds of 64 items between two looks at the rings' fill: <= RW - 1 + RW items then
  __shared__ __align__(16) uint32_t s_ring[kSplitWaves][F * RS];
  __shared__ uint32_t s_cnt[kSplitWaves][16];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
  const uint32_t bucket = bucket0 + blockIdx.y;
  uint32_t* const ring = s_ring[wave];
  uint32_t* const cnt = s_cnt[wave];
  unsigned int* const gc = gcur + (size_t)bucket * F;
  unsigned int* const gs = gshort + (size_t)bucket * F;
  uint32_t* const o = out + (size_t)bucket * F * cap;
  const uint32_t hole = 0xFFFFFFFFu;
  const uint32_t nw = gridDim.x * kSplitWaves, wid = blockIdx.x * kSplitWaves + wave;
  // lane d < F keeps destination d: items flushed, the reservation in use [pos, pos + room), the one asked for ahead
  uint32_t fl = 0, pos = 0, room = 0, nxt = 0, stored = 0, direct_n = 0;
  bool has_nxt = false, dead = false;
  uint32_t c0 = 0, c1 = 0;                                      // F <= 2: the rings' counts (wave-uniform)
  if(lane < 16) cnt[lane] = 0;
  (void)__ballot(true);
  auto request = [&]() {                                         // (called by lane d alone)
    if(!dead && !has_nxt) { nxt = atomicAdd(&gc[lane], res); has_nxt = true; }
  };
  // a row of destination d out of its ring (n < RW only at the end: the rest of the row becomes holes)
  auto flush = [&](uint32_t d, uint32_t n) {
    uint32_t fl_d = __shfl(fl, d, 64), room_d = __shfl(room, d, 64), pos_d = __shfl(pos, d, 64);
    if(room_d == 0) {                                            // (wave-uniform) take the reservation asked for ahead
      if(lane == d) {
        request();
        if(!dead) {
          if((uint64_t)nxt + res <= cap) { pos = nxt; room = res; }
          else { dead = true; if(nxt < cap) atomicMax(&gs[d], cap - nxt); }      // everything below nxt was handed out successfully
          has_nxt = false;
        }
      }
      room_d = __shfl(room, d, 64); pos_d = __shfl(pos, d, 64);
    }
    // (rows start at multiples of RW in a ring of 2 RW: contiguous, aligned)
    uint32_t v[V];
    const uint32_t* rp = ring + d * RS + (fl_d & (RS - 1)) + lane * V;
    if constexpr(V == 4) { const uint4 x = *reinterpret_cast<const uint4*>(rp); v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; }
    else if constexpr(V == 2) { const uint2 x = *reinterpret_cast<const uint2*>(rp); v[0] = x.x; v[1] = x.y; }
    else v[0] = rp[0];
#pragma unroll
    for(uint32_t j = 0; j < V; ++j) if(lane * V + j >= n) v[j] = hole;
    if(room_d) {
      uint32_t* op = o + (size_t)d * cap + pos_d + lane * V;
      if constexpr(V == 4) *reinterpret_cast<uint4*>(op) = make_uint4(v[0], v[1], v[2], v[3]);
      else if constexpr(V == 2) *reinterpret_cast<uint2*>(op) = make_uint2(v[0], v[1]);
      else op[0] = v[0];
      if(lane == d) { pos += RW; room -= RW; stored += n; fl += n; if(room == 0) request(); }
    } else {                                                     // region exhausted: straight to the table
#pragma unroll
      for(uint32_t j = 0; j < V; ++j) if(v[j] != hole) { D(bucket & bucket_mask, v[j]); ++direct_n; }
      if(lane == d) fl += n;
    }
  };
  for(uint32_t s = 0; s < S.n; ++s) {
    const uint64_t o0 = seg_lo(S, s, bucket), len = seg_hi(S, s, bucket) - o0;
    // this wave's part of the region: rows of 64 items dealt out evenly
    const uint64_t rows = (len + 63) >> 6, r_lo = rows * wid / nw, r_hi = rows * (wid + 1) / nw;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(S.items[s]) + o0;
    const uint64_t i_hi = r_hi << 6 < len ? r_hi << 6 : len;
    // lane l's four items of a request are 4 l .. 4 l + 3 of its 256: the order of a region's items means nothing
    constexpr int PD = 3;
    auto ld = [&](uint64_t i0) -> uint4 {
      const uint64_t i = i0 + 4 * lane;
      if(i >= i_hi) return make_uint4(hole, hole, hole, hole);
      uint4 x = *reinterpret_cast<const uint4*>(src + i);
      if(i + 1 >= i_hi) x.y = hole;
      if(i + 2 >= i_hi) x.z = hole;
      if(i + 3 >= i_hi) x.w = hole;
      return x;
    };
    uint4 q[PD];
#pragma unroll
    for(int k = 0; k < PD; ++k) q[k] = ld((r_lo << 6) + (uint64_t)k * 256);
    for(uint64_t i0 = r_lo << 6; i0 < i_hi; i0 += 256) {
      const uint4 cur = q[0];
#pragma unroll
      for(int k = 0; k + 1 < PD; ++k) q[k] = q[k + 1];
      q[PD - 1] = ld(i0 + (uint64_t)PD * 256);
      const uint32_t it[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
      for(int r = 0; r < 4; ++r) {
        const bool valid = it[r] != hole;
        const uint32_t d = (it[r] >> split_at) & (F - 1);
        uint32_t p = 0;
        if constexpr(F == 1) {
          const unsigned long long m = __ballot(valid);
          p = c0 + (uint32_t)__popcll(m & below); c0 += (uint32_t)__popcll(m);
        } else if constexpr(F == 2) {
          const unsigned long long m0 = __ballot(valid && d == 0), m1 = __ballot(valid && d == 1);
          p = d ? c1 + (uint32_t)__popcll(m1 & below) : c0 + (uint32_t)__popcll(m0 & below);
          c0 += (uint32_t)__popcll(m0); c1 += (uint32_t)__popcll(m1);
        } else if(valid) p = atomicAdd(&cnt[d], 1u);
        if(valid) ring[d * RS + (p & (RS - 1))] = it[r];
        if((r + 1) % kCheck != 0) continue;
        (void)__ballot(true);                                    // (written before read: lockstep on the device, a rendezvous in the host emulation)
        const uint32_t c = F <= 2 ? (lane == 0 ? c0 : c1) : (lane < F ? cnt[lane] : 0u);
        unsigned long long full = __ballot(lane < F && c - fl >= RW);
        while(full) {                                            // (wave-uniform) one row per destination at most: RW items came in since the last look
          const uint32_t d1 = (uint32_t)__ffsll((long long)full) - 1u;
          full &= full - 1;
          flush(d1, RW);
        }
        (void)__ballot(true);                                    // (read before the next round writes)
      }
    }
  }
  // what is left in the rings, then holes over what was reserved and not used
  {
    const uint32_t c = F <= 2 ? (lane == 0 ? c0 : c1) : (lane < F ? cnt[lane] : 0u);
    for(uint32_t d = 0; d < F; ++d) {
      const uint32_t left = __shfl(c, d, 64) - __shfl(fl, d, 64);
      if(left) flush(d, left);
      for(int pass = 0; pass < 2; ++pass) {                      // the reservation in use, then the one asked for and never used
        if(pass == 1 && lane == d) {
          pos = nxt; room = has_nxt && (uint64_t)nxt + res <= cap ? res : 0u;
          if(has_nxt && (uint64_t)nxt + res > cap && nxt < cap) atomicMax(&gs[d], cap - nxt);
        }
        const uint32_t room_d = __shfl(room, d, 64), pos_d = __shfl(pos, d, 64);
        for(uint32_t qq = lane; qq < room_d; qq += 64) o[(size_t)d * cap + pos_d + qq] = hole;
      }
    }
    if(lane < F && stored && tot) atomicAdd(&tot[(size_t)bucket * F + lane], (unsigned long long)stored);
    unsigned long long dn = direct_n;
    for(int off = 32; off > 0; off >>= 1) dn += __shfl_down(dn, off, 64);
    if(lane == 0 && dn) atomicAdd(D.direct_counter(), dn);
  }
}

// ---- multi-GPU: P1 on the sending side (abi_comm.inl) ------------------------------------------------------------
// The single-pass P1 over the GLOBAL table: T is a view of the shard's table whose geometry says "one table of 2^lsize_g
// slots" (same matrix, same tags), so bucket = top 10 bits of the global position = (owner rank, the owner's coarse
// bucket) and the regions of one owner are contiguous: they are what travels, 4 bytes per k-mer, already grouped for the
// receiver.  It is the count path's p1_ring_kernel with RouteListDirect (kernels_p1ring.hip.hpp; round 3's sort-based
// p1_route_granule_kernel: 48 ms per 10 Gbp where the ring kernel takes 35): nothing may be inserted here (the k-mers
// belong to other GPUs), so what the local P1 would insert directly -- items that do not fit their region, the item that
// looks like a hole, runs of identical k-mers -- is appended to a list of stragglers (bucket << 32 | item) every rank
// receives.
struct StragList { unsigned long long* n; uint64_t* rec; uint32_t cap; uint32_t pad_; };

// The stragglers a rank received: those of its own buckets go straight to the table.  cbits: bits of the coarse bucket
// index inside an owner; P: the geometry item_direct_insert needs for (coarse bucket, item).
template <bool RETURNING>
__global__ __launch_bounds__(kBlock) void straggler_insert_kernel(DevTable T, PartGeom P, const uint64_t* __restrict__ rec,
                                                                  const unsigned long long* __restrict__ n_ptr, uint32_t cap,
                                                                  uint32_t owner, uint32_t cbits, unsigned long long* __restrict__ kept = nullptr) {
  const unsigned long long n = *n_ptr < cap ? *n_ptr : cap;
  uint32_t mine = 0;                                             // kept: how many of the list were this owner's (the receiver's count of what arrived)
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = rec[i];
    const uint32_t gb = (uint32_t)(r >> 32);
    if((gb >> cbits) == owner) { item_direct_insert<RETURNING>(T, P, gb & ((1u << cbits) - 1), r & 0xFFFFFFFFull); ++mine; }
  }
  if(kept && mine) atomicAdd(kept, (unsigned long long)mine);
}

// After the granule pass: bucket bounds in the pair format of SegList (sh == 1).
__global__ void granule_finish_kernel(const unsigned int* __restrict__ gcur, uint32_t cap, uint32_t nb, uint64_t* __restrict__ off2) {
  for(uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nb; j += gridDim.x * blockDim.x) {
    const uint32_t shortfall = gcur[nb + j];                     // > 0: some reservation did not fit
    uint64_t used = gcur[j];
    if(shortfall) used = shortfall <= cap ? cap - shortfall : 0;
    else if(used > cap) used = cap;
    off2[2 * (size_t)j] = (uint64_t)j * cap;
    off2[2 * (size_t)j + 1] = (uint64_t)j * cap + used;
  }
}

// The same for destinations d0 .. d0 + nd of nb (a flush in groups).
__global__ void granule_finish_range_kernel(const unsigned int* __restrict__ gcur, uint32_t cap, uint32_t nb, uint64_t* __restrict__ off2,
                                            uint32_t d0, uint32_t nd) {
  for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nd; i += gridDim.x * blockDim.x) {
    const uint32_t j = d0 + i;
    const uint32_t shortfall = gcur[nb + j];
    uint64_t used = gcur[j];
    if(shortfall) used = shortfall <= cap ? cap - shortfall : 0;
    else if(used > cap) used = cap;
    off2[2 * (size_t)j] = (uint64_t)j * cap;
    off2[2 * (size_t)j + 1] = (uint64_t)j * cap + used;
  }
}

}  // namespace jfgpu
