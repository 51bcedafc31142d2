// jellyfish_amd/csrc/kernels_wide_part.hip.hpp -- the partitioned insert path for two-word keys (33 <= k <= 64,
// BASELINE config 5), gfx950.  Same three stages as kernels_part.hip.hpp -- no global atomics on the hot path:
//
//   P1w  p1_wide_granule_kernel: encode + canonical + GF(2) hash of 128-bit k-mers, 128-bit items counting-sorted by the
//        upper tile bits in LDS and written as whole runs into fixed bucket regions (granule_emit)
//   P2   the generic p2_kernel / scan_matrix_kernel / p2_scatter_sorted_kernel with ITEM = unsigned __int128
//   Tw   tile_insert_wide_kernel: one workgroup owns one tile of 8192 x 16 B = 128 KiB in LDS; the two-word claim of
//        kernels_wide.hip.hpp (CAS hi 0 -> occ|tag_hi, CAS lo 0 -> tag_lo|valid, add to hi) on LDS words
//
// item = (tile_rest << tag_full) | tag,  tag = (idx0 << rem_bits) | (key >> lsize): 2k - b1 bits in all.
// The table format is the one of kernels_wide.hip.hpp (reference: multi-word keys with a per-word "set" bit,
// /root/reference/include/jellyfish/large_hash_array.hpp:542-579, offsets_key_value.hpp:28-31), so look-ups, stats,
// growth and the sorted dump do not care which path inserted a key.
#pragma once
#include "kernels_part.hip.hpp"
#include "kernels_wide.hip.hpp"

namespace jfgpu {

constexpr int kWidePer = 4;                                  // items per lane per round of the P1 kernel
constexpr int kWideChunk = kPBlock * kWidePer;               // 4096 items = 64 KiB of LDS

__device__ inline u128 make_item_wide(const WideGeom& W, const PartGeom& P, u128 key, uint64_t local) {
  const u128 rest = (u128)(local & ((1ull << P.rest_shift) - 1));        // tile_rest (b2 bits) above idx0 (tile_bits)
  return (rest << W.g.rem_bits) | (key >> W.g.lsize_g);
}

// One item of P1 bucket `bucket` straight into the table (global two-word claim; same protocol as wide_add).
template <bool RETURNING>
__device__ inline void wide_item_direct(const WideTable& T, const PartGeom& P, uint32_t bucket, u128 it) {
  const TableGeom& g = T.W.g;
  const u128 tag = it & ((((u128)1) << T.W.tag_full) - 1);
  const uint64_t tile = ((uint64_t)bucket << P.b2) | ((uint64_t)(it >> T.W.tag_full) & ((1ull << P.b2) - 1));
  const uint64_t tile_base = tile << g.tile_bits;
  const uint32_t idx0 = (uint32_t)(tag >> g.rem_bits);
  const uint64_t wlo = (((uint64_t)tag) << 1) | 1ull, whi = g.occ_bit | (uint64_t)(tag >> 63);
  if(T.dirty) { uint8_t* d = &T.dirty[tile]; if(!*d) *d = 1; }
  for(uint32_t p = 0; p <= T.max_probe; ++p) {
    const uint64_t slot = tile_base + probe_slot(idx0, p, (uint32_t)g.tile_mask);
    unsigned long long* hi = (unsigned long long*)&T.slots[2 * slot + 1];
    unsigned long long* lo = (unsigned long long*)&T.slots[2 * slot];
    const unsigned long long old = atomicCAS(hi, 0ull, (unsigned long long)whi);
    if(old != 0ull && (old & g.low_mask) != whi) continue;
    const unsigned long long l = atomicCAS(lo, 0ull, (unsigned long long)wlo);
    if(l != 0ull && l != wlo) continue;
    if(RETURNING) {
      const unsigned long long prev = atomicAdd(hi, (unsigned long long)g.inc);
      if((prev >> (g.tag_bits + 1)) + 1 > g.cnt_max) { const DevTable d = ovf_view(T); ovf_add(d, slot, 1); }
    } else {
      __hip_atomic_fetch_add(hi, (unsigned long long)g.inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  atomicAdd((unsigned long long*)&T.counters[CTR_FULL], 1ull);
}

// p2_granule_kernel's overflow policy for two-word keys, and its chunk (7 Ki items = 112 KiB of LDS)
template <bool RETURNING>
struct WideDirect {
  WideTable T; PartGeom P;
  __device__ void operator()(uint32_t bucket, u128 item) const { wide_item_direct<RETURNING>(T, P, bucket, item); }
  __device__ unsigned long long* direct_counter() const { return (unsigned long long*)&T.counters[CTR_DIRECT]; }
};
constexpr int kP2WidePer = 7;

// ---- P1w --------------------------------------------------------------------------------------------------
// One block iteration = 16384 sequence positions, in rounds of kWidePer positions per lane (4096 items per round).
// XS: the table's matrix is the xor-shift one (kmer_core.hpp: xs_hash_wide): a dozen register instructions instead of
// sixteen 8-byte table reads per k-mer.
template <bool RETURNING, bool BLOOM, bool XS = false>
__global__ __launch_bounds__(kPBlock) void p1_wide_granule_kernel(WideTable T, PartGeom P, const uint8_t* __restrict__ base, int64_t lo, int64_t hi,
                                                                  uint32_t cap, unsigned int* __restrict__ gcur,
                                                                  unsigned long long* __restrict__ tot, u128* __restrict__ out) {
  JF_DYN_LDS(s_dyn);
  u128* s_item = reinterpret_cast<u128*>(s_dyn);                                       // [kWideChunk]
  uint16_t* s_bkt = reinterpret_cast<uint16_t*>(s_dyn + (size_t)kWideChunk * 16);      // [kWideChunk]
  uint64_t* s_fwd = reinterpret_cast<uint64_t*>(s_dyn + (size_t)kWideChunk * 18);      // [nbytes * 256]
  __shared__ uint32_t s_codes[kPBlock + 4];
  __shared__ uint32_t s_inv[kPBlock + 4];
  __shared__ GranuleLds G;
  const WideGeom& W = T.W;
  const TableGeom& g = W.g;
  const uint32_t nb = 1u << P.b1;
  if constexpr(!XS) load_tables_lds(s_fwd, T.fwd_tbl, g.nbytes);
  granule_init(G, nb);
  const uint32_t k = g.k, bshift = g.lsize_l - P.b1;
  const u128 kwin = (((u128)1) << k) - 1;
  const uint32_t rc_shift = 2 * (k - 1);
  uint32_t my_mers = 0, my_direct = 0;
  const int64_t n_tiles = (hi + kPTilePos - 1) / kPTilePos;
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();
    const LaneWordsW L = stage_tile_wide(base, tile * kPTilePos, lo, hi, s_codes, s_inv);   // barrier inside
    u128 fw = ((((u128)L.p4 << 96) | ((u128)L.p3 << 64) | ((u128)L.p2 << 32) | L.p1)) & W.key_mask;
    u128 rc = revcomp128(fw, k);
#pragma unroll 1
    for(int j0 = 0; j0 < kPerLane; j0 += kWidePer) {
      lds_barrier();
      for(uint32_t q = threadIdx.x; q < nb; q += blockDim.x) G.hist[q] = 0;
      lds_barrier();
      u128 it[kWidePer]; uint32_t dr[kWidePer];
#pragma unroll
      for(int e = 0; e < kWidePer; ++e) {
        const int j = j0 + e;
        dr[e] = 0xFFFFFFFFu; it[e] = 0;
        const uint64_t c = (L.cur >> (2 * (15 - j))) & 3u;
        fw = ((fw << 2) | c) & W.key_mask;
        rc = (rc >> 2) | ((u128)(3ull - c) << rc_shift);
        if(((L.inv80 >> (15 - j)) & kwin) == 0) {
          ++my_mers;
          const u128 key = (g.canonical && rc < fw) ? rc : fw;
          if(!BLOOM || bloom_admits_wide(T.bloom, key)) {
            const uint64_t pos = XS ? xs_hash_wide((uint64_t)key, (uint64_t)(key >> 64), g.lsize_g) : hash_tables_wide(s_fwd, key, g.nbytes);
            const uint64_t local = pos & g.local_mask;
            const uint32_t b = P.b1 ? (uint32_t)(local >> bshift) : 0u;
            it[e] = make_item_wide(W, P, key, local);
            dr[e] = (b << 16) | atomicAdd(&G.hist[b], 1u);
          }
        }
      }
      my_direct += granule_emit(G, nb, cap, gcur, out, s_item, s_bkt, it, dr,
                                [&](uint32_t b, u128 v) { wide_item_direct<RETURNING>(T, P, b, v); });
    }
  }
  granule_finish(G, nb, cap, tot, out);
  if(my_direct) atomicAdd((unsigned long long*)&T.counters[CTR_DIRECT], (unsigned long long)my_direct);
  uint64_t w = my_mers;
  for(int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
  if((threadIdx.x & 63) == 0 && w) atomicAdd((unsigned long long*)&T.counters[CTR_MERS], (unsigned long long)w);
}

// ---- Tw: one workgroup owns one 128 KiB tile in LDS -----------------------------------------------------------
template <bool RETURNING>
__device__ inline void tile_insert_wide_one(const WideTable& T, unsigned long long* s_tile, u128 item, uint64_t tile_index) {
  const TableGeom& g = T.W.g;
  const u128 tag = item & ((((u128)1) << T.W.tag_full) - 1);
  const uint32_t idx0 = (uint32_t)(tag >> g.rem_bits);
  const unsigned long long wlo = (((uint64_t)tag) << 1) | 1ull, whi = g.occ_bit | (uint64_t)(tag >> 63);
  const uint32_t tmask = (uint32_t)g.tile_mask;
  for(uint32_t p = 0; p <= T.max_probe; ++p) {
    const uint32_t slot = probe_slot(idx0, p, tmask);
    const unsigned long long old = atomicCAS(&s_tile[2 * slot + 1], 0ull, whi);      // (a plain look first, to pass foreign tags without an atomic, was measured: slower, 63.6 -> 71.3 ms)
    if(old != 0ull && (old & g.low_mask) != whi) continue;
    const unsigned long long l = atomicCAS(&s_tile[2 * slot], 0ull, wlo);
    if(l != 0ull && l != wlo) continue;
    if(RETURNING) {
      const unsigned long long prev = atomicAdd(&s_tile[2 * slot + 1], (unsigned long long)g.inc);
      if((prev >> (g.tag_bits + 1)) + 1 > g.cnt_max) { const DevTable d = ovf_view(T); ovf_add(d, (tile_index << g.tile_bits) + slot, 1); }
    } else {
      atomicAdd(&s_tile[2 * slot + 1], (unsigned long long)g.inc);
    }
    return;
  }
  atomicAdd((unsigned long long*)&T.counters[CTR_FULL], 1ull);
}

template <bool RETURNING>
__global__ __launch_bounds__(kPBlock) void tile_insert_wide_kernel(WideTable T, SegList S, uint64_t tile0, uint32_t n_tiles) {
  JF_DYN_LDS(s_raw);
  unsigned long long* s_tile = reinterpret_cast<unsigned long long*>(s_raw);
  const TableGeom& g = T.W.g;
  const uint32_t words = 2u << g.tile_bits;                            // 64-bit words of one tile
  for(uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    uint64_t n_items = 0;
    for(uint32_t s = 0; s < S.n; ++s) n_items += seg_hi(S, s, t) - seg_lo(S, s, t);
    if(n_items == 0) continue;                                          // block-uniform
    uint64_t* gt = T.slots + ((tile0 + t) << (g.tile_bits + 1));
    const bool load = T.dirty[tile0 + t] != 0;                          // block-uniform
    for(uint32_t i = threadIdx.x * 2; i < words; i += blockDim.x * 2) {
      ulonglong2 v = make_ulonglong2(0ull, 0ull);
      if(load) v = *reinterpret_cast<const ulonglong2*>(gt + i);
      *reinterpret_cast<ulonglong2*>(s_tile + i) = v;
    }
    lds_barrier();
    for(uint32_t s = 0; s < S.n; ++s) {
      const uint64_t a = seg_lo(S, s, t), b = seg_hi(S, s, t);
      const u128* src = reinterpret_cast<const u128*>(S.items[s]);
      const bool holes = S.sh[s] != 0;
      constexpr int U = 4;                                              // 16-byte loads in flight per lane
      for(uint64_t v0 = a + threadIdx.x; v0 < b; v0 += (uint64_t)U * blockDim.x) {
        u128 x[U];
#pragma unroll
        for(int u = 0; u < U; ++u) { const uint64_t v = v0 + (uint64_t)u * blockDim.x; x[u] = v < b ? src[v] : ~(u128)0; }
#pragma unroll
        for(int u = 0; u < U; ++u) {
          const uint64_t v = v0 + (uint64_t)u * blockDim.x;
          if(v < b && !(holes && x[u] == ~(u128)0)) tile_insert_wide_one<RETURNING>(T, s_tile, x[u], tile0 + t);
        }
      }
    }
    lds_barrier();
    for(uint32_t i = threadIdx.x * 2; i < words; i += blockDim.x * 2)
      *reinterpret_cast<ulonglong2*>(gt + i) = *reinterpret_cast<const ulonglong2*>(s_tile + i);
    if(threadIdx.x == 0) T.dirty[tile0 + t] = 1;
    lds_barrier();
  }
}

// ---- Tw, pipelined (round 5) ------------------------------------------------------------------------------------------------
// tile_insert_wide_kernel runs its phases one after the other on the one workgroup a CU has room for (128 KiB of LDS):
// zero the tile, fetch the items four at a time, claim them, store -- 18 us a tile at config 5's load (rocprofv3 r04: 75 ms
// for 94 GB of items and 137 GB of table).  Same claims here, but nothing waits for HBM inside a tile: a chunk's items (8 per
// lane) are requested as soon as the previous chunk's are claimed -- they travel during its store -- the offsets two units
// ahead, and the store zeroes the LDS tile behind itself, so a tile nothing was ever inserted into costs no fill pass:
// 75.6 -> 63.6 ms on config 5.  One item array (a flush's P2 output); flushes of single-level tables (several pending
// batches per tile) keep the kernel above.
template <bool RETURNING>
__global__ __launch_bounds__(kPBlock) void tile_insert_wide_pipe_kernel(WideTable T, SegList S, uint64_t tile0, uint32_t n_tiles) {
  constexpr int NPW = 8;                                                // items per lane and chunk
  constexpr uint64_t kRound = (uint64_t)NPW * kPBlock;
  JF_DYN_LDS(s_raw);
  unsigned long long* s_tile = reinterpret_cast<unsigned long long*>(s_raw);
  const TableGeom& g = T.W.g;
  const uint32_t words = 2u << g.tile_bits;                            // 64-bit words of one tile
  const uint64_t* off = S.off[0];
  const uint32_t sh = S.sh[0];                                         // 1: (begin, end) pairs, items may be holes
  const bool holes = sh != 0;
  const u128* src = reinterpret_cast<const u128*>(S.items[0]);
  const u128 hole = ~(u128)0;
  const uint32_t G = gridDim.x;
  struct Unit { uint64_t a, b; uint32_t d; };
  auto unit_of = [&](uint32_t t) -> Unit {
    Unit u{0, 0, 0};
    if(t < n_tiles) { u.a = off[(size_t)t << sh]; u.b = off[((size_t)t << sh) + 1]; u.d = T.dirty[tile0 + t]; }
    return u;
  };
  u128 cur[NPW];
  auto fetch = [&](uint64_t c, uint64_t b) {                           // items src[c .. min(b, c + kRound)): clamped indices, no branch per item
    if(b <= c) return;                                                 // (block-uniform)
    const uint32_t n = (uint32_t)((b - c) < kRound ? (b - c) : kRound);
    const u128* ub = src + c;
    uint32_t tid = threadIdx.x;
    JF_OPAQUE(tid);
#pragma unroll
    for(int r = 0; r < NPW; ++r) { const uint32_t i = (uint32_t)r * kPBlock + tid; cur[r] = ub[i < n ? i : n - 1]; }
  };
  for(uint32_t i = threadIdx.x * 2; i < words; i += blockDim.x * 2) *reinterpret_cast<ulonglong2*>(s_tile + i) = make_ulonglong2(0ull, 0ull);
  uint32_t t = blockIdx.x;
  Unit u0 = unit_of(t), u1 = unit_of(t + G);
  while(t < n_tiles && u0.b <= u0.a) { t += G; u0 = u1; u1 = unit_of(t + G); }      // (block-uniform) units without items
  uint64_t c0 = u0.a;
#pragma unroll
  for(int r = 0; r < NPW; ++r) cur[r] = hole;
  if(t < n_tiles) fetch(c0, u0.b);
  lds_barrier();
  [[maybe_unused]] PhaseClk pc;
  while(t < n_tiles) {
    const bool first = c0 == u0.a, last = c0 + kRound >= u0.b;
    // the chunk after this one: the unit's next round, or the first round of the next unit that has items
    uint32_t tn = t; Unit un = u0, un1 = u1; uint64_t cn = c0 + kRound;
    if(last) {
      tn = t + G; un = u1; un1 = unit_of(tn + G);
      while(tn < n_tiles && un.b <= un.a) { tn += G; un = un1; un1 = unit_of(tn + G); }
      cn = un.a;
    }
    uint64_t* gt = T.slots + ((tile0 + t) << (g.tile_bits + 1));
    // a further round of a unit starts from the tile as the previous round stored it
    if(first ? u0.d != 0 : true) {
      for(uint32_t i = threadIdx.x * 2; i < words; i += blockDim.x * 2)
        *reinterpret_cast<ulonglong2*>(s_tile + i) = *reinterpret_cast<const ulonglong2*>(gt + i);
      lds_barrier();
    }
    const uint32_t n0 = (uint32_t)((u0.b - c0) < kRound ? (u0.b - c0) : kRound);
    JF_PHASE(pc, 1);
    // The claims, one item after the other.  Measured and dropped in round 5 (profiles/r05_c5_tile_experiments.log): the first
    // probe of all eight items together + a queue for the rest (91.6 ms: the queue's items are read again from memory), all
    // probes of all items in lock step (154 ms: eight items x four predicated passes per step), a per-lane state machine
    // issuing one atomic a step (128 ms: the 128-bit tag arithmetic of every step), a plain look before each compare-and-
    // swap (71.3 ms).  The phase clocks say the claims are 48 % of this kernel and the waves' wait for each other at the
    // barrier behind them 39 %; what removes them is placement by rank (kernels_tile.hip.hpp), which needs the bucketed
    // probe sequence in every two-word kernel -- not built.
#pragma unroll
    for(int r = 0; r < NPW; ++r)
      if((uint32_t)r * kPBlock + threadIdx.x < n0 && !(holes && cur[r] == hole)) tile_insert_wide_one<RETURNING>(T, s_tile, cur[r], tile0 + t);
    JF_PHASE(pc, 2);
    // cur[] is dead: the next chunk's items travel during the store
    if(tn < n_tiles) fetch(cn, un.b);
    lds_barrier();
    JF_PHASE(pc, 3);
    for(uint32_t i = threadIdx.x * 2; i < words; i += blockDim.x * 2) {
      *reinterpret_cast<ulonglong2*>(gt + i) = *reinterpret_cast<const ulonglong2*>(s_tile + i);
      *reinterpret_cast<ulonglong2*>(s_tile + i) = make_ulonglong2(0ull, 0ull);            // the next unit starts from an empty tile
    }
    if(threadIdx.x == 0) T.dirty[tile0 + t] = 1;
    lds_barrier();
    JF_PHASE(pc, 4);
    t = tn; u0 = un; u1 = un1; c0 = cn;
  }
  JF_PHASE_FLUSH(pc, 16);
}

// Too few items to be worth streaming the tiles: pending granule batches inserted with global atomics.
template <bool RETURNING>
__global__ __launch_bounds__(kBlock) void items_direct_wide_kernel(WideTable T, PartGeom P, const u128* __restrict__ items,
                                                                   const uint64_t* __restrict__ off, uint64_t cap) {
  const uint32_t nb = 1u << P.b1;
  const uint64_t n = (uint64_t)nb * cap;
  for(uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t b = (uint32_t)(v / cap);
    if(v >= off[2 * (size_t)b + 1]) continue;
    const u128 it = items[v];
    if(it == ~(u128)0) continue;
    wide_item_direct<RETURNING>(T, P, b, it);
  }
}

}  // namespace jfgpu
