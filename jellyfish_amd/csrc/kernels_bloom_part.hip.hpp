// jellyfish_amd/csrc/kernels_bloom_part.hip.hpp -- the Bloom counter's insert pass without global atomics (gfx950).
//
// bloom_counter2_base::insert__ (/root/reference/include/jellyfish/bloom_counter2.hpp:56-107) bumps nb_hashes cells per
// k-mer, cell_i = (h0 % m + i * (h1 % m)) % m, each a base-3 digit of a byte.  On a 28 GB filter (config 3: m = 14e10)
// those are 10 random byte read-modify-writes per k-mer; random 64-byte HBM accesses retire at ~21 G/s on MI355X
// (profiles/r02_call1.log: the direct kernel, 2.1 G k-mers/s).  The final array does not depend on the order of the
// increments (they saturate at 2 and commute), so the same machinery as the count path applies: route every CELL
// UPDATE to the 64 KiB segment of the byte array it falls in, and apply a segment's updates in LDS.
//
//   P1b  p1_bloom_granule_kernel: encode + canonical + two 64-row GF(2) hashes, the nh cells of every k-mer as
//        32-bit items, counting-sorted by the upper segment bits in LDS and written as whole runs into fixed bucket
//        regions (the single-pass placement of kernels_part.hip.hpp: granule_emit)
//   P2   the count path's p2_kernel / scan_matrix_kernel / p2_scatter_sorted_kernel, unchanged (items carry their
//        sub-bucket at bit kBloomItemLow)
//   Tb   bloom_segment_kernel: one workgroup owns one segment in LDS: load, saturating digit bumps by 32-bit LDS
//        compare-and-swap, store
//
// item = (segment & (2^b2 - 1)) << 19 | byte offset in the segment (16 bits) << 3 | digit (p % 5)
#pragma once
#include "kernels_part.hip.hpp"
#include "kernels_bloom.hip.hpp"

namespace jfgpu {

constexpr uint32_t kBloomSegBits = 16;                       // a segment = 64 KiB of the byte array = 327 680 cells
constexpr uint32_t kBloomItemLow = kBloomSegBits + 3;        // bits of an item below the sub-bucket
constexpr int kBloomPer = 10;                                // items per lane per round of the P1 kernel
constexpr int kBloomChunk = kPBlock * kBloomPer;             // items sorted per round

struct BloomPart {
  uint32_t b1, b2;          // bucket bits of P1 / P2 (b2 == 0: P1 buckets are segments)
  uint32_t n_seg;           // segments holding data
  uint32_t pad_;
};

// One cell update straight into the filter (region exhausted): the item's cell, bumped with a global CAS.
__device__ inline void bloom_item_direct(const DevBloom& B, const BloomPart& BP, uint32_t bucket, uint32_t item) {
  const uint64_t seg = ((uint64_t)bucket << BP.b2) | (item >> kBloomItemLow);
  const uint64_t byte = (seg << kBloomSegBits) | ((item >> 3) & 0xFFFFu);
  bloom_bump(B.data, byte, item & 7u);
}

// p2_granule_kernel's overflow policy for cell updates: the cell is bumped in place.  counter: any device word (the
// number of such updates, not read by anyone yet).
struct BloomDirect {
  DevBloom B; BloomPart BP; unsigned long long* counter;
  __device__ void operator()(uint32_t bucket, uint32_t item) const { bloom_item_direct(B, BP, bucket, item); }
  __device__ unsigned long long* direct_counter() const { return counter; }
};

// The same for p2_ring_kernel (kernels_p1ring.hip.hpp), which speaks of destinations: destination = segment.  A real call
// with three scalars-worth of arguments: inlined into the ring kernel with the whole DevBloom, the rare path put 304 bytes
// of every lane in scratch and the stage ran at half its speed (535 ms against 246 for the sort-based kernel on config 3).
__device__ __attribute__((noinline)) void bloom_segment_item_direct_call(uint32_t* data, uint32_t seg, uint32_t item) {
  const uint64_t byte = ((uint64_t)seg << kBloomSegBits) | ((item >> 3) & 0xFFFFu);
  bloom_bump(data, byte, item & 7u);
}
struct BloomRingDirect {
  uint32_t* data;
  __device__ void operator()(uint32_t dest, uint64_t item, uint32_t cnt) const { for(uint32_t i = 0; i < cnt; ++i) bloom_segment_item_direct_call(data, dest, (uint32_t)item); }
};

// ---- P1b ----------------------------------------------------------------------------------------------------
// One block iteration = 16384 sequence positions.  The lanes roll their 16 windows together, one position per round:
// a round gives every lane at most kBloomPer items (nh > kBloomPer takes more rounds), 10240 per block, sorted and
// placed by granule_emit.  NB: key bytes fed to the two hash tables (compile-time for the common widths).
// PER: cell updates per lane and round (nh > PER takes more rounds).  NIB: the two hashes come from nibble tables -- one
// 16-byte entry (both hashes) per value of every 4 key bits, 512 bytes per key byte instead of 4 KiB, built here from
// the byte tables: sixteen ds_read_b128 per k-mer instead of sixteen ds_read_b64.  <NB, 5, true> needs 35 KB of dynamic
// LDS instead of 93: two workgroups per CU (the pass is latency-bound at 16 waves per CU, section 3.4 of DESIGN.md).
template <int NB, int PER, bool NIB>
__device__ __forceinline__ void p1_bloom_granule_body(const DevBloom& B, const BloomPart& BP, const TableGeom& g, const uint8_t* __restrict__ base,
                                                      int64_t lo, int64_t hi, uint32_t cap,
                                                      unsigned int* __restrict__ gcur, unsigned long long* __restrict__ tot,
                                                      uint32_t* __restrict__ out, unsigned long long* __restrict__ mers) {
  constexpr int kChunk = kPBlock * PER;
  JF_DYN_LDS(s_dyn);
  uint32_t* s_item = reinterpret_cast<uint32_t*>(s_dyn);                               // [kChunk]
  uint16_t* s_bkt = reinterpret_cast<uint16_t*>(s_dyn + (size_t)kChunk * 4);           // [kChunk]
  uint64_t* s_t1 = reinterpret_cast<uint64_t*>(s_dyn + (size_t)kChunk * 6);            // [nbytes * 256]   (NIB: uint4 [nbytes * 32])
  uint64_t* s_t2 = s_t1 + (size_t)B.nbytes * 256;
  uint4* s_tn = reinterpret_cast<uint4*>(s_dyn + (size_t)kChunk * 6);
  __shared__ uint32_t s_codes[kPBlock + 2];
  __shared__ uint32_t s_inv[kPBlock + 2];
  __shared__ GranuleLds G;
  const uint32_t nb = 1u << BP.b1;
  if constexpr(NIB) {
    for(uint32_t i = threadIdx.x; i < B.nbytes * 32; i += blockDim.x) {      // entry (nibble j, value v) = the byte tables' entry of that value in that nibble
      const uint32_t j = i >> 4, v = i & 15u;
      const uint32_t at = (j >> 1) * 256 + ((j & 1) ? (v << 4) : v);
      const uint64_t a = B.tbl1[at], c = B.tbl2[at];
      s_tn[i] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)c, (uint32_t)(c >> 32));
    }
  } else {
    load_tables_lds(s_t1, B.tbl1, B.nbytes);
    load_tables_lds(s_t2, B.tbl2, B.nbytes);
  }
  granule_init(G, nb);
  const uint32_t k = g.k;
  const uint64_t kwin = k >= 64 ? ~0ull : ((1ull << k) - 1);
  const uint32_t rc_shift = 2 * (k - 1);
  uint32_t my_mers = 0, my_direct = 0;
  const int64_t n_tiles = (hi + kPTilePos - 1) / kPTilePos;
  TileRaw R = tile_fetch(base, (int64_t)blockIdx.x * kPTilePos, lo, hi);
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    lds_barrier();
    const LaneWords L = tile_stage(R, tile * kPTilePos, lo, hi, s_codes, s_inv);      // barrier inside
    R = tile_fetch(base, (tile + gridDim.x) * kPTilePos, lo, hi);                      // next tile's bytes travel meanwhile
    uint64_t fw = (((uint64_t)L.p2 << 32) | L.p1) & g.key_mask;
    uint64_t rc = revcomp64(fw, k);
#pragma unroll 1
    for(int j = 0; j < kPerLane; ++j) {
      const uint64_t c = (L.cur >> (2 * (15 - j))) & 3u;
      fw = ((fw << 2) | c) & g.key_mask;
      rc = (rc >> 2) | ((3ull - c) << rc_shift);
      const bool valid = ((L.inv48 >> (15 - j)) & kwin) == 0;
      uint64_t cell = 0, inc = 0;
      if(valid) {
        ++my_mers;
        const uint64_t key = (g.canonical && rc < fw) ? rc : fw;
        if constexpr(NIB) {
          constexpr int kNibs = NB ? 2 * NB : 16;
          uint32_t a0 = 0, a1 = 0, c0 = 0, c1 = 0;
#pragma unroll
          for(int j = 0; j < kNibs; ++j)
            if(NB || (uint32_t)j < 2 * B.nbytes) { const uint4 e = s_tn[j * 16 + ((uint32_t)(key >> (4 * j)) & 15u)]; a0 ^= e.x; a1 ^= e.y; c0 ^= e.z; c1 ^= e.w; }
          cell = bloom_mod(((uint64_t)a1 << 32) | a0, B.m, B.recip);
          inc = bloom_mod(((uint64_t)c1 << 32) | c0, B.m, B.recip);
        } else {
          cell = bloom_mod(hash_tables_t<NB>(s_t1, key, B.nbytes), B.m, B.recip);
          inc = bloom_mod(hash_tables_t<NB>(s_t2, key, B.nbytes), B.m, B.recip);
        }
      }
      for(uint32_t h0 = 0; h0 < B.nh; h0 += PER) {                                      // block-uniform trip count
        lds_barrier();
        for(uint32_t q = threadIdx.x; q < nb; q += blockDim.x) G.hist[q] = 0;
        lds_barrier();
        uint32_t it[PER], dr[PER];
#pragma unroll
        for(int e = 0; e < PER; ++e) {
          dr[e] = 0xFFFFFFFFu; it[e] = 0;
          if(valid && h0 + e < B.nh) {
            uint64_t byte; uint32_t dig;
            divmod5(cell, byte, dig);
            const uint64_t seg = byte >> kBloomSegBits;
            const uint32_t b = (uint32_t)(seg >> BP.b2);
            it[e] = ((uint32_t)(seg & ((1u << BP.b2) - 1)) << kBloomItemLow) | ((uint32_t)(byte & 0xFFFFu) << 3) | dig;
            dr[e] = (b << 16) | atomicAdd(&G.hist[b], 1u);
            cell += inc; if(cell >= B.m) cell -= B.m;
          }
        }
        my_direct += granule_emit(G, nb, cap, gcur, out, s_item, s_bkt, it, dr,
                                  [&](uint32_t b, uint32_t v) { bloom_item_direct(B, BP, b, v); });
      }
    }
  }
  granule_finish(G, nb, cap, tot, out);
  uint64_t w = my_mers;
  for(int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
  if((threadIdx.x & 63) == 0 && w) atomicAdd(mers, (unsigned long long)w);
  (void)my_direct;
}

template <int NB>
__global__ __launch_bounds__(kPBlock) void p1_bloom_granule_kernel(DevBloom B, BloomPart BP, TableGeom g, const uint8_t* __restrict__ base,
                                                                   int64_t lo, int64_t hi, uint32_t cap,
                                                                   unsigned int* __restrict__ gcur, unsigned long long* __restrict__ tot,
                                                                   uint32_t* __restrict__ out, unsigned long long* __restrict__ mers) {
  p1_bloom_granule_body<NB, kBloomPer, false>(B, BP, g, base, lo, hi, cap, gcur, tot, out, mers);
}
// two workgroups per CU: eight waves per SIMD, so at most 64 vector registers
template <int NB>
__global__ __launch_bounds__(kPBlock) __attribute__((amdgpu_waves_per_eu(8, 8)))
void p1_bloom_granule2_kernel(DevBloom B, BloomPart BP, TableGeom g, const uint8_t* __restrict__ base,
                              int64_t lo, int64_t hi, uint32_t cap,
                              unsigned int* __restrict__ gcur, unsigned long long* __restrict__ tot,
                              uint32_t* __restrict__ out, unsigned long long* __restrict__ mers) {
  p1_bloom_granule_body<NB, 5, true>(B, BP, g, base, lo, hi, cap, gcur, tot, out, mers);
}

// (P1b through the count path's rings -- a ring of 32 cell updates per bucket, `cpr` cells of the current k-mer appended per
// lane and round, two barriers a round -- was built and measured in round 4, profiles/r04_c3_p1b.log: 450 ms for config 3's
// pass against 415 for the kernel above; a round cost 4.4 - 5.9 us whether it appended 2, 3 or 4 cells a lane.  Not the
// reservations (static slices of the regions, nothing reserved: 597 ms), not the division by five per cell (cells walked
// in (byte, digit) coordinates with additions only: 548 ms, and 431 in the kernel above).  SQ counters of the kernel
// above: VALU busy 50 % of the CU's cycles, LDS 47 %, waves waiting 64 % of theirs at four waves per SIMD -- the pass is
// bound by latency at this occupancy (93 KB of LDS per workgroup), which rings of 64 KB + 32 KB of tables do not change.)

// ---- Tb: one workgroup owns one 64 KiB segment of the byte array in LDS ---------------------------------------
__device__ inline void bloom_lds_bump(uint32_t* s_seg, uint32_t item) {
  const uint32_t off = (item >> 3) & 0xFFFFu, dig = item & 7u;
  uint32_t* w = s_seg + (off >> 2);
  const uint32_t sh = 8 * (off & 3);
  const uint32_t add = bloom_pow3(dig) << sh;
  uint32_t old = *w;
  while(true) {
    if(bloom_digit((old >> sh) & 0xFFu, dig) >= 2) return;        // saturated: bloom_counter2.hpp:78-104 stops at 2
    const uint32_t seen = atomicCAS(w, old, old + add);
    if(seen == old) return;
    old = seen;
  }
}

// Segment index = seg0 + t; its items are, for each pending array s, items[s][off(s, t) .. ) as in tile_insert_kernel.
__global__ __launch_bounds__(kPBlock) void bloom_segment_kernel(DevBloom B, SegList S, uint32_t n_seg, uint32_t seg0 = 0) {
  JF_DYN_LDS(s_raw);
  uint32_t* s_seg = reinterpret_cast<uint32_t*>(s_raw);
  constexpr uint32_t kWords = (1u << kBloomSegBits) / 4;
  for(uint32_t t = blockIdx.x; t < n_seg; t += gridDim.x) {
    uint64_t n_items = 0;
    for(uint32_t s = 0; s < S.n; ++s) n_items += seg_hi(S, s, t) - seg_lo(S, s, t);
    if(n_items == 0) continue;                                      // block-uniform
    uint32_t* gseg = B.data + ((size_t)seg0 + t) * kWords;             // (offsets are indexed from seg0: a group of buckets)
    for(uint32_t i = threadIdx.x * 4; i < kWords; i += blockDim.x * 4)
      *reinterpret_cast<uint4*>(s_seg + i) = *reinterpret_cast<const uint4*>(gseg + i);
    lds_barrier();
    for(uint32_t s = 0; s < S.n; ++s) {
      const uint64_t a = seg_lo(S, s, t), b = seg_hi(S, s, t);
      const uint32_t* src = reinterpret_cast<const uint32_t*>(S.items[s]);
      const bool holes = S.sh[s] != 0;
      constexpr int U = 8;                                          // loads in flight per lane
      for(uint64_t v0 = a + threadIdx.x; v0 < b; v0 += (uint64_t)U * blockDim.x) {
        uint32_t x[U];
#pragma unroll
        for(int u = 0; u < U; ++u) { const uint64_t v = v0 + (uint64_t)u * blockDim.x; x[u] = v < b ? src[v] : 0xFFFFFFFFu; }
#pragma unroll
        for(int u = 0; u < U; ++u) {
          const uint64_t v = v0 + (uint64_t)u * blockDim.x;
          if(v < b && !(holes && x[u] == 0xFFFFFFFFu)) bloom_lds_bump(s_seg, x[u]);
        }
      }
    }
    lds_barrier();
    for(uint32_t i = threadIdx.x * 4; i < kWords; i += blockDim.x * 4)
      *reinterpret_cast<uint4*>(gseg + i) = *reinterpret_cast<const uint4*>(s_seg + i);
    lds_barrier();
  }
}

// Too few updates to be worth streaming the filter: apply pending items with global CAS.  Same item layouts as
// items_direct_kernel (cap == 0: packed, bucket by binary search; cap > 0: granule regions with holes).
__global__ __launch_bounds__(kBlock) void bloom_items_direct_kernel(DevBloom B, BloomPart BP, const uint32_t* __restrict__ items,
                                                                    const uint64_t* __restrict__ off, uint64_t cap) {
  const uint32_t nb = 1u << BP.b1;
  const uint64_t n = (uint64_t)nb * cap;
  for(uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t b = (uint32_t)(v / cap);
    if(v >= off[2 * (size_t)b + 1]) continue;
    const uint32_t it = items[v];
    if(it == 0xFFFFFFFFu) continue;
    bloom_item_direct(B, BP, b, it);
  }
}

}  // namespace jfgpu
