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
#include "kernels_p1ring.hip.hpp"
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

// ---- P1b through rings of 256 bytes (round 6) ---------------------------------------------------------------------------
// The sort-based kernel above writes a chunk's run of a bucket wherever the region's cursor stands: 735 GB of HBM writes for
// 320 GB of cell updates on config 3 (profiles/r05_traffic_C3.json: 2.3 x), five to seven barriers per 5 Ki updates, 406 ms.
// Round 4's ring version (rings of 32 updates, two barriers a round, at most four cells a lane and round because a ring of 32
// takes no more) paid 4.4 - 5.9 us a round whatever it held.  Here: config 3's P1b has 512 buckets (417 in use: the array
// ends inside the last one), so a ring may have 256 bytes -- 64 updates, four 64-byte units -- and ONE round takes all ten
// cells of a k-mer: 10 Ki updates a round, 24.5 a ring in use (15 left over + 24.5 + 3 sigma = 55 < 64).  The rest is the
// count path's ring kernel (kernels_p1ring.hip.hpp): appends by one returning ds_add and one store per update in two
// sweeps, one barrier a round, owner lanes write the complete units out as aligned 64-byte runs (reservations asked a
// round ahead), what finds its ring full goes on the workgroup's list (p1_stragglers_kernel).  A cell update never equals
// the hole marker (its digit field is at most 4).  The two hashes come from nibble tables (4 KB for k = 31) like the
// two-workgroup kernel's.  PER: cells per lane and round (10: >= 400 buckets in use, 5: >= 200).
constexpr uint32_t kBloomRingBytes = 256;
__device__ __attribute__((noinline)) void bloom_p1_item_direct_call(uint32_t* data, uint32_t b2, uint32_t bucket, uint32_t item) {
  const uint64_t seg = ((uint64_t)bucket << b2) | (item >> kBloomItemLow);
  const uint64_t byte = (seg << kBloomSegBits) | ((item >> 3) & 0xFFFFu);
  bloom_bump(data, byte, item & 7u);
}
struct BloomP1RingDirect {
  static constexpr bool kCountsDirect = false;
  uint32_t* data; uint32_t b2;
  __device__ void operator()(uint32_t bucket, uint64_t item, uint32_t cnt) const { for(uint32_t i = 0; i < cnt; ++i) bloom_p1_item_direct_call(data, b2, bucket, (uint32_t)item); }
};

template <int NB, int PER>
__global__ __launch_bounds__(kPBlock) void p1_bloom_ring_kernel(DevBloom B, BloomPart BP, TableGeom g, BloomP1RingDirect D, const uint8_t* __restrict__ base,
                                                                int64_t lo, int64_t hi, uint32_t cap,
                                                                unsigned int* __restrict__ gcur, unsigned long long* __restrict__ tot,
                                                                uint32_t* __restrict__ out, unsigned long long* __restrict__ mers,
                                                                uint64_t* __restrict__ strag, uint32_t* __restrict__ strag_n) {
  using R = Ring<uint32_t, kBloomRingBytes>;
  JF_DYN_LDS(s_dyn);
  const uint32_t nb = 1u << BP.b1;
  uint32_t* s_ring = reinterpret_cast<uint32_t*>(s_dyn);                                // [nb][64], then 64 dump slots
  uint4* s_tn = reinterpret_cast<uint4*>(s_dyn + (size_t)nb * kBloomRingBytes + kBloomRingBytes);      // [nbytes * 32] nibble tables
  __shared__ uint32_t s_fill[kGranMaxB + 32];
  __shared__ uint32_t s_nstrag;
  __shared__ uint32_t s_codes[kPBlock + 2];
  __shared__ uint32_t s_inv[kPBlock + 2];
  const uint32_t t = threadIdx.x, lane = t & 63;
  const bool owner = t < nb;
  for(uint32_t i = t; i < B.nbytes * 32; i += blockDim.x) {        // entry (nibble j, value v) = the byte tables' entry of that value in that nibble
    const uint32_t j = i >> 4, v = i & 15u;
    const uint32_t at = (j >> 1) * 256 + ((j & 1) ? (v << 4) : v);
    const uint64_t a = B.tbl1[at], c = B.tbl2[at];
    s_tn[i] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)c, (uint32_t)(c >> 32));
  }
  ring_init<uint32_t, kBloomRingBytes>(s_ring, s_fill, nb, &s_nstrag);
  const uint32_t dump = nb * R::kSlots + (lane & (R::kSlots - 1));
  unsigned int* const gshort = gcur + nb;
  uint64_t* const my_strag = strag + (size_t)blockIdx.x * kStragPerBlock;
  const uint32_t k = g.k, rc_shift = 2 * (k - 1);
  const uint32_t sub_mask = (1u << BP.b2) - 1u;
  RingBooks Bk;
  if(owner) { Bk.nxt = atomicAdd(&gcur[t], kGran); Bk.nxt_asked = true; }
  uint32_t* const my_region = out + (uint64_t)t * cap;
  uint32_t my_mers = 0;
  auto straggler = [&](uint32_t b, uint32_t item, uint32_t cnt) {
    const uint32_t at = atomicAdd(&s_nstrag, 1u);
    if(at < kStragPerBlock) strag_store<uint32_t>(my_strag + at, b, item, cnt);
    else D(b, (uint64_t)item, cnt);
  };
  const int64_t n_tiles = (hi + kPTilePos - 1) / kPTilePos;
  TileRaw Rw = tile_fetch(base, (int64_t)blockIdx.x * kPTilePos, lo, hi);
  lds_barrier();
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const LaneWords L = tile_stage(Rw, tile * kPTilePos, lo, hi, s_codes, s_inv);      // barrier inside
    Rw = tile_fetch(base, (tile + gridDim.x) * kPTilePos, lo, hi);
    uint64_t fw = (((uint64_t)L.p2 << 32) | L.p1) & g.key_mask;
    uint64_t rc = revcomp64(fw, k);
    uint64_t smear = L.inv48;                                        // (kernels_p1ring.hip.hpp: which positions end a window of k valid bases)
    for(uint32_t s = 1; s < k; ) { const uint32_t step = s < k - s ? s : k - s; smear |= smear >> step; s += step; }
    const uint32_t vmask = ~(uint32_t)smear & 0xFFFFu;
    my_mers += (uint32_t)__popc(vmask);
#pragma unroll 1
    for(int j = 0; j < kPerLane; ++j) {
      const uint64_t c = (L.cur >> (2 * (15 - j))) & 3u;
      fw = ((fw << 2) | c) & g.key_mask;
      rc = (rc >> 2) | ((3ull - c) << rc_shift);
      const bool valid = (vmask >> (15 - j)) & 1u;
      const uint64_t key = (g.canonical && rc < fw) ? rc : fw;
      // (a position without a k-mer hashes whatever its registers hold and appends to the spare fill words: no branch)
      uint32_t a0 = 0, a1 = 0, c0 = 0, c1 = 0;
      constexpr int kNibs = NB ? 2 * NB : 16;
#pragma unroll
      for(int q = 0; q < kNibs; ++q)
        if(NB || (uint32_t)q < 2 * B.nbytes) { const uint4 e = s_tn[q * 16 + ((uint32_t)(key >> (4 * q)) & 15u)]; a0 ^= e.x; a1 ^= e.y; c0 ^= e.z; c1 ^= e.w; }
      uint64_t cell = bloom_mod(((uint64_t)a1 << 32) | a0, B.m, B.recip);
      const uint64_t inc = bloom_mod(((uint64_t)c1 << 32) | c0, B.m, B.recip);
      // (Round 6 also walked the cells in (byte, digit) coordinates -- one division by five per k-mer and step instead of one
      // per cell: the carry and wrap tests came out as branches, 655 vector instructions a round instead of 526.  The round
      // is not bound by them anyway: 526 x 16 waves = 3.5 us of issue against the 9.1 us a round takes.)
      for(uint32_t h0 = 0; h0 < B.nh; h0 += PER) {                    // block-uniform trip count (one round for nh = PER)
        uint32_t ea[PER], eo[PER], ei[PER];
#pragma unroll
        for(int e = 0; e < PER; ++e) {
          const bool on = valid && h0 + e < B.nh;
          uint64_t byte; uint32_t dig;
          divmod5(cell, byte, dig);
          const uint32_t seg = (uint32_t)(byte >> kBloomSegBits);
          const uint32_t b = seg >> BP.b2;
          ei[e] = ((seg & sub_mask) << kBloomItemLow) | ((uint32_t)(byte & 0xFFFFu) << 3) | dig;
          const uint32_t o = atomicAdd(&s_fill[on ? b : nb + (lane & 31u)], 1u);
          ea[e] = on ? b * R::kSlots : dump; eo[e] = on ? o : 0u;
          cell += inc; if(cell >= B.m) cell -= B.m;
        }
        uint32_t ghosts = 0;
#pragma unroll
        for(int e = 0; e < PER; ++e) {
          const uint32_t full = eo[e] & R::kFull;
          ghosts |= full;
          const uint32_t at = ea[e] + ((eo[e] + (eo[e] >> 16)) & (R::kSlots - 1));
          s_ring[full ? dump : at] = ei[e];
        }
        if(ghosts) {
#pragma unroll 1
          for(int e = 0; e < PER; ++e) if(eo[e] & R::kFull) straggler(ea[e] / R::kSlots, ei[e], 1u);
        }
        lds_barrier();                                                // the round's updates have all landed
        wave_prio<3>();
        if(owner) ring_flush<uint32_t, false, kBloomRingBytes>(s_ring, s_fill, t, false, Bk, my_region, cap, gcur, gshort, straggler);
        wave_prio<0>();
#ifdef JFGPU_BLOOM_P1_BAR2
        lds_barrier();
#endif
      }
    }
  }
  lds_barrier();
  if(owner) { ring_flush<uint32_t, false, kBloomRingBytes>(s_ring, s_fill, t, true, Bk, my_region, cap, gcur, gshort, straggler); ring_finish<uint32_t>(Bk, t, my_region, cap, gshort, tot); }
  lds_barrier();
  if(t == 0) strag_n[blockIdx.x] = s_nstrag < kStragPerBlock ? s_nstrag : kStragPerBlock;
  uint64_t w = my_mers;
  for(int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
  if((threadIdx.x & 63) == 0 && w) atomicAdd(mers, (unsigned long long)w);
}

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
