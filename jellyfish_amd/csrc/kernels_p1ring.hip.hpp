// jellyfish_amd/csrc/kernels_p1ring.hip.hpp -- stage P1 of the partitioned insert path for 32-bit items (gfx950).
//
// Round 2's single-pass P1 counting-sorted every chunk of 16 Ki positions by bucket in LDS (histogram, scan, ranked
// scatter, read-back by bucket, placement table) and wrote runs of ~16 items wherever the chunk's run of a bucket
// happened to start: seven barriers per chunk on one workgroup per CU, and 64-byte runs straddling 128-byte lines
// (1.6 x write amplification, profiles/r02_traffic_C2.json).  Here the sort is gone: every bucket owns a RING of 32 items
// in LDS (1024 x 128 B), a k-mer's item is appended with one returning ds_add on the bucket's fill word and one store,
// and after a round of 8 Ki positions every bucket that has 16 items together emits them as one aligned 64-byte unit
// (tools/probes/scatter_write_probe.hip: aligned 64-byte runs travel at twice the rate of straddling ones).  Lane t of the
// workgroup keeps bucket t's bookkeeping (read cursor, place in the region, reservations) in registers; the copy LDS ->
// region is done by four lanes per unit.  Region format, reservations of kGran items and holes are exactly the granule
// kernels' (kernels_part.hip.hpp), so P2 and the tile kernel read the output as before.
#pragma once
#include "kernels_part.hip.hpp"

namespace jfgpu {

// admission mask of count --bc (bit j <-> position j) in the bit order of the validity masks (bit 15 - j <-> position j)
__device__ __forceinline__ uint32_t adm_to_vmask(uint32_t adm) {
  uint32_t r = 0;
#pragma unroll
  for(int j = 0; j < 16; ++j) r |= ((adm >> j) & 1u) << (15 - j);
  return r;
}

constexpr uint32_t kRingSlots = 32, kRingUnit = 16;      // items per ring and per emitted unit
constexpr uint32_t kRingDirect = 0xFFFFFFFFu;            // a unit with nowhere to go in its region: inserted directly

// fill word of a bucket: (ring position of its oldest item) << 16 | items in the ring.  The count may run past the ring's
// size inside a round (what came too late was inserted directly by its own lane); it is cut back at the round's end.
template <bool RETURNING, bool BLOOM, int NB>
__global__ __launch_bounds__(kPBlock) void p1_ring_kernel(DevTable T, const DevTable* __restrict__ Tmem, PartGeom P, const uint8_t* __restrict__ base,
                                                          int64_t lo, int64_t hi, uint32_t cap,
                                                          unsigned int* __restrict__ gcur,
                                                          unsigned long long* __restrict__ tot,
                                                          uint32_t* __restrict__ out) {
  JF_DYN_LDS(s_dyn);
  uint32_t* s_ring = reinterpret_cast<uint32_t*>(s_dyn);          // [nb][kRingSlots]
  __shared__ uint64_t s_fwd[8 * 256];
  __shared__ uint32_t s_codes[kPBlock + 2];
  __shared__ uint32_t s_inv[kPBlock + 2];
  __shared__ uint32_t s_fill[kGranMaxB];
  const TableGeom& g = T.g;
  const uint32_t nb = 1u << P.b1;
  const uint32_t t = threadIdx.x, lane = t & 63;
  const bool owner = t < nb;                                       // this lane keeps bucket t's books
  load_tables_lds(s_fwd, T.fwd_tbl, g.nbytes);
  for(uint32_t j = t; j < nb; j += blockDim.x) s_fill[j] = 0;
  unsigned int* const gshort = gcur + nb;
  const uint32_t k = g.k, bshift = g.lsize_l - P.b1;
  const uint32_t rc_shift = 2 * (k - 1);
  const uint32_t hole = 0xFFFFFFFFu;
  // bucket t's place in its region: gpos .. gpos + room of the current reservation, `nxt` the reservation asked for in
  // advance (its answer is first looked at a round later), kNoRoom when the region has none left
  uint32_t gpos = 0, room = 0, nxt = 0, stored = 0;
  bool nxt_asked = false, exhausted = false;
  if(owner) { nxt = atomicAdd(&gcur[t], kGran); nxt_asked = true; }
  uint32_t my_direct = 0, my_mers = 0;

  auto direct = [&](uint32_t b, uint32_t item, uint32_t cnt = 1) { item_direct_call(Tmem, P.b2, b, item, cnt, RETURNING ? 1 : 0); ++my_direct; };

  // After a round: every bucket with 16 items or more emits whole units (at most two: the ring holds 32); `all`: the
  // kernel's last call also emits what is left, padded with holes.
  auto flush = [&](bool all) {
    uint32_t has[2] = {0, 0}, roff[2] = {0, 0}, dest[2] = {0, 0};
    if(owner) {
      const uint32_t w = s_fill[t];
      uint32_t cnt = w & 0xFFFFu, rb = (w >> 16) & (kRingSlots - 1);
      if(cnt > kRingSlots) cnt = kRingSlots;
      uint32_t units = cnt / kRingUnit;
      if(all && (cnt % kRingUnit)) {                               // the last, partial unit: holes behind its items
        for(uint32_t i = cnt; i < (units + 1) * kRingUnit; ++i) s_ring[t * kRingSlots + ((rb + i) & (kRingSlots - 1))] = hole;
        ++units;
      }
      for(uint32_t s = 0; s < units; ++s) {
        if(room == 0 && nxt_asked) {                               // take the reservation asked for earlier
          if((uint64_t)nxt + kGran <= cap) { gpos = nxt; room = kGran; }
          else { exhausted = true; if(nxt < cap) atomicMax(&gshort[t], cap - nxt); }     // (everything below nxt was handed out)
          nxt_asked = false;
        }
        // no reservation in hand and none asked for (the one taken above was used up inside this flush): ask now and
        // wait -- rare, and cheaper than sixteen global-atomic inserts for a region that still has room
        if(room == 0 && !exhausted) {
          const uint32_t r0 = atomicAdd(&gcur[t], kGran);
          if((uint64_t)r0 + kGran <= cap) { gpos = r0; room = kGran; }
          else { exhausted = true; if(r0 < cap) atomicMax(&gshort[t], cap - r0); }
        }
        has[s] = 1; roff[s] = (rb + s * kRingUnit) & (kRingSlots - 1);
        const uint32_t real = cnt - s * kRingUnit < kRingUnit ? cnt - s * kRingUnit : kRingUnit;
        if(room) { dest[s] = gpos; gpos += kRingUnit; room -= kRingUnit; stored += real; }
        else dest[s] = kRingDirect;
      }
      const uint32_t taken = units * kRingUnit < cnt ? units * kRingUnit : cnt;
      s_fill[t] = (((rb + units * kRingUnit) & (kRingSlots - 1)) << 16) | (cnt - taken);
      // keep one reservation in hand whenever the current one cannot take a ring's worth: its round trip to L2 hides
      // behind the next round
      if(!all && !nxt_asked && !exhausted && room < kRingSlots) { nxt = atomicAdd(&gcur[t], kGran); nxt_asked = true; }
    }
    // the copies: four lanes per unit (16 bytes each), sixteen owners per wave instruction
#pragma unroll
    for(int s = 0; s < 2; ++s) {
      const unsigned long long m = __ballot(has[s] != 0);
      if(!m) continue;                                             // wave-uniform
#pragma unroll 1
      for(int j = 0; j < 4; ++j) {
        if(!((m >> (16 * j)) & 0xFFFFull)) continue;               // wave-uniform
        const int own = 16 * j + (int)(lane >> 2);
        const uint32_t f = __shfl(has[s], own, 64), ro = __shfl(roff[s], own, 64), d = __shfl(dest[s], own, 64);
        if(!f) continue;
        const uint32_t b = (t & ~63u) + (uint32_t)own, part = lane & 3;
        const uint4 v = *reinterpret_cast<const uint4*>(s_ring + b * kRingSlots + ro + 4 * part);
        if(d != kRingDirect) *reinterpret_cast<uint4*>(out + (uint64_t)b * cap + d + 4 * part) = v;
        else {
          if(v.x != hole) direct(b, v.x);
          if(v.y != hole) direct(b, v.y);
          if(v.z != hole) direct(b, v.z);
          if(v.w != hole) direct(b, v.w);
        }
      }
    }
  };

  const int64_t n_tiles = (hi + kPTilePos - 1) / kPTilePos;
  TileRaw R = tile_fetch(base, (int64_t)blockIdx.x * kPTilePos, lo, hi);
  [[maybe_unused]] PhaseClk pc;
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    lds_barrier();
    const LaneWords L = tile_stage(R, tile * kPTilePos, lo, hi, s_codes, s_inv);      // barrier inside
    JF_PHASE(pc, 0);
    R = tile_fetch(base, (tile + gridDim.x) * kPTilePos, lo, hi);                      // next tile's bytes travel while this one is worked on
    const uint32_t adm = BLOOM ? bloom_admit_mask(T.bloom, g, L) : 0xFFFFu;
    uint64_t fw = (((uint64_t)L.p2 << 32) | L.p1) & g.key_mask;
    uint64_t rc = revcomp64(fw, k);
    // which of the lane's 16 positions end a window of k valid bases: the invalid-base bits smeared over the k - 1
    // positions after them, once per tile (bit 15 - j <-> position j, like inv48)
    uint64_t smear = L.inv48;
    for(uint32_t s = 1; s < k; ) { const uint32_t step = s < k - s ? s : k - s; smear |= smear >> step; s += step; }
    const uint32_t rawmask = ~(uint32_t)smear & 0xFFFFu;
    const uint32_t vmask = BLOOM ? (rawmask & adm_to_vmask(adm)) : rawmask;
    my_mers += (uint32_t)__popc(rawmask);
    // a k-mer is emitted one position late, when it is known whether the next one repeats it (homopolymers, tandem
    // repeats: one insert for the run): pk / pv / run describe the position before
    uint64_t pk = 0; uint32_t pv = 0, run = 0;
#pragma unroll 1
    for(int j0 = 0; j0 < kPerLane; j0 += kPerLane / 2) {            // two rounds of eight positions per lane
      constexpr int NE = kPerLane / 2 + 1;                          // (the +1: the tile's last k-mer, emitted after the loop)
      uint32_t eb[NE], ei[NE], eo[NE]; uint32_t em = 0;
      auto emit = [&](int e, uint64_t key, uint32_t cnt) {
        const uint64_t pos = hash_tables_t<NB>(s_fwd, key, g.nbytes);
        const uint64_t local = pos & g.local_mask;
        eb[e] = (uint32_t)(local >> bshift);
        ei[e] = make_item<uint32_t>(g, P, key, local);
        if(ei[e] == hole || cnt > 1) direct(eb[e], ei[e], cnt);      // (it would read as a hole; a run goes in at once)
        else { eo[e] = atomicAdd(&s_fill[eb[e]], 1u); em |= 1u << e; }
      };
#pragma unroll
      for(int e = 0; e < kPerLane / 2; ++e) {
        const int j = j0 + e;
        const uint64_t c = (L.cur >> (2 * (15 - j))) & 3u;
        fw = ((fw << 2) | c) & g.key_mask;
        rc = (rc >> 2) | ((3ull - c) << rc_shift);
        const uint32_t v = (vmask >> (15 - j)) & 1u;
        const uint64_t key = (g.canonical && rc < fw) ? rc : fw;
        const bool same = v && pv && key == pk;
        if(pv && !same) emit(e, pk, run);
        run = same ? run + 1 : 1;
        pk = key; pv = v;
      }
      if(j0) { if(pv) emit(NE - 1, pk, run); pv = 0; }
      // (second sweep: the ring stores, once the fill adds are back -- not one wait per item)
#pragma unroll
      for(int e = 0; e < NE; ++e)
        if((em >> e) & 1) {
          const uint32_t r = eo[e] & 0xFFFFu;
          if(r < kRingSlots) s_ring[eb[e] * kRingSlots + (((eo[e] >> 16) + r) & (kRingSlots - 1))] = ei[e];
          else direct(eb[e], ei[e]);                                 // more than a ring's worth for one bucket in one round: skewed input
        }
      JF_PHASE(pc, 1);
      lds_barrier();
      JF_PHASE(pc, 2);
      flush(false);
      JF_PHASE(pc, 3);
      lds_barrier();
    }
  }
  flush(true);
  if(owner) {
    // what is left of the reservations becomes holes; the exact count of the bucket goes to tot
    for(uint32_t r = 0; r < room; ++r) out[(uint64_t)t * cap + gpos + r] = hole;
    if(nxt_asked) {
      if((uint64_t)nxt + kGran <= cap) { for(uint32_t r = 0; r < kGran; ++r) out[(uint64_t)t * cap + nxt + r] = hole; }
      else if(nxt < cap) atomicMax(&gshort[t], cap - nxt);
    }
    if(tot && stored) atomicAdd(&tot[t], (unsigned long long)stored);
  }
  JF_PHASE(pc, 4);
  JF_PHASE_FLUSH(pc, 0);
  if(my_direct) atomicAdd((unsigned long long*)&T.counters[CTR_DIRECT], (unsigned long long)my_direct);
  uint64_t w = my_mers;
  for(int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
  if((threadIdx.x & 63) == 0 && w) atomicAdd((unsigned long long*)&T.counters[CTR_MERS], (unsigned long long)w);
}

}  // namespace jfgpu
