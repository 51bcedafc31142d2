// jellyfish_amd/csrc/kernels_tile.hip.hpp -- stage T of the partitioned insert path, one-word keys (gfx950).
//
// One workgroup owns one tile (TPB = 2: two adjacent tiles, one contiguous range of the table) in LDS and places the
// items P2 routed to it.  Round 2 claimed slots with an LDS compare-and-swap per probe; what that costs is the number of
// LDS atomic *instructions* a wave issues, not the number of lanes in them, and a chain of dependent probes runs with a
// handful of live lanes per wave (tools/probes/r03_lds_probe.hip: 14.5 us per pair of tiles against 7.5 for the scheme
// below, at the metric's load).  Every item of a tile is known before the first one is placed, so placement needs no
// claim at all:
//
//   slots are grouped in BUCKETS of four (16 bytes of 32-bit slots); a key's home bucket is the one its hash position
//   falls in, and the slots of a bucket fill front to back (kmer_core.hpp::probe_lin -- linear probing from the start of
//   the home bucket, still wrapping inside the tile, so look-ups, dumps, growth and the global-atomic path see one rule).
//
//   A  every item: one returning ds_add on its home bucket's 16-bit counter -> rank; rank < 4: plain store of the slot
//      word at bucket * 4 + rank; otherwise the item goes on a small LDS queue.  All of a lane's items are in flight
//      together (no dependent chain, full waves); for a tile nothing was ever inserted into, the adds are issued before
//      the tile is zeroed and their round trip is hidden behind that.
//   M  one lane per bucket, all buckets at once: equal tags among the bucket's (at most four) entries are merged into the
//      first of them and the bucket is compacted -- the k-mers that occur several times in this flush or were already in
//      the table.  This comes BEFORE C: a bucket that looks full but holds equal tags has room, and an item walking
//      past it would strand its key behind a free slot.  (Folding such buckets from inside C, next to lanes that add,
//      cannot be made safe with plain adds: an add issued on an old look can land in a slot whose entry was folded
//      away meanwhile.  With M first, nothing is ever taken out of a slot while C runs.)
//   C  the queued items (a few per cent at load 0.5; most of the input on high-coverage data, where they find their tag
//      in the home bucket and end as one add): vector look at a bucket, add on a match, compare-and-swap on the first
//      empty slot, next bucket otherwise.  C creates no equal tags, so the tile then goes out as it is.
//
// Same table format and count semantics as the global-atomic path (kernels.hip.hpp::table_add): (tile, tag) identifies
// the key wherever in the tile it lands; large_hash_array.hpp:509-597,741-752 (claim_key / add_val) restated.
#pragma once
#include "kernels_part.hip.hpp"

namespace jfgpu {

// ovf_add as a real call: the rare count-field wrap must not cost the hot path registers or code size.
__device__ __attribute__((noinline)) void ovf_add_call(uint64_t* ovf_key, uint64_t* ovf_cnt, uint64_t ovf_mask, uint64_t* counters,
                                                       uint64_t slot, uint64_t units) {
  DevTable V; V.ovf_key = ovf_key; V.ovf_cnt = ovf_cnt; V.ovf_mask = ovf_mask; V.counters = counters;
  ovf_add(V, slot, units);
}

#ifndef JFGPU_T_BLOCK
#define JFGPU_T_BLOCK 512
#endif
constexpr int kTileBlock = JFGPU_T_BLOCK;               // threads per workgroup: two workgroups per CU (LDS), 128 registers per lane
constexpr uint32_t kQueueLook = 2;             // buckets one look of phase C fetches (1: 24.9-25.1 ms, 2: 24.8, 4: 27.3 on the metric's job)
#ifndef JFGPU_T_QUEUE
#define JFGPU_T_QUEUE 6144
#endif
constexpr uint32_t kTileQueueBytes = JFGPU_T_QUEUE;    // LDS queue of phase C (what does not fit stays with the lane)

// dynamic LDS of one workgroup: slots | bucket counters (16 bit each) | queue header | queue
inline size_t tile_rank_lds(size_t slot_bytes, uint32_t tile_bits, int tpb) {
  const size_t nslots = (size_t)tpb << tile_bits;
  return nslots * slot_bytes + (nslots >> kBucketBits) * 2 + 16 + kTileQueueBytes;
}

// HEAVY: the instantiation for high-coverage input, where most of a round's items find their bucket full of -- after M --
// their own key.  One at a time through the queue that is a chain of dependent LDS round trips per item and lane
// (bench.py --dist G: T 53 ms against 28 on uniform reads).  Here a round's items stay in their registers past M and are
// resolved in bulk first (C0): home buckets of six items fetched together, tags compared, adds issued together; only
// what found no match goes on to the queue, and the prefetch of the next round's items starts after C0 instead of before
// M.  Which instantiation a flush runs is decided by the host from a sample of the flush itself (the first units go
// through the plain kernel, which counts how many items went past rank 3: SAMPLE) -- both paths in one kernel cost the
// common case 3 % (instruction cache).
template <typename ITEM, bool RETURNING, typename SLOT, int TPB, int BLOCK = kTileBlock, bool HEAVY = false, bool SAMPLE = false>
__global__ __launch_bounds__(BLOCK, (BLOCK >= 512 ? 2 * BLOCK / 256 : 4)) void tile_rank_insert_kernel(DevTable T, SegList S, uint64_t tile0, uint32_t n_tiles) {
  // S holds ONE item array (the P2 output, or one pending batch of a single-level table: the host launches per batch)
  // register-held items per lane and round: 9216 items per round; 4608 for 8-byte items (k = 31 into a single tile of
  // 8-byte slots: as many items as such a tile takes in one flush at load 0.5, and half the registers -- with 18
  // two-register items those instantiations spilled 100 registers and ran 4 x slower per item)
  // (-DJFGPU_T_BLOCK=256, an experiment of round 6: four workgroups of four waves per CU on single tiles, rounds of 4608 items)
  constexpr int NP = ((sizeof(ITEM) == 8 || (BLOCK < 512 && TPB == 1)) ? 4608 : 9216) / BLOCK;
  constexpr uint32_t kVec = 16 / sizeof(SLOT);              // slots per 16-byte vector
  constexpr uint32_t kB = 1u << kBucketBits;                // slots per bucket (kmer_core.hpp: the probe rule every path follows)
  constexpr uint32_t kBV = kB / kVec;                       // vectors per bucket
  constexpr uint32_t kSlotBits = 8 * sizeof(SLOT);
  JF_DYN_LDS(s_raw);
  const TableGeom& g = T.g;
  // (the partitioned path only exists for full-size tiles: part_geom_init)
  constexpr uint32_t tsz = 1u << kMaxTileBits, nslots = TPB * tsz, nbkt = nslots >> kBucketBits, tmask = tsz - 1;
  constexpr uint32_t NBK = nbkt / BLOCK;                    // buckets per lane
  SLOT* const s_tile = reinterpret_cast<SLOT*>(s_raw);
  uint32_t* const s_cnt = reinterpret_cast<uint32_t*>(s_raw + (size_t)nslots * sizeof(SLOT));      // two 16-bit counters per word
  uint32_t* const s_qn = s_cnt + (nbkt >> 1);                 // (16 bytes of padding)
  // phase C's queue: one segment per wave (positions come from a ballot: no atomic, no count to read back)
  constexpr uint32_t qcap = kTileQueueBytes / sizeof(ITEM) / (BLOCK / 64);
  static_assert(qcap >= 64, "phase C stages one held-back item per lane through the queue");
  ITEM* const s_q = reinterpret_cast<ITEM*>(s_qn + 4) + (threadIdx.x >> 6) * qcap;
  SLOT* const gslots = reinterpret_cast<SLOT*>(T.slots);
  const SLOT lmask = (SLOT)g.low_mask, inc = (SLOT)g.inc, occ = (SLOT)g.occ_bit;
  const uint32_t cshift = g.tag_bits + 1, idshift = kSlotBits - cshift;
  const uint32_t lane = threadIdx.x & 63;
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
  const ITEM hole = (ITEM)~(ITEM)0;

  auto unit_dirty = [&](uint32_t t) -> uint32_t {
    if(TPB == 2) return *reinterpret_cast<const uint16_t*>(T.dirty + tile0 + 2 * (uint64_t)t);
    return T.dirty[tile0 + t];
  };
  auto zero_counters = [&]() {
    for(uint32_t i = threadIdx.x; i < (nbkt >> 1); i += BLOCK) s_cnt[i] = 0;
  };
  auto zero_tile = [&]() {
    for(uint32_t i = threadIdx.x * kVec; i < nslots; i += BLOCK * kVec) *reinterpret_cast<uint4*>(s_tile + i) = make_uint4(0, 0, 0, 0);
  };
  // tile(s) -> LDS, one bucket per lane and step (halves nothing was ever inserted into are not read); the bucket
  // counters start at the number of entries already there (front to back, so that is also the first free slot)
  auto load_tile = [&](const SLOT* gt, uint32_t d) {
    for(uint32_t b = threadIdx.x; b < nbkt; b += BLOCK) {
      const bool ld = (d >> (8 * ((b << kBucketBits) >> kMaxTileBits))) & 0xFFu;
      uint32_t c = 0;
#pragma unroll
      for(uint32_t q = 0; q < kBV; ++q) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if(ld) v = *reinterpret_cast<const uint4*>(gt + ((size_t)b << kBucketBits) + q * kVec);
        *reinterpret_cast<uint4*>(s_tile + ((size_t)b << kBucketBits) + q * kVec) = v;
        if(sizeof(SLOT) == 4) c += (v.x != 0) + (v.y != 0) + (v.z != 0) + (v.w != 0);
        else c += ((v.x | v.y) != 0) + ((v.z | v.w) != 0);
      }
      reinterpret_cast<uint16_t*>(s_cnt)[b] = (uint16_t)c;
    }
  };
  // an item's home slot inside the unit: idx0, plus the tile-select bit of a pair (it sits right above idx0 in the item)
  auto home_of = [&](ITEM x) -> uint32_t {
    if constexpr(sizeof(ITEM) == 4) return ((uint32_t)x >> g.rem_bits) & (nslots - 1);
    else return (uint32_t)((uint64_t)x >> g.rem_bits) & (nslots - 1);
  };
  auto load_bucket = [&](uint32_t bs, SLOT (&w)[kB]) {
#pragma unroll
    for(uint32_t q = 0; q < kBV; ++q) {
      const uint4 v = *reinterpret_cast<const uint4*>(s_tile + bs + q * kVec);
      if constexpr(sizeof(SLOT) == 4) { w[4 * q] = (SLOT)v.x; w[4 * q + 1] = (SLOT)v.y; w[4 * q + 2] = (SLOT)v.z; w[4 * q + 3] = (SLOT)v.w; }
      else { w[2 * q] = (SLOT)(((uint64_t)v.y << 32) | v.x); w[2 * q + 1] = (SLOT)(((uint64_t)v.w << 32) | v.z); }
    }
  };
  auto store_bucket = [&](uint32_t bs, const SLOT (&w)[kB]) {
#pragma unroll
    for(uint32_t q = 0; q < kBV; ++q) {
      uint4 v;
      if constexpr(sizeof(SLOT) == 4) v = make_uint4((uint32_t)w[4 * q], (uint32_t)w[4 * q + 1], (uint32_t)w[4 * q + 2], (uint32_t)w[4 * q + 3]);
      else v = make_uint4((uint32_t)w[2 * q], (uint32_t)((uint64_t)w[2 * q] >> 32), (uint32_t)w[2 * q + 1], (uint32_t)((uint64_t)w[2 * q + 1] >> 32));
      *reinterpret_cast<uint4*>(s_tile + bs + q * kVec) = v;
    }
  };
  // do two of a bucket's entries carry the same tag?  identity = occupied bit + tag (count shifted out).  Six compares
  // whose results meet in scalar registers; an empty slot equals nothing.
  auto has_dups = [&](const SLOT (&w)[kB]) -> bool {
    SLOT sh[kB];
#pragma unroll
    for(uint32_t i = 0; i < kB; ++i) sh[i] = w[i] << idshift;
    bool d = false;
#pragma unroll
    for(uint32_t j = 1; j < kB; ++j) {
      bool e = false;
#pragma unroll
      for(uint32_t i = 0; i < j; ++i) e = e | (sh[i] == sh[j]);
      d = d | (e & (w[j] != 0));
    }
    return d;
  };
  // M: equal tags of one bucket into the first of them, compacted to the front.  Straight-line code (selects, no
  // branches: a wave runs this whenever one of its 64 buckets needs it) on static indices (registers).
  auto merge_bucket = [&](SLOT (&w)[kB], uint64_t bucket_slot0) {
    const SLOT cmask = (SLOT)g.cnt_max;
    SLOT carry[kB];                                          // count-field wrap-arounds, in units of 2^cnt_bits
#pragma unroll
    for(uint32_t i = 0; i < kB; ++i) carry[i] = 0;
#pragma unroll
    for(uint32_t j = 1; j < kB; ++j)
#pragma unroll
      for(uint32_t i = 0; i < j; ++i) {
        const bool same = (w[j] != 0) & (w[i] != 0) & (((w[i] ^ w[j]) & lmask) == 0);
        const SLOT sum = (SLOT)(w[i] >> cshift) + (same ? (SLOT)(w[j] >> cshift) : (SLOT)0);      // (no carry out of the word: both counts < 2^cnt_bits <= 2^(bits - 2))
        w[i] = (w[i] & lmask) | (SLOT)((sum & cmask) << cshift);
        carry[i] += (SLOT)(sum >> g.cnt_bits);
        w[j] = same ? (SLOT)0 : w[j];
      }
    // stable compaction: entry j moves to the number of non-empty entries before it (selects on static indices)
    SLOT o[kB], oc[kB];
#pragma unroll
    for(uint32_t i = 0; i < kB; ++i) { o[i] = 0; oc[i] = 0; }
    uint32_t n = 0;
#pragma unroll
    for(uint32_t j = 0; j < kB; ++j) {
      const bool ne = w[j] != 0;
#pragma unroll
      for(uint32_t d = 0; d <= j; ++d) { const bool here = ne & (n == d); o[d] = here ? w[j] : o[d]; oc[d] = here ? carry[j] : oc[d]; }
      n += ne ? 1u : 0u;
    }
    SLOT any = 0;
#pragma unroll
    for(uint32_t q = 0; q < kB; ++q) { w[q] = o[q]; any |= oc[q]; }
    if(RETURNING && any != 0) {                                                // rare: some count left its field
#pragma unroll
      for(uint32_t q = 0; q < kB; ++q) if(oc[q]) ovf_add_call(T.ovf_key, T.ovf_cnt, T.ovf_mask, T.counters, bucket_slot0 + q, (uint64_t)oc[q]);
    }
  };
  // M for the whole unit, between A and C.  Every lane looks at its NBK buckets (fetched MB at a time in one LDS round
  // trip); the buckets that hold equal tags -- a few per wave on uniform reads, a hundred per unit on high-coverage
  // input -- go on a per-wave list and are folded from there one per lane: a wave runs the (long, branch-free) merge once
  // per 64 such buckets instead of once per bucket position in which any of its lanes has one.  The list lives in the
  // bucket counters' LDS, which nothing reads between the ranks of A and the store that zeroes them.
  auto merge_tile = [&](uint64_t unit_slot0) {
    constexpr uint32_t MB0 = (TPB == 2 && sizeof(ITEM) + sizeof(SLOT) == 8 && NBK >= 4) ? 4 : (NBK >= 2 ? 2 : 1);      // buckets in flight per lane (registers: the next round's items are too)
    // ring of bucket numbers per wave, in the bucket counters' LDS (2 bytes a bucket): 63 waiting + MB x 64 new at most
    constexpr uint32_t Rmax = nbkt / (BLOCK / 64);
    static_assert(Rmax >= 128, "the merge list must fit the bucket counters");
    constexpr uint32_t R = Rmax >= 512 ? 512 : Rmax >= 256 ? 256 : 128;
    constexpr uint32_t MBr = R >= 512 ? 4 : R >= 256 ? 2 : 1;                                    // what the list takes
    constexpr uint32_t MBw = 16 / kB >= 1 ? 16 / kB : 1;                                         // at most sixteen slot words in flight
    constexpr uint32_t MB = MB0 < MBr ? (MB0 < MBw ? MB0 : MBw) : (MBr < MBw ? MBr : MBw);
    static_assert((BLOCK / 64) * R * 2 <= nbkt * 2, "the merge list must fit the bucket counters");
    uint16_t* const lst = reinterpret_cast<uint16_t*>(s_cnt) + (threadIdx.x >> 6) * R;
    uint32_t head = 0, cnt = 0;                               // wave-uniform
#pragma unroll 1
    for(uint32_t k0 = 0; k0 < NBK; k0 += MB) {
      SLOT wb[MB][kB];
#pragma unroll
      for(uint32_t k = 0; k < MB; ++k) load_bucket((threadIdx.x + (k0 + k) * BLOCK) << kBucketBits, wb[k]);
#pragma unroll
      for(uint32_t k = 0; k < MB; ++k) {
        const bool dup = has_dups(wb[k]);
        const unsigned long long m = __ballot(dup);
        if(dup) lst[(head + cnt + (uint32_t)__popcll(m & below)) & (R - 1)] = (uint16_t)(threadIdx.x + (k0 + k) * BLOCK);
        cnt += (uint32_t)__popcll(m);
      }
      const bool last = k0 + MB >= NBK;
      while(cnt >= 64 || (last && cnt)) {                     // (wave-uniform)
        const uint32_t n = cnt < 64 ? cnt : 64;
        (void)__ballot(true);                                 // (the list entries are written before they are read: lockstep on the device, a rendezvous in the host emulation)
        if(lane < n) {
          const uint32_t b = lst[(head + lane) & (R - 1)];
          SLOT w[kB];
          load_bucket(b << kBucketBits, w);
          merge_bucket(w, unit_slot0 + ((uint64_t)b << kBucketBits));
          store_bucket(b << kBucketBits, w);
        }
        head += n; cnt -= n;
      }
    }
  };
  // LDS -> table; tile and counters are cleared behind it for the next unit
  auto store_tile = [&](SLOT* gt, uint32_t t) {
#pragma unroll
    for(uint32_t k = 0; k < NBK; ++k) {
      const uint32_t b = threadIdx.x + k * BLOCK;
#pragma unroll
      for(uint32_t q = 0; q < kBV; ++q) {
        const uint4 v = *reinterpret_cast<const uint4*>(s_tile + ((size_t)b << kBucketBits) + q * kVec);
        *reinterpret_cast<uint4*>(gt + ((size_t)b << kBucketBits) + q * kVec) = v;
        *reinterpret_cast<uint4*>(s_tile + ((size_t)b << kBucketBits) + q * kVec) = make_uint4(0, 0, 0, 0);      // the next unit starts from an empty tile
      }
    }
    zero_counters();
    if(threadIdx.x == 0) {
      if(TPB == 2) *reinterpret_cast<uint16_t*>(T.dirty + tile0 + 2 * (uint64_t)t) = 0x0101;
      else T.dirty[tile0 + t] = 1;
    }
  };
  // C for one item: find the key or a free slot, from its home bucket on, kQueueLook buckets per look (one LDS round trip).
  // M has run: no bucket holds a tag twice, and nothing here can change that -- slots only ever go from empty to one key,
  // and two lanes with the same new key go for the same first empty slot: the loser's compare-and-swap returns the
  // winner's entry and it adds to that.  So a full bucket is full for good, an add lands on the key it was meant for, and
  // a look that has gone stale can only lead to a compare-and-swap that fails.
  // What this phase costs is its chain of dependent LDS round trips, not its instructions: 3.85 % of the metric's items come
  // here, and compiling the insert out takes T from 27.0 to 18.9 ms.  Round 5 cut the chain -- the home bucket of a queued
  // item is full of other keys nearly always, so one bucket per look meant two looks for every item; a lost slot meant
  // a new look, now the next candidate of the same look -- 27.0 -> 24.8 ms.  Measured and not kept: four buckets per look
  // (27.3: registers), the eight waves' queues dealt out as one list (25.1-25.4), s_setprio around the phase (25.6),
  // waits between the rank adds of A to keep the LDS queue short (24.9).
  auto slow_insert = [&](ITEM x, uint64_t unit_slot0) {
    constexpr uint32_t LK = kQueueLook;
    const SLOT low = occ | (SLOT)((uint64_t)x & (g.occ_bit - 1)), neww = inc | low;
    const uint32_t h = home_of(x), hbase = h & ~tmask, hb = h & tmask & ~(kB - 1u);
    // the walk ends where every other path's does (T.max_probe: table_add, lookups, update_add, the direct inserts) -- a key
    // placed further from home would be in the table and invisible to them (round-3 advisor finding)
    const uint32_t max_steps = (T.max_probe >> kBucketBits) + 1;
    for(uint32_t step = 0; step < max_steps; ) {
      SLOT w[LK][kB];
#pragma unroll
      for(uint32_t k = 0; k < LK; ++k) load_bucket(hbase + ((hb + ((step + k) << kBucketBits)) & tmask), w[k]);
      uint32_t hitm = 0, empm = 0;
#pragma unroll
      for(uint32_t k = 0; k < LK; ++k)
#pragma unroll
        for(uint32_t i = 0; i < kB; ++i) {
          hitm |= (uint32_t)((w[k][i] & lmask) == low) << (kB * k + i);
          empm |= (uint32_t)(w[k][i] == 0) << (kB * k + i);
        }
      // candidates in probe order: the first slot that holds the key or nothing.  A compare-and-swap that fails says what
      // the slot holds now -- the key (a lane with the same new key got there first: add to it), or another key: then the
      // next candidate of the same look is still good (what was seen full stays full, and the key cannot have gone in
      // behind a slot that was empty when it did), so losing a slot costs one more LDS trip and not a new look.
      for(uint32_t cand = hitm | empm; cand; cand &= cand - 1) {
        const uint32_t j = (uint32_t)__ffs((int)cand) - 1u;
        if(step + (j >> kBucketBits) >= max_steps) { step = max_steps; break; }
        const uint32_t at = hbase + ((hb + (step << kBucketBits) + j) & tmask);
        bool add = (hitm >> j) & 1;
        if(!add) {
          const SLOT was = atomicCAS(&s_tile[at], (SLOT)0, neww);
          if(was == 0) return;
          add = (was & lmask) == low;
        }
        if(add) {
          if(RETURNING) {
            const SLOT prev = atomicAdd(&s_tile[at], inc);
            if(((uint64_t)prev >> cshift) + 1 > g.cnt_max) ovf_add_call(T.ovf_key, T.ovf_cnt, T.ovf_mask, T.counters, unit_slot0 + at, 1);
          } else atomicAdd(&s_tile[at], inc);
          return;
        }
      }
      step += LK;
    }
    atomicAdd((unsigned long long*)&T.counters[CTR_FULL], 1ull);
  };

  [[maybe_unused]] PhaseClk pc;
  // A, first half: the returning adds of a round's items (vm: which of it[] hold one)
  auto rank_request = [&](const ITEM (&it)[NP], uint32_t vm, uint32_t (&old)[NP]) {
#pragma unroll
    for(int r = 0; r < NP; ++r)
      if((vm >> r) & 1) { const uint32_t b = home_of(it[r]) >> kBucketBits; old[r] = atomicAdd(&s_cnt[b >> 1], 1u << ((b & 1) * 16)); }
  };
  // A, second half, then M and C.  `again`: where the round's items came from (held-back items are read again from there:
  // indexing the register array would put it in scratch).  after_a(): it[] is dead from there on.
  [[maybe_unused]] uint32_t smp_items = 0, smp_queued = 0;   // SAMPLE: this lane's items / this wave's items past rank 3
  auto place_round = [&](ITEM (&it)[NP], uint32_t vm, const uint32_t (&old)[NP], uint64_t unit_slot0, const ITEM* again, auto&& after_a) {
    uint32_t qn = 0, pend = 0;                                 // qn: items of this wave past rank 3 (wave-uniform)
    [[maybe_unused]] uint32_t ovm = 0;
#pragma unroll
    for(int r = 0; r < NP; ++r) JF_OPAQUE(it[r]);
#pragma unroll
    for(int r = 0; r < NP; ++r) {
      bool ov = false;
      if((vm >> r) & 1) {
        const uint32_t b = home_of(it[r]) >> kBucketBits;
        const uint32_t rank = (old[r] >> ((b & 1) * 16)) & 0xFFFFu;
        if(rank < kB) s_tile[(b << kBucketBits) + rank] = inc | occ | (SLOT)((uint64_t)it[r] & (g.occ_bit - 1));
        else ov = true;
      }
      if constexpr(HEAVY) { if(ov) ovm |= 1u << r; }
      else {
        const unsigned long long m = __ballot(ov);
        if(ov) {
          const uint32_t at = qn + (uint32_t)__popcll(m & below);
          if(at < qcap) s_q[at] = it[r]; else pend |= 1u << r;
        }
        qn += (uint32_t)__popcll(m);
      }
    }
    if constexpr(SAMPLE) { smp_items += (uint32_t)__popc(vm); smp_queued += qn; }
    if constexpr(!HEAVY) after_a();
    lds_barrier();
    JF_PHASE(pc, 2);
    merge_tile(unit_slot0);
    lds_barrier();
    JF_PHASE(pc, 6);
    uint32_t nq = qn < qcap ? qn : qcap;                       // wave-uniform
    if constexpr(HEAVY) {
      // ---- C0: the items past rank 3, in bulk (it[] is still this round's)
      uint32_t q2 = 0, carry = 0;
      constexpr int CB = kB == 4 ? 6 : 3;                      // (buckets fetched together: at most 24 slot words)
#pragma unroll
      for(int r0 = 0; r0 < NP; r0 += CB) {
        SLOT wb[CB][kB];
#pragma unroll
        for(int k = 0; k < CB; ++k)
          if(r0 + k < NP && ((ovm >> (r0 + k)) & 1)) load_bucket(home_of(it[r0 + k < NP ? r0 + k : 0]) & ~(kB - 1u), wb[k]);
#pragma unroll
        for(int k = 0; k < CB; ++k) {
          if(r0 + k >= NP) continue;
          const int r = r0 + k;
          bool miss = false;
          if((ovm >> r) & 1) {
            const ITEM x = it[r];
            const SLOT low = occ | (SLOT)((uint64_t)x & (g.occ_bit - 1));
            const uint32_t bs = home_of(x) & ~(kB - 1u);
            int hit = -1;
#pragma unroll
            for(int i = (int)kB - 1; i >= 0; --i) if((wb[k][i] & lmask) == low) hit = i;
            if(hit >= 0) {
              if(RETURNING) {
                const SLOT prev = atomicAdd(&s_tile[bs + hit], inc);
                if(((uint64_t)prev >> cshift) + 1 > g.cnt_max) carry |= 1u << r;
              } else atomicAdd(&s_tile[bs + hit], inc);
            } else miss = true;
          }
          const unsigned long long m2 = __ballot(miss);
          if(miss) {
            const uint32_t at = q2 + (uint32_t)__popcll(m2 & below);
            if(at < qcap) s_q[at] = it[r]; else pend |= 1u << r;
          }
          q2 += (uint32_t)__popcll(m2);
        }
      }
      (void)__ballot(true);                                    // (the wave's queue entries are all written before any lane reads them: lockstep on the device, a rendezvous in the host emulation)
      after_a();
      while(carry) {                                           // rare: an add above left its count field
        const uint32_t r = (uint32_t)__ffs((int)carry) - 1u; carry &= carry - 1;
        const ITEM x = again[r * BLOCK + threadIdx.x];
        const SLOT low = occ | (SLOT)((uint64_t)x & (g.occ_bit - 1));
        const uint32_t bs = home_of(x) & ~(kB - 1u);
        SLOT w[kB];
        load_bucket(bs, w);
        uint32_t hit = 0;
#pragma unroll
        for(int i = (int)kB - 1; i >= 1; --i) if((w[i] & lmask) == low) hit = (uint32_t)i;      // (the entry is there and does not move)
        if((w[0] & lmask) == low) hit = 0;
        ovf_add_call(T.ovf_key, T.ovf_cnt, T.ovf_mask, T.counters, unit_slot0 + bs + hit, 1);
      }
      nq = q2 < qcap ? q2 : qcap;
    }
    // ---- C: the queues, then what lanes held back -- staged through the queue too, so that the loop reads LDS and nothing
    // else.  (A loop over "s_q[i] or again[...]" makes the item a FLAT load, and the wait behind a flat load is for every
    // outstanding memory operation: the next unit's items, requested in after_a(), would be waited for right here.)
    for(;;) {
      for(uint32_t i = lane; i < nq; i += 64) slow_insert(s_q[i], unit_slot0);
      const unsigned long long m = __ballot(pend != 0);      // (wave-uniform; nearly always 0)
      if(!m) break;
      if(pend) {
        const uint32_t r = (uint32_t)__ffs((int)pend) - 1u;
        pend &= pend - 1;
        s_q[(uint32_t)__popcll(m & below)] = again[r * BLOCK + threadIdx.x];
      }
      nq = (uint32_t)__popcll(m);
      (void)__ballot(true);                                  // (written before read: lockstep on the device, a rendezvous in the host emulation)
    }
    JF_PHASE(pc, 5);
    lds_barrier();
    JF_PHASE(pc, 3);
  };
  zero_counters();
  zero_tile();
  lds_barrier();
  // One loop over CHUNKS: a unit's items in rounds of NP x BLOCK (nearly always one).  Offsets are fetched two units ahead, and
  // a chunk's items are requested as soon as the previous chunk's are placed (same registers): they travel during that
  // chunk's queue phase and store.
  const uint64_t* off = S.off[0];
  const uint32_t sh = S.sh[0];                              // 0: packed offsets off[t], off[t + 1]; 1: pairs (begin, end), items may be holes
  const bool holes = sh != 0;
  const ITEM* src = reinterpret_cast<const ITEM*>(S.items[0]);
  const uint32_t G = gridDim.x;
  constexpr uint64_t kRound = (uint64_t)NP * BLOCK;
  uint32_t t = blockIdx.x;
  uint64_t a0 = 0, b0 = 0, a1 = 0, b1 = 0; uint32_t d0 = 0, d1 = 0;
  ITEM cur[NP];
  auto fetch = [&](uint64_t a, uint64_t b) {                // items src[a .. min(b, a + NP x BLOCK)): block-uniform base + 32-bit index
    const ITEM* ub = src + a;
    const uint32_t n = (uint32_t)((b - a) < kRound ? (b - a) : kRound);
    if(b <= a) return;                                      // (block-uniform; such a unit is skipped)
    uint32_t tid = threadIdx.x;
    JF_OPAQUE(tid);                                         // (or the item indices are hoisted out of the loop and spilled)
#pragma unroll
    for(int r = 0; r < NP; ++r) {                           // unconditional loads at clamped indices: no branch per item
      const uint32_t i = (uint32_t)r * BLOCK + tid;
      cur[r] = ub[i < n ? i : n - 1];
    }
  };
  if(t < n_tiles) { a0 = off[(size_t)t << sh]; b0 = off[((size_t)t << sh) + 1]; d0 = unit_dirty(t); }
  if(t + G < n_tiles) { a1 = off[(size_t)(t + G) << sh]; b1 = off[((size_t)(t + G) << sh) + 1]; d1 = unit_dirty(t + G); }
  fetch(a0, b0);
  uint64_t c0 = a0;                                          // start of the current chunk inside [a0, b0)
  while(t < n_tiles) {
    if(b0 <= a0) {                                           // nothing for this unit (block-uniform): next one
      uint64_t a2 = 0, b2 = 0; uint32_t d2 = 0;
      if(t + 2 * G < n_tiles) { a2 = off[(size_t)(t + 2 * G) << sh]; b2 = off[((size_t)(t + 2 * G) << sh) + 1]; d2 = unit_dirty(t + 2 * G); }
      t += G; a0 = a1; b0 = b1; d0 = d1; a1 = a2; b1 = b2; d1 = d2; c0 = a0;
      fetch(a0, b0);
      continue;
    }
    const bool first = c0 == a0, last = c0 + kRound >= b0;
    const uint64_t unit_slot0 = (tile0 + (uint64_t)TPB * t) << kMaxTileBits;
    SLOT* gt = gslots + unit_slot0;
    JF_PHASE(pc, 0);
    const uint32_t n0 = (uint32_t)((b0 - c0) < kRound ? (b0 - c0) : kRound);
    uint32_t vm = 0, tid = threadIdx.x;
    JF_OPAQUE(tid);
#pragma unroll
    for(int r = 0; r < NP; ++r)
      if((uint32_t)r * BLOCK + tid < n0 && !(holes && cur[r] == hole)) vm |= 1u << r;      // (what a clamped load fetched is not an item)
    // a further round of the same unit (more items than one round holds: skewed input) starts from the tile as the previous round
    // stored it -- merged, compacted, counted again on the way in -- by the lanes that stored it
    const uint32_t d_eff = first ? d0 : (TPB == 2 ? 0x0101u : 1u);
    if(d_eff) { load_tile(gt, d_eff); lds_barrier(); }       // (otherwise the previous store left tile and counters zeroed)
    uint32_t old[NP];
    rank_request(cur, vm, old);
    JF_PHASE(pc, 1);
    uint64_t a2 = 0, b2 = 0; uint32_t d2 = 0;
    if(last && t + 2 * G < n_tiles) { a2 = off[(size_t)(t + 2 * G) << sh]; b2 = off[((size_t)(t + 2 * G) << sh) + 1]; d2 = unit_dirty(t + 2 * G); }
    place_round(cur, vm, old, unit_slot0, src + c0, [&] { if(last) fetch(a1, b1); else fetch(c0 + kRound, b0); });
    store_tile(gt, t);                            // (every round ends on a barrier)
    lds_barrier();
    JF_PHASE(pc, 4);
    if(last) { t += G; a0 = a1; b0 = b1; d0 = d1; a1 = a2; b1 = b2; d1 = d2; c0 = a0; }
    else c0 += kRound;
  }
  JF_PHASE_FLUSH(pc, 16);
  if constexpr(SAMPLE) {
    uint32_t w = smp_items;
    for(int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
    if(lane == 0) {
      atomicAdd((unsigned long long*)&T.counters[CTR_T_ITEMS], (unsigned long long)w);
      atomicAdd((unsigned long long*)&T.counters[CTR_T_QUEUED], (unsigned long long)smp_queued);
    }
  }
}

}  // namespace jfgpu
