// jellyfish_amd/csrc/kernels_p1ring.hip.hpp -- stage P1 of the partitioned insert path for 32-bit items (gfx950).
//
// Round 2's single-pass P1 counting-sorted every chunk of 16 Ki positions by bucket in LDS (histogram, scan, ranked
// scatter, read-back by bucket, placement table) and wrote runs of ~16 items wherever the chunk's run of a bucket
// happened to start: seven barriers per chunk on one workgroup per CU, and 64-byte runs straddling 128-byte lines
// (1.6 x write amplification, profiles/r02_traffic_C2.json).  Since round 3 the sort is gone: every bucket owns a RING of
// 32 items in LDS (1024 x 128 B), a k-mer's item is appended with one returning ds_add on the bucket's fill word and one
// store, and after a round of 8 Ki positions every bucket that has 16 items together emits them as one aligned 64-byte
// unit (tools/probes/scatter_write_probe.hip: aligned 64-byte runs travel at twice the rate of straddling ones).  Region
// format, reservations of kGran items and holes are exactly the granule kernels' (kernels_part.hip.hpp), so P2 and the
// tile kernel read the output as before.  Round 4 rebuilt the kernel around what its counters said (see below).
#pragma once
#include <type_traits>
#include "kernels_part.hip.hpp"

namespace jfgpu {

// s_setprio: the issue priority of this wave among the waves of its SIMD (0 .. 3)
template <int P> __device__ __forceinline__ void wave_prio() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_setprio(P);
#endif
}

// admission mask of count --bc (bit j <-> position j) in the bit order of the validity masks (bit 15 - j <-> position j)
__device__ __forceinline__ uint32_t adm_to_vmask(uint32_t adm) {
  uint32_t r = 0;
#pragma unroll
  for(int j = 0; j < 16; ++j) r |= ((adm >> j) & 1u) << (15 - j);
  return r;
}


// ---- round 4: rings owned lane by lane, stragglers on a list ------------------------------------------------------------
// SQ counters of the round-3 kernel (profiles/r04_sq_counters_p1.txt): 13.4 G vector instructions per 10 Gbp = 85 per
// position and lane, the SIMDs issuing them 64 % of the time; 10 LDS instructions per position, the LDS array busy 41 % of
// the CU cycles plus 22 % in bank conflicts.  Both units are two-thirds busy and a wave's instruction stream alternates
// between them, so the kernel moves only when BOTH get less to do.  What was done, with the P1 time of the 10 Gbp job:
//   39.1 ms  round 3
//   35.8     pack16 four characters at a time (kmer_core.hpp: 16 -> 5 vector instructions per character)
//   35.2     this kernel:
//   * the flush is every owner lane's own business: lane t reads bucket t's units out of its ring (four 16-byte reads) and
//     writes them to the region (four 16-byte stores) -- no ballots, no shuffles, no four-lane teams.  A ring starts life
//     at position 4 t (mod 32), so the 64 owners of a wave read 64 different bank groups.
//   * nothing is inserted from inside the loop: an item that finds its ring full, a run of identical k-mers, the item
//     that looks like a hole go on the workgroup's STRAGGLER list (an LDS counter, a plain global store) and
//     p1_stragglers_kernel appends them to their regions afterwards, before the regions' bounds are taken.  Round 3
//     inserted them on the spot with global atomics: a few microseconds during which the wave, and at the next barrier
//     the workgroup, stood still -- and 121 K tiles per job were dirtied before the tile stage ran, which then had to
//     read them (T 28.0 -> 26.9 ms).
//   * a flush needs no quiescence after it: a ring slot that holds no item holds the HOLE marker (an item never equals
//     it), a unit is complete exactly when none of its 16 slots reads as a hole, complete units go out, their slots are
//     reset to holes and only then released (compare-and-swap on the fill word: ring position advanced, count reduced),
//     so the next round's appends of other waves may run beside it.  An append that finds its ring full (rank >= 32)
//     stays counted in the fill word as a "ghost" until the owner's next release drops it (min(count, 32) is what the
//     ring really holds).  ONE barrier per round is left, before the flush, so that everything due has landed.
//   * fewer instructions per position: the second sweep stores without a branch per item (positions without an item
//     store into 32 dump slots behind the rings), the rolling forward / reverse-complement words move by funnel shifts
//     on their two dwords, the six table words meet in v_bitop3_b32.
// Measured and dropped (profiles/r04_p1_experiments.log): no barrier at all (a unit whose newest item is still on its
// way -- an append stores its item only after all of its round's returning adds are back -- is put off by a round, its
// ring overflows: 9 % of the items became stragglers, 130 ms); halo lanes instead of staging the neighbours' bases
// through LDS (lanes 0 and 1 of a wave load the 32 bases before it and produce nothing: no staging barriers, but 3 % more
// work: 37.4 ms); a second barrier after the flush (36.2 with LDS staging); half the table look-ups, as a sensitivity
// test with wrong results (-7 %: neither unit alone is the limit).
constexpr uint32_t kStragPerBlock = 16384;                                     // entries of a workgroup's straggler list

// ---- the ring machinery, shared by the P1 kernels of every item width ---------------------------------------------------
// A ring is 128 bytes of LDS whatever the item (32 four-byte, 16 eight-byte or 8 sixteen-byte items), a unit half of it
// (one aligned 64-byte run in the bucket's region), and it is read and reset in 16-byte chunks.  RB: the ring's bytes
// (round 6: the Bloom counter's P1b takes rings of 256 bytes -- 64 cell updates, four units -- for its 512 buckets).
template <typename ITEM, uint32_t RB = 128> struct Ring {
  static constexpr uint32_t kSlots = RB / sizeof(ITEM);
  static constexpr uint32_t kUnit = 64 / sizeof(ITEM);
  static constexpr uint32_t kChunk = 16 / sizeof(ITEM);                        // items per 16-byte chunk
  static constexpr uint32_t kFull = 0xFFFFu & ~(kSlots - 1);                   // bits of a rank that say "the ring is full"
  static constexpr uint32_t kWords = sizeof(ITEM) <= 4 ? 1 : 1 + sizeof(ITEM) / 8;   // 64-bit words of a straggler entry
  // Bucket b's ring starts life at ring position b chunks (mod the ring): the 64 owners of a wave then read 16-byte
  // chunks of 64 different bank groups at every step, with no address arithmetic spent on it.
  __device__ static uint32_t first_fill(uint32_t b) { return ((kChunk * b) & (kSlots - 1)) << 16; }
};

// 0xFFFFFFFF exactly when the chunk holds a hole (the all-ones item)
template <typename ITEM> __device__ __forceinline__ uint32_t chunk_hole_key(const uint4& v) {
  if constexpr(sizeof(ITEM) == 4) { const uint32_t a = v.x > v.y ? v.x : v.y, c = v.z > v.w ? v.z : v.w; return a > c ? a : c; }
  else if constexpr(sizeof(ITEM) == 8) { const uint32_t a = v.x & v.y, c = v.z & v.w; return a > c ? a : c; }
  else return v.x & v.y & v.z & v.w;
}
template <typename ITEM, typename F> __device__ __forceinline__ void chunk_items(const uint4& v, F&& f) {
  if constexpr(sizeof(ITEM) == 4) { f((ITEM)v.x); f((ITEM)v.y); f((ITEM)v.z); f((ITEM)v.w); }
  else if constexpr(sizeof(ITEM) == 8) { f((ITEM)(((uint64_t)v.y << 32) | v.x)); f((ITEM)(((uint64_t)v.w << 32) | v.z)); }
  else f((ITEM)(((unsigned __int128)(((uint64_t)v.w << 32) | v.z) << 64) | (((uint64_t)v.y << 32) | v.x)));
}

// entry of a straggler list: occurrences (8 bits) << 56 | bucket (24 bits) << 32 (| the item, when it has 32 bits), then the
// item's words
template <typename ITEM> __device__ __forceinline__ void strag_store(uint64_t* rec, uint32_t b, ITEM item, uint32_t cnt) {
  const uint64_t meta = ((uint64_t)(cnt < 255u ? cnt : 255u) << 56) | ((uint64_t)(b & 0xFFFFFFu) << 32);
  if constexpr(sizeof(ITEM) == 4) rec[0] = meta | (uint32_t)item;
  else if constexpr(sizeof(ITEM) == 8) { rec[0] = meta; rec[1] = (uint64_t)item; }
  else { rec[0] = meta; rec[1] = (uint64_t)item; rec[2] = (uint64_t)(item >> 64); }
}
template <typename ITEM> __device__ __forceinline__ void strag_load(const uint64_t* rec, uint32_t& b, ITEM& item, uint32_t& cnt) {
  const uint64_t meta = rec[0];
  b = (uint32_t)(meta >> 32) & 0xFFFFFFu; cnt = (uint32_t)(meta >> 56);
  if constexpr(sizeof(ITEM) == 4) item = (ITEM)(uint32_t)meta;
  else if constexpr(sizeof(ITEM) == 8) item = (ITEM)rec[1];
  else item = (ITEM)(((unsigned __int128)rec[2] << 64) | rec[1]);
}

// an owner lane's books: its place in the region (gpos .. gpos + room of the current reservation), `nxt` the reservation
// asked for in advance (its answer is first looked at a round later), the items it stored
struct RingBooks { uint32_t gpos = 0, room = 0, nxt = 0, stored = 0; bool nxt_asked = false, exhausted = false; };

// What bucket t has complete goes out (owner lanes only).  `all`: the kernel's last call, after a barrier -- every append
// has landed, and the partial last unit goes out too (the slots behind its items are holes already).
// OWNED: the region has one writer (this lane) -- B.room is what is left of it, nothing is reserved.
template <typename ITEM, bool OWNED = false, uint32_t RB = 128, typename STRAG>
__device__ __forceinline__ void ring_flush(ITEM* s_ring, uint32_t* s_fill, uint32_t t, bool all, RingBooks& B, ITEM* my_region, uint32_t cap,
                                           unsigned int* gcur, unsigned int* gshort, STRAG&& straggler) {
  using R = Ring<ITEM, RB>;
  const ITEM hole = (ITEM)~(ITEM)0;
  const uint32_t w = s_fill[t];
  uint32_t cnt = w & 0xFFFFu; if(cnt > R::kSlots) cnt = R::kSlots;
  const uint32_t rb = (w >> 16) & (R::kSlots - 1);
  uint32_t units = cnt / R::kUnit; if(all && (cnt % R::kUnit)) ++units;
  uint32_t nout = 0;
  for(uint32_t s = 0; s < units; ++s) {
    uint4 v[4];
    const uint32_t s0 = (rb + s * R::kUnit) & (R::kSlots - 1);
#pragma unroll
    for(int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const uint4*>(s_ring + t * R::kSlots + ((s0 + R::kChunk * q) & (R::kSlots - 1)));
    if(!all) {
      uint32_t mx = 0;
#pragma unroll
      for(int q = 0; q < 4; ++q) { const uint32_t a = chunk_hole_key<ITEM>(v[q]); mx = mx > a ? mx : a; }
      if(mx == 0xFFFFFFFFu) break;                                 // an item of this unit is still on its way: next time
    }
    if constexpr(!OWNED) {
    if(B.room == 0 && B.nxt_asked) {                               // take the reservation asked for earlier
      if((uint64_t)B.nxt + kGran <= cap) { B.gpos = B.nxt; B.room = kGran; }
      else { B.exhausted = true; if(B.nxt < cap) atomicMax(&gshort[t], cap - B.nxt); }     // (everything below nxt was handed out)
      B.nxt_asked = false;
    }
    if(B.room == 0 && !B.exhausted) {                              // none in hand, none asked for: ask now and wait (rare)
      const uint32_t r0 = atomicAdd(&gcur[t], kGran);
      if((uint64_t)r0 + kGran <= cap) { B.gpos = r0; B.room = kGran; }
      else { B.exhausted = true; if(r0 < cap) atomicMax(&gshort[t], cap - r0); }
    }
    }
    if(B.room) {
      uint4* dst = reinterpret_cast<uint4*>(my_region + B.gpos);
#pragma unroll
      for(int q = 0; q < 4; ++q) dst[q] = v[q];
      const uint32_t real = cnt - s * R::kUnit < R::kUnit ? cnt - s * R::kUnit : R::kUnit;
      B.gpos += R::kUnit; B.room -= R::kUnit; B.stored += real;
    } else {                                                       // the region is exhausted (skewed input): the list
#pragma unroll
      for(int q = 0; q < 4; ++q) chunk_items<ITEM>(v[q], [&](ITEM x) { if(x != hole) straggler(t, x, 1u); });
    }
#pragma unroll
    for(int q = 0; q < 4; ++q)
      *reinterpret_cast<uint4*>(s_ring + t * R::kSlots + ((s0 + R::kChunk * q) & (R::kSlots - 1))) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    ++nout;
  }
  // keep one reservation in hand whenever the current one cannot take a ring's worth: its round trip to L2 hides
  // behind the next round
  if constexpr(!OWNED) { if(!all && !B.nxt_asked && !B.exhausted && B.room < R::kSlots) { B.nxt = atomicAdd(&gcur[t], kGran); B.nxt_asked = true; } }
  // the release: ring position past what went out, the count without it -- and without the ghosts
  uint32_t expect = w;
  while(nout || (expect & 0xFFFFu) > R::kSlots) {
    uint32_t c = expect & 0xFFFFu; if(c > R::kSlots) c = R::kSlots;
    const uint32_t taken = nout * R::kUnit < c ? nout * R::kUnit : c;                       // (`all`: the partial unit takes what is there)
    const uint32_t neww = ((((expect >> 16) + nout * R::kUnit) & 0xFFFFu) << 16) | (c - taken);
    const uint32_t old = atomicCAS(&s_fill[t], expect, neww);
    if(old == expect) break;
    expect = old;                                                   // somebody appended meanwhile: the same release on the newer word
  }
}

// rings, fill words, the list's counter: before the first append (the caller's barrier follows)
template <typename ITEM, uint32_t RB = 128>
__device__ __forceinline__ void ring_init(ITEM* s_ring, uint32_t* s_fill, uint32_t nb, uint32_t* s_nstrag) {
  using R = Ring<ITEM, RB>;
  for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) s_fill[j] = R::first_fill(j);
  uint4* r4 = reinterpret_cast<uint4*>(s_ring);
  for(uint32_t j = threadIdx.x; j < nb * (RB / 16) + RB / 16; j += blockDim.x) r4[j] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);      // (+ the dump slots)
  if(threadIdx.x == 0) *s_nstrag = 0;
}

// kernel end (owner lanes): what is left of the reservations becomes holes; the exact count of the bucket goes to tot
template <typename ITEM>
__device__ __forceinline__ void ring_finish(const RingBooks& B, uint32_t t, ITEM* my_region, uint32_t cap, unsigned int* gshort, unsigned long long* tot) {
  const ITEM hole = (ITEM)~(ITEM)0;
  for(uint32_t r = 0; r < B.room; ++r) my_region[B.gpos + r] = hole;
  if(B.nxt_asked) {
    if((uint64_t)B.nxt + kGran <= cap) { for(uint32_t r = 0; r < kGran; ++r) my_region[B.nxt + r] = hole; }
    else if(B.nxt < cap) atomicMax(&gshort[t], cap - B.nxt);
  }
  if(tot && B.stored) atomicAdd(&tot[t], (unsigned long long)B.stored);
}

// What p1_ring_kernel / p1_stragglers_kernel do with an entry that cannot be stored in a region: DIRECT(bucket, item,
// occurrences).  kCountsDirect: the calls are counted in the table's CTR_DIRECT.
// (bucket, item, occurrences) of a one-word key into the table with global atomics (item_direct_call)
struct OneWordDirect {
  static constexpr bool kCountsDirect = true;
  const DevTable* Tm; uint32_t b2; int returning;
  __device__ void operator()(uint32_t b, uint64_t item, uint32_t cnt) const { item_direct_call(Tm, b2, b, item, cnt, returning); }
};
// The sending side of the multi-GPU exchange (abi_comm.inl): nothing may be inserted here -- the k-mers belong to other
// GPUs -- so such entries go on the list of stragglers every rank receives (bucket << 32 | item, once per occurrence).
struct RouteListDirect {
  static constexpr bool kCountsDirect = false;
  StragList SL;
  __device__ void operator()(uint32_t b, uint64_t item, uint32_t cnt) const {
    for(uint32_t i = 0; i < cnt; ++i) { const unsigned long long at = atomicAdd(SL.n, 1ull); if(at < SL.cap) SL.rec[at] = ((uint64_t)b << 32) | (uint32_t)item; }
  }
};

// ---- one-word keys, 4-byte items (k <= 21 at the metric's geometry) -----------------------------------------------------
// CANON: 0 forward k-mers, 1 canonical, 2 decided at run time (the table's flag).  A round is 8 positions per lane.
// The kernel is written over the item type, and was measured with the wider ones (profiles/r04_c5_ring_experiment.log,
// r04_k31_stage_times.txt): 8-byte items (k = 22 .. 32: rings of 16, rounds of 4 positions) 29.6 ms per 5 Gbp at k = 31
// against 28.4 for the sort-based p1_granule64_kernel before this round's pack16; 16-byte items (k = 63: rings of 8,
// units of 4) 130 - 152 ms against 106.  A ring of 128 bytes holds too few wide items: either the rounds get short (a
// barrier and a flush every two or four positions) or 0.6 % of the items overflow.  Those widths keep the sort.
// NB: key bytes of the table hash compiled in (0: decided at run time); kHashXS / kHashXSLow: the table's matrix is the
// xor-shift one and is evaluated in registers (on the key's two dwords for 32 < lsize_g, else in 64-bit arithmetic).
template <typename ITEM, bool BLOOM, int NB, int CANON, typename DIRECT = OneWordDirect>
__global__ __launch_bounds__(kPBlock) void p1_ring_kernel(DevTable T, DIRECT D, PartGeom P, const uint8_t* __restrict__ base,
                                                          int64_t lo, int64_t hi, uint32_t cap,
                                                          unsigned int* __restrict__ gcur,
                                                          unsigned long long* __restrict__ tot,
                                                          ITEM* __restrict__ out,
                                                          uint64_t* __restrict__ strag, uint32_t* __restrict__ strag_n) {
  using R = Ring<ITEM>;
  constexpr int RP = sizeof(ITEM) == 4 ? 8 : 4;                     // positions per lane and round
  JF_DYN_LDS(s_dyn);
  ITEM* s_ring = reinterpret_cast<ITEM*>(s_dyn);                    // [nb][R::kSlots], then 128 bytes of dump slots
  __shared__ uint64_t s_fwd[NB < 0 ? 1 : 8 * 256];                 // (NB < 0: the xor-shift matrix, evaluated in registers -- no tables)
  __shared__ uint32_t s_fill[kGranMaxB + 32];                      // (+ 32 spare words: where positions without a k-mer append)
  __shared__ uint32_t s_nstrag;
  __shared__ uint32_t s_codes[kPBlock + 2];
  __shared__ uint32_t s_inv[kPBlock + 2];
  const TableGeom& g = T.g;
  const uint32_t nb = 1u << P.b1;
  const uint32_t t = threadIdx.x, lane = t & 63;
  const bool owner = t < nb;                                       // this lane keeps bucket t's books
  const ITEM hole = (ITEM)~(ITEM)0;
  if constexpr(NB >= 0) load_tables_lds(s_fwd, T.fwd_tbl, g.nbytes);
  ring_init<ITEM>(s_ring, s_fill, nb, &s_nstrag);
  const uint32_t dump = nb * R::kSlots + (lane & (R::kSlots - 1));  // behind the rings: where the stores of positions without an item go
  [[maybe_unused]] const uint32_t xs_hi_mask = g.lsize_g > 32 ? (g.lsize_g >= 64 ? 0xFFFFFFFFu : ((1u << (g.lsize_g - 32)) - 1u)) : 0u;
  unsigned int* const gshort = gcur + nb;
  uint64_t* const my_strag = strag + (size_t)blockIdx.x * kStragPerBlock * R::kWords;
  const uint32_t k = g.k, bshift = g.lsize_l - P.b1;
  const uint32_t rc_shift = 2 * (k - 1);
  // 32-bit items in 32-bit arithmetic: item = (rest << rem_bits) | rem, rest_shift + rem_bits <= 32
  const uint32_t rest_mask = P.rest_shift >= 32 ? 0xFFFFFFFFu : ((1u << P.rest_shift) - 1u);
  RingBooks B;
  if(owner) { B.nxt = atomicAdd(&gcur[t], kGran); B.nxt_asked = true; }
  ITEM* const my_region = out + (uint64_t)t * cap;
  uint32_t my_mers = 0, my_direct = 0;

  // not for the rings: on the workgroup's list (p1_stragglers_kernel takes it from there); a full list (an input that
  // sends everything to a few buckets) falls back to DIRECT on the spot (the table's global claim: slow, never wrong)
  auto straggler = [&](uint32_t b, ITEM item, uint32_t cnt) {
    const uint32_t at = atomicAdd(&s_nstrag, 1u);
    if(at < kStragPerBlock) strag_store<ITEM>(my_strag + (size_t)at * R::kWords, b, item, cnt);
    else { D(b, (uint64_t)item, cnt); ++my_direct; }
  };

  const int64_t n_tiles = (hi + kPTilePos - 1) / kPTilePos;
  TileRaw Rw = tile_fetch(base, (int64_t)blockIdx.x * kPTilePos, lo, hi);
  lds_barrier();                                                   // tables, fill words and holes are in place
  [[maybe_unused]] PhaseClk pc;
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    // (No barrier before the staging: the readers of the previous tile's words read them a round barrier ago.  Round 6 also
    // tried staging without any barrier of its own -- the next tile's words published before the barrier that ends this
    // tile's last round, two barriers a tile instead of three: 30.7 -> 34.2 ms.  A sweep that starts right behind a flush,
    // without the staging barrier in between, lets the fast waves append into rings their owners have not released yet:
    // ghosts, stragglers (p1_stragglers_kernel shows up in the trace).  profiles/r06_p1_experiments.log)
    const LaneWords L = tile_stage(Rw, tile * kPTilePos, lo, hi, s_codes, s_inv);     // barrier inside
    Rw = tile_fetch(base, (tile + gridDim.x) * kPTilePos, lo, hi);                     // next tile's bytes travel while this one is worked on
    JF_PHASE(pc, 0);
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
    // RUNS: a k-mer is emitted one position late, when it is known whether the next one repeats it (homopolymers: one entry
    // for the run, with its length): pk / pv / run describe the position before.  Round 6: that logic -- the compare with the
    // k-mer before, the run counter, a ninth entry per round for the tile's last k-mer -- is worth 1.2 ms of the metric's job
    // (profiles/r06_p1_experiments.log: -DJFGPU_P1_NO_RUNS), and a run of k + 1 equal bases (k >= 14) always covers one
    // aligned chunk of eight bases of the lane's own word or of the word before it.  So a wave whose lanes see no such chunk
    // takes the rounds WITHOUT the run logic (every position emits its own k-mer at once, eight entries a round); a wave that
    // sees one (2 % of the waves on random sequence: a chunk of eight equal bases by chance) takes the rounds above as they were.
    auto rounds = [&](auto runs_tag) {
    constexpr bool RUNS = decltype(runs_tag)::value;
    uint64_t pk = 0; uint32_t pv = 0, run = 0;
#pragma unroll 1
    for(int j0 = 0; j0 < kPerLane; j0 += RP) {
      constexpr int NE = RUNS ? RP + 1 : RP;                        // (the +1: the tile's last k-mer, emitted after the loop)
      // ea: the ring's first slot (bucket * slots), ei: item, eo: fill word before the append.  A position without an item
      // keeps (dump, 0): the second sweep stores unconditionally, those stores land in the dump slots
      uint32_t ea[NE], eo[NE]; ITEM ei[NE];
      [[maybe_unused]] uint32_t spm = 0;                            // bit e: entry e is a special one (JFGPU_P1_LATE_SPECIAL)
#pragma unroll
      for(int e = 0; e < NE; ++e) { ea[e] = dump; ei[e] = 0; eo[e] = 0; }
      // No branch around a position's hash and append (round 5: 35.4 -> 33.4 ms; with `if(k-mer to emit) { hash, append }` per
      // position the compiler kept eight separate blocks, each waiting for its own table reads): a position without a k-mer
      // to emit hashes whatever its registers hold and appends to one of 32 spare fill words behind the buckets'; its store
      // goes to the dump slots.
      auto emit = [&](int e, uint64_t key, uint32_t cnt, bool on) {
        uint32_t b; ITEM item;
        if constexpr(NB == kHashXS && sizeof(ITEM) == 4) {
          // the xor-shift matrix on the key's two dwords (the host takes this instantiation for 32 < lsize_g < 2k and
          // bucket shifts below 32): position, bucket and item without a 64-bit shift and without a table
          uint32_t ylo, yhi;
          xs_hash_halves<false>((uint32_t)key, (uint32_t)(key >> 32), xs_hi_mask, ylo, yhi);     // (32-bit items: keys of at most 42 bits, no fold)
          b = funnel_r(yhi, ylo, bshift) & (nb - 1);
          item = ((ylo & rest_mask) << g.rem_bits) | ((uint32_t)(key >> 32) >> (g.lsize_g - 32));
        } else {
          const uint64_t pos = NB < 0 ? xs_hash(key, g.lsize_g, g.key_bits) : hash_tables_t<(NB < 0 ? 0 : NB)>(s_fwd, key, g.nbytes);
          b = (uint32_t)(pos >> bshift) & (nb - 1);
          if constexpr(sizeof(ITEM) == 4) item = (((uint32_t)pos & rest_mask) << g.rem_bits) | (uint32_t)(key >> g.lsize_g);      // (lsize_g <= 2k <= 42)
          else item = make_item<ITEM>(g, P, key, pos & g.local_mask);
        }
        const bool special = on && (item == hole || cnt > 1);        // (it would read as a hole; a run goes in at once)
        const bool normal = on && !special;
#ifndef JFGPU_P1_LATE_SPECIAL
        if(special) straggler(b, item, cnt);                          // (rare; putting these off to one place after the round cost nine registers and 1.6 ms)
        const uint32_t o = atomicAdd(&s_fill[normal ? b : nb + (lane & 31u)], 1u);
        ea[e] = normal ? b * R::kSlots : dump; ei[e] = item; eo[e] = normal ? o : 0u;
#else
        // a special entry rides through the second sweep as a "ghost" of its own bucket: rank field full, its occurrences in
        // the ring-position field; the sweep's rare branch puts it on the list
        const uint32_t o = atomicAdd(&s_fill[normal ? b : nb + (lane & 31u)], 1u);
        spm |= special ? 1u << e : 0u;
        ea[e] = on ? b * R::kSlots : dump; ei[e] = item; eo[e] = normal ? o : special ? (R::kFull | ((cnt < 0xFFFFu ? cnt : 0xFFFFu) << 16)) : 0u;
#endif
      };
      // the round's codes, their complements and its validity bits moved up once (j0 is a run-time 0 or RP), so that every
      // position's field sits at a compile-time offset: one bit-field extract each instead of shift + and by a scalar amount
#ifndef JFGPU_P1_OLD_EXTRACT
      const uint32_t cur_r = L.cur << (2 * j0), ncur_r = ~cur_r, vm_r = vmask << j0;
#endif
#pragma unroll
      for(int e = 0; e < RP; ++e) {
#ifndef JFGPU_P1_OLD_EXTRACT
        const uint32_t c = (cur_r >> (2 * (15 - e))) & 3u, nc = (ncur_r >> (2 * (15 - e))) & 3u;
        const uint32_t v = vm_r & (1u << (15 - e));
#else
        const int j = j0 + e;
        const uint32_t c = (L.cur >> (2 * (15 - j))) & 3u, nc = 3u - c;
        const uint32_t v = vmask & (1u << (15 - j));
#endif
        if constexpr(NB >= 5 || NB == kHashXS) {
          // keys of more than 32 bits (k >= 17): the two dwords by hand -- funnel shifts instead of 64-bit shifts, and the
          // new base of the reverse complement enters the high dword directly (rc_shift >= 32)
          const uint32_t flo = (uint32_t)fw, fhi = (uint32_t)(fw >> 32), rlo = (uint32_t)rc, rhi = (uint32_t)(rc >> 32);
          fw = ((uint64_t)(funnel_r(fhi, flo, 30) & (uint32_t)(g.key_mask >> 32)) << 32) | ((flo << 2) | c);
          rc = ((uint64_t)((rhi >> 2) | (nc << (rc_shift - 32))) << 32) | funnel_r(rhi, rlo, 2);
        } else {
          fw = ((fw << 2) | c) & g.key_mask;
          rc = (rc >> 2) | ((uint64_t)nc << rc_shift);
        }
        const uint64_t key = ((CANON == 1 || (CANON == 2 && g.canonical)) && rc < fw) ? rc : fw;
        if constexpr(RUNS) {
#ifdef JFGPU_P1_NO_RUNS                                            /* ablation (round 6): every occurrence its own item */
          const bool same = false;
#else
          const bool same = v && pv && key == pk;
#endif
          emit(e, pk, run, pv && !same);
          run = same ? run + 1 : 1;
          pk = key; pv = v;
        } else emit(e, key, 1u, v != 0);
      }
      if constexpr(RUNS) { if(j0 + RP >= kPerLane) { emit(NE - 1, pk, run, pv != 0); pv = 0; } }
      // second sweep: the ring stores, once the fill adds are back (not one wait per item), without a branch per item
      uint32_t ghosts = 0;
#pragma unroll
      for(int e = 0; e < NE; ++e) {
        const uint32_t full = eo[e] & R::kFull;                      // rank >= the ring's size: the ring is full
        ghosts |= full;
        const uint32_t at = ea[e] + ((eo[e] + (eo[e] >> 16)) & (R::kSlots - 1));
        s_ring[full ? dump : at] = ei[e];
      }
      if(ghosts) {                                                  // rare: a ghost in its bucket's count until the owner's next release
#pragma unroll 1
#ifndef JFGPU_P1_LATE_SPECIAL
        for(int e = 0; e < NE; ++e) {                               // (entry e by selects on static indices: a dynamic index would put the three arrays in scratch)
          uint32_t a = 0, o = 0; ITEM it = 0;
#pragma unroll
          for(int q = 0; q < NE; ++q) if(q == e) { a = ea[q]; o = eo[q]; it = ei[q]; }
          if(o & R::kFull) straggler(a / R::kSlots, it, 1u);
        }
#else
        for(int e = 0; e < NE; ++e) if(eo[e] & R::kFull) straggler(ea[e] / R::kSlots, ei[e], ((spm >> e) & 1u) ? eo[e] >> 16 : 1u);
#endif
      }
      JF_PHASE(pc, 1);
      lds_barrier();                                               // the round's items have all landed: what is due goes out now
      JF_PHASE(pc, 2);
      // the flush at raised priority (round 6: 30.7 -> 29.8 ms): its few instructions are issued ahead of the sweeps of the
      // waves already in the next round, so a ring is released before much is appended to it again
#ifndef JFGPU_P1_NO_FLUSH_PRIO
      wave_prio<3>();
#endif
      if(owner) ring_flush<ITEM>(s_ring, s_fill, t, false, B, my_region, cap, gcur, gshort, straggler);
#ifndef JFGPU_P1_NO_FLUSH_PRIO
      wave_prio<0>();
#endif
      JF_PHASE(pc, 3);
    }
    };
    bool may_run = k < 14;
    {
      const uint32_t hc = L.cur ^ (L.cur >> 2), hp = L.p1 ^ (L.p1 >> 2);      // zero fields: a base equal to the one before it
      // (a run that ends at the lane's position j covers bases j - k .. j, so its last aligned chunk starts at -8, 0 or 8:
      //  the low half of the word before, either half of the lane's own)
      may_run = may_run || (hc & 0x3FFFu) == 0 || ((hc >> 16) & 0x3FFFu) == 0 || (hp & 0x3FFFu) == 0;
    }
#ifdef JFGPU_P1_ALWAYS_RUNS
    may_run = true;
#endif
    if(__ballot(may_run) != 0ull) rounds(std::true_type{}); else rounds(std::false_type{});      // (wave-uniform: both sides meet the same barriers)
  }
  lds_barrier();                                                   // every append of every wave has landed
  if(owner) { ring_flush<ITEM>(s_ring, s_fill, t, true, B, my_region, cap, gcur, gshort, straggler); ring_finish<ITEM>(B, t, my_region, cap, gshort, tot); }
  lds_barrier();                                                   // (the final flush may have put items on the list)
  if(t == 0) strag_n[blockIdx.x] = s_nstrag < kStragPerBlock ? s_nstrag : kStragPerBlock;
  JF_PHASE(pc, 4);
  JF_PHASE_FLUSH(pc, 0);
  if(DIRECT::kCountsDirect && my_direct) atomicAdd((unsigned long long*)&T.counters[CTR_DIRECT], (unsigned long long)my_direct);
  uint64_t w = my_mers;
  for(int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
  if((threadIdx.x & 63) == 0 && w) atomicAdd((unsigned long long*)&T.counters[CTR_MERS], (unsigned long long)w);
}

// The straggler lists of a P1 launch, after it: an item takes the next free place of its bucket's region (nobody reserves
// granules any more, so places are handed out one by one), counted in tot like the others; what cannot be stored in a
// region -- the region is full, the item reads as a hole, a run of identical k-mers -- is inserted with global atomics
// (DIRECT: (bucket, item, occurrences), the table's global claim for this key width).
template <typename ITEM, typename DIRECT>
__global__ __launch_bounds__(256) void p1_stragglers_kernel(DIRECT D, unsigned long long* __restrict__ ctr_direct, const uint64_t* __restrict__ strag,
                                                            const uint32_t* __restrict__ strag_n, uint32_t n_lists, uint32_t cap,
                                                            unsigned int* __restrict__ gcur, unsigned long long* __restrict__ tot,
                                                            ITEM* __restrict__ out, uint32_t list_cap = kStragPerBlock) {
  using R = Ring<ITEM>;
  uint32_t my_direct = 0;
  for(uint32_t l = blockIdx.x; l < n_lists; l += gridDim.x) {
    const uint32_t n = strag_n[l];
    const uint64_t* rec = strag + (size_t)l * list_cap * R::kWords;
    for(uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
      uint32_t b, cnt; ITEM item;
      strag_load<ITEM>(rec + (size_t)i * R::kWords, b, item, cnt);
      if(cnt == 1 && item != (ITEM)~(ITEM)0) {
        const uint32_t at = atomicAdd(&gcur[b], 1u);
        if(at < cap) { out[(uint64_t)b * cap + at] = item; if(tot) atomicAdd(&tot[b], 1ull); continue; }
      }
      D(b, item, cnt);
      ++my_direct;
    }
  }
  if(ctr_direct && my_direct) atomicAdd(ctr_direct, (unsigned long long)my_direct);
}

// ---- P2 with the same rings (round 4): every P1 bucket -> its pairs of tiles, 32-bit items -----------------------------
// p2_granule_kernel counting-sorts chunks of 28 Ki items in LDS (histogram, scan, ranked scatter, read-back with a placement
// look-up per item: five barriers a chunk) and writes runs of ~28 items wherever a destination's run of the chunk starts:
// 29.5 ms for the 8.67 G items of the metric's job.  P2 has no hashing to do -- an item names its destination in a few of
// its own bits -- so with P1's rings (kernels above) it is a stream: two 16-byte loads give a lane its eight items of a
// round, each is appended to its destination's ring (one returning ds_add, one store), and after the round's barrier the
// owner lanes write out the units that are complete (aligned 64-byte runs).  grid = (blocks per bucket, buckets); the
// blocks of a bucket share its destinations' regions through the granule reservations, like P1's blocks.
// DIRECT: (destination, item, occurrences) for what cannot be stored in a region; strag: one list of kP2StragPerBlock
// entries per block.
// Measured on the metric's job (profiles/r04_p2_ring.log): 27.8 ms with one barrier a round -- a wave that is through its
// flush appended the next round into rings not yet released, 6.7 M stragglers a flush against the 0.3 M the rings' own
// overflow gives, and on BASELINE.md's secondary distribution the lists ran full (350 K global-atomic inserts dirtying
// tiles); 21.7 ms with a second barrier after the flush (0.4 M stragglers); 19.6 ms with the next round's loads issued
// before the current round is appended, in straight-line code (with the loads under the tail's conditions the compiler
// waited for them on the spot: s_waitcnt vmcnt(0) at the control-flow merge).
constexpr uint32_t kP2StragPerBlock = 2048;                                    // entries of a block's straggler list (the iid model: ~70 a block)

struct P2RingDirect {
  const DevTable* Tm; uint32_t b2; uint32_t dest_shift; int returning;      // destination >> dest_shift = the P1 bucket
  __device__ void operator()(uint32_t dest, uint64_t item, uint32_t cnt) const { item_direct_call(Tm, b2, dest >> dest_shift, item, cnt, returning); }
};

template <typename DIRECT>
__global__ __launch_bounds__(kPBlock) void p2_ring_kernel(DIRECT D, uint32_t b2e, uint32_t tag_bits, SegList S, uint32_t cap,
                                                          unsigned int* __restrict__ gcur, unsigned int* __restrict__ gshort,
                                                          uint32_t* __restrict__ out, uint32_t bucket0, unsigned long long* __restrict__ tot,
                                                          uint64_t* __restrict__ strag, uint32_t* __restrict__ strag_n, unsigned long long* __restrict__ ctr_direct) {
  using R = Ring<uint32_t>;
  constexpr int RP = 8;                                            // items per lane and round: two 16-byte loads
  JF_DYN_LDS(s_dyn);
  uint32_t* s_ring = reinterpret_cast<uint32_t*>(s_dyn);          // [nb][32], then 32 dump slots
  __shared__ uint32_t s_fill[kGranMaxB];
  __shared__ uint32_t s_nstrag;
  const uint32_t nb = 1u << b2e;
  const uint32_t bucket = bucket0 + blockIdx.y;
  const uint32_t dest0 = bucket * nb;                              // this bucket's first destination
  const uint32_t t = threadIdx.x, lane = t & 63;
  const bool owner = t < nb;
  const uint32_t hole = 0xFFFFFFFFu;
  ring_init<uint32_t>(s_ring, s_fill, nb, &s_nstrag);
  const uint32_t dump = nb * R::kSlots + (lane & (R::kSlots - 1));
  unsigned int* const gc = gcur + dest0;
  unsigned int* const gs = gshort + dest0;
  const uint32_t list = blockIdx.y * gridDim.x + blockIdx.x;
  uint64_t* const my_strag = strag + (size_t)list * kP2StragPerBlock;
  RingBooks B;
  if(owner) { B.nxt = atomicAdd(&gc[t], kGran); B.nxt_asked = true; }
  uint32_t* const my_region = out + ((uint64_t)dest0 + t) * cap;
  uint32_t my_direct = 0;
  auto straggler = [&](uint32_t d, uint32_t item, uint32_t cnt) {   // d: destination inside this bucket
    const uint32_t at = atomicAdd(&s_nstrag, 1u);
    if(at < kP2StragPerBlock) strag_store<uint32_t>(my_strag + at, dest0 + d, item, cnt);
    else { D(dest0 + d, (uint64_t)item, cnt); ++my_direct; }
  };
  // ring_flush speaks of "bucket t" with gcur / gshort indexed from the workgroup's first one: here, destinations
  auto flush = [&](bool all) { if(owner) ring_flush<uint32_t>(s_ring, s_fill, t, all, B, my_region, cap, gc, gs, straggler); };
  lds_barrier();
  // The rounds of this block, over the batches' regions for the bucket: batch s's region is cut among the bucket's blocks
  // at multiples of four items (16-byte loads; regions start at multiples of 64 items).  The items of round r + 1 are
  // requested before round r is appended, so their way from HBM hides behind the appends, the barriers and the flush.
  uint32_t seg = 0; uint64_t my_a = 0, my_b = 0;
  const uint32_t* src = nullptr;
  auto next_seg = [&]() -> bool {
    for(; seg < S.n; ++seg) {
      const uint64_t a0 = seg_lo(S, seg, bucket), b0 = seg_hi(S, seg, bucket);
      const uint64_t len4 = (b0 - a0 + 3) / 4;
      const uint64_t per4 = (len4 + gridDim.x - 1) / gridDim.x;
      const uint64_t q0 = (uint64_t)blockIdx.x * per4, q1 = q0 + per4;
      const uint64_t e1 = a0 + 4 * (q1 < len4 ? q1 : len4);
      my_a = a0 + 4 * (q0 < len4 ? q0 : len4); my_b = e1 < b0 ? e1 : b0;
      src = reinterpret_cast<const uint32_t*>(S.items[seg]);
      if(my_a < my_b) return true;
    }
    return false;
  };
  constexpr uint64_t RS = (uint64_t)kPBlock * RP;                  // items of a round
  // a whole round: two unconditional 16-byte loads (straight-line code, so that the wait for them can be counted and sits
  // where the values are first used)
  auto load_full = [&](uint64_t r0, uint32_t (&it)[RP]) {
#pragma unroll
    for(int h = 0; h < RP / 4; ++h) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + r0 + (uint64_t)h * kPBlock * 4 + 4 * (uint64_t)t);
      it[4 * h] = v.x; it[4 * h + 1] = v.y; it[4 * h + 2] = v.z; it[4 * h + 3] = v.w;
    }
  };
  // the last, partial round of a batch's region
  auto load_partial = [&](uint64_t r0, uint32_t (&it)[RP]) {
#pragma unroll
    for(int h = 0; h < RP / 4; ++h) {
      const uint64_t i = r0 + (uint64_t)h * kPBlock * 4 + 4 * (uint64_t)t;
      uint4 v = make_uint4(hole, hole, hole, hole);
      if(i + 4 <= my_b) v = *reinterpret_cast<const uint4*>(src + i);
      else if(i < my_b) { v.x = src[i]; if(i + 1 < my_b) v.y = src[i + 1]; if(i + 2 < my_b) v.z = src[i + 2]; }
      it[4 * h] = v.x; it[4 * h + 1] = v.y; it[4 * h + 2] = v.z; it[4 * h + 3] = v.w;
    }
  };
  auto process = [&](const uint32_t (&it)[RP]) {
    uint32_t ea[RP], eo[RP];
#pragma unroll
    for(int e = 0; e < RP; ++e) {
      ea[e] = dump; eo[e] = 0;
      if(it[e] != hole) { const uint32_t d = (it[e] >> tag_bits) & (nb - 1); ea[e] = d * R::kSlots; eo[e] = atomicAdd(&s_fill[d], 1u); }
    }
    uint32_t ghosts = 0;
#pragma unroll
    for(int e = 0; e < RP; ++e) {
      const uint32_t full = eo[e] & R::kFull;
      ghosts |= full;
      const uint32_t at = ea[e] + ((eo[e] + (eo[e] >> 16)) & (R::kSlots - 1));
      s_ring[full ? dump : at] = it[e];
    }
    if(ghosts) {
#pragma unroll 1
      for(int e = 0; e < RP; ++e) if(eo[e] & R::kFull) straggler(ea[e] / R::kSlots, it[e], 1u);
    }
    lds_barrier();                                                 // the round's items have all landed
    flush(false);
    // Without this barrier a wave that is through its flush appends the next round into rings whose owners have not
    // released them yet: ~25 x the stragglers of the rings' own overflow (measured: 6.7 M a flush of 8.67 G items against
    // the 0.3 M of the iid model; P2 27.4 ms instead of 21.7).
    lds_barrier();
  };
  for(bool have = next_seg(); have; ++seg, have = next_seg()) {    // (have: the same for every lane of the block)
    uint64_t r0 = my_a;
    const uint64_t n_full = (my_b - my_a) / RS;
    if(n_full) {
      uint32_t nx[RP];
      load_full(r0, nx);
#pragma unroll 1
      for(uint64_t k = 0; k < n_full; ++k) {
        uint32_t it[RP];
#pragma unroll
        for(int e = 0; e < RP; ++e) it[e] = nx[e];
        r0 += RS;
        load_full(k + 1 < n_full ? r0 : my_a, nx);                 // (after the last round: a load nobody looks at)
        process(it);
      }
    }
    if(r0 < my_b) { uint32_t it[RP]; load_partial(r0, it); process(it); }
  }
  lds_barrier();
  flush(true);
  if(owner) ring_finish<uint32_t>(B, t, my_region, cap, gs, tot ? tot + dest0 : nullptr);
  lds_barrier();
  if(t == 0) strag_n[list] = s_nstrag < kP2StragPerBlock ? s_nstrag : kP2StragPerBlock;
  if(my_direct) atomicAdd(ctr_direct, (unsigned long long)my_direct);
}

// ---- P2 with loader waves and storer waves (round 4) ---------------------------------------------------------------------
// What bounds p2_ring_kernel's round is not its work but a property of the vector-memory counter: on this chip loads and
// stores of a wave are counted by ONE in-order counter (vmcnt), so a wave that waits for the items it requested a round
// ago also waits for every store it issued in between -- the units of its last flush -- to be acknowledged by memory
// (the compiler cannot count stores issued under divergent control flow and waits for all but the newest loads), and a
// lane taking up a granule reservation (a returning atomic) waits the same way.  A round therefore cost ~4 us whatever it
// held (measured with the Bloom P1b ring kernel, whose rounds of 2, 3 and 4 cells a lane took 4.4, 5.3 and 5.9 us).
// Here no wave does both: waves 0-7 (storers) own the rings -- lane t those of destinations t and t + 512 -- and only
// read LDS and store units; waves 8-15 (loaders) only load items and append them.  One workgroup per bucket, so a
// destination's region has a single writer: no reservations, no atomics, a cursor in a register.  One barrier a round:
// the storers write out round r while the loaders append round r + 1, which the rings can take because a round is 4 Ki
// items (mean 4 a ring: 15 left over + 4 + 4 stays far below 32).  The hole markers tell a storer which units are complete.
// NV: 16-byte loads per loader lane and round -- 2 for 1024 destinations, 1 for 512 (a ring must not see more than ~4
// appends a round with the flush a round behind).  The kernel is written over the item type and was measured with 8-byte
// items (k = 31: rings of 16, rounds of 1 Ki items): 28.5 ms per 5 Gbp against 23.4 for the sort-based kernel -- a round
// costs ~1.9 us however little it holds, and a ring of 128 bytes takes too few wide items for long rounds.  Instantiated
// for 4-byte items only.
template <typename ITEM, int NV, typename DIRECT, int PD = 1>
__global__ __launch_bounds__(kPBlock) void p2_ring_roles_kernel(DIRECT D, uint32_t b2e, uint32_t tag_bits, SegList S, uint32_t cap,
                                                                unsigned int* __restrict__ gcur, ITEM* __restrict__ out, uint32_t bucket0,
                                                                uint64_t* __restrict__ strag, uint32_t* __restrict__ strag_n, unsigned long long* __restrict__ ctr_direct) {
  using R = Ring<ITEM>;
  constexpr int VPI = (int)R::kChunk;                              // items per load
  constexpr int RP = NV * VPI;                                     // items per loader lane and round
  constexpr uint32_t kHalf = kPBlock / 2;
  constexpr uint64_t RS = (uint64_t)kHalf * RP;                    // items of a round
  JF_DYN_LDS(s_dyn);
  ITEM* s_ring = reinterpret_cast<ITEM*>(s_dyn);                  // [nb][R::kSlots], then 128 bytes of dump slots
  __shared__ uint32_t s_fill[kGranMaxB];
  __shared__ uint32_t s_nstrag;
  const uint32_t nb = 1u << b2e;
  const uint32_t bucket = bucket0 + blockIdx.x;
  const uint32_t dest0 = bucket * nb;
  const uint32_t t = threadIdx.x, lane = t & 63;
  const bool storer = t < kHalf;                                   // (whole waves)
  const ITEM hole = (ITEM)~(ITEM)0;
  ring_init<ITEM>(s_ring, s_fill, nb, &s_nstrag);
  const uint32_t dump = nb * R::kSlots + (lane & (R::kSlots - 1));
  uint64_t* const my_strag = strag + (size_t)blockIdx.x * kP2StragPerBlock * R::kWords;
  uint32_t my_direct = 0;
  auto straggler = [&](uint32_t d, ITEM item, uint32_t cnt) {       // d: destination inside this bucket
    const uint32_t at = atomicAdd(&s_nstrag, 1u);
    if(at < kP2StragPerBlock) strag_store<ITEM>(my_strag + (size_t)at * R::kWords, dest0 + d, item, cnt);
    else { D(dest0 + d, item, cnt); ++my_direct; }
  };
  lds_barrier();
  // the rounds of the bucket, the same sequence for both roles: the batches' regions one after the other, whole rounds
  // first, then a partial one
  auto seg_bounds = [&](uint32_t seg, uint64_t& a, uint64_t& b) { a = seg_lo(S, seg, bucket); b = seg_hi(S, seg, bucket); };
  if(storer) {
    const uint32_t d0 = t, d1 = t + kHalf;
    const bool has0 = d0 < nb, has1 = d1 < nb;
    RingBooks B0, B1;
    B0.room = B1.room = cap & ~(R::kUnit - 1);
    ITEM* const r0p = out + ((uint64_t)dest0 + d0) * cap;
    ITEM* const r1p = out + ((uint64_t)dest0 + d1) * cap;
    auto flush = [&](bool all) {
      if(has0) ring_flush<ITEM, true>(s_ring, s_fill, d0, all, B0, r0p, cap, gcur, gcur, straggler);
      if(has1) ring_flush<ITEM, true>(s_ring, s_fill, d1, all, B1, r1p, cap, gcur, gcur, straggler);
    };
    for(uint32_t seg = 0; seg < S.n; ++seg) {
      uint64_t a, b; seg_bounds(seg, a, b);
      const uint64_t rounds = (b - a + RS - 1) / RS;
#pragma unroll 1
      for(uint64_t r = 0; r < rounds; ++r) { lds_barrier(); flush(false); }
    }
    lds_barrier();                                                 // the loaders are through: every append has landed
    flush(true);
    // the region's fill for what comes after (p1_stragglers_kernel appends one by one, granule_finish_kernel reads it)
    if(has0) gcur[dest0 + d0] = B0.gpos;
    if(has1) gcur[dest0 + d1] = B1.gpos;
  } else {
    const uint32_t l = t - kHalf;                                  // 0 .. 511
    for(uint32_t seg = 0; seg < S.n; ++seg) {
      uint64_t a, b; seg_bounds(seg, a, b);
      const ITEM* src = reinterpret_cast<const ITEM*>(S.items[seg]);
      auto load_full = [&](uint64_t r0, ITEM (&it)[RP]) {
#pragma unroll
        for(int h = 0; h < NV; ++h) {
          const uint4 v = *reinterpret_cast<const uint4*>(src + r0 + (uint64_t)h * kHalf * VPI + (uint64_t)VPI * l);
          int q = 0;
          chunk_items<ITEM>(v, [&](ITEM x) { it[VPI * h + q] = x; ++q; });
        }
      };
      auto load_partial = [&](uint64_t r0, ITEM (&it)[RP]) {
#pragma unroll
        for(int h = 0; h < NV; ++h) {
          const uint64_t i = r0 + (uint64_t)h * kHalf * VPI + (uint64_t)VPI * l;
#pragma unroll
          for(int q = 0; q < VPI; ++q) it[VPI * h + q] = i + q < b ? src[i + q] : hole;
        }
      };
      // A batch of the exact two-pass P1 (a small batch: a file's tail chunk, add_keys) has no holes and starts anywhere:
      // item by item, and an item that EQUALS the hole marker -- a real k-mer when items are 32 bits wide -- goes on the
      // straggler list, from where it is inserted directly (round-4 advisor finding: it used to be dropped as a hole).
      auto load_exact = [&](uint64_t r0, ITEM (&it)[RP]) {
#pragma unroll 1
        for(int h = 0; h < NV; ++h) {
          const uint64_t i = r0 + (uint64_t)h * kHalf * VPI + (uint64_t)VPI * l;
#pragma unroll
          for(int q = 0; q < VPI; ++q) {
            ITEM x = hole;
            if(i + q < b) { x = src[i + q]; if(x == hole) straggler((uint32_t)(x >> tag_bits) & (nb - 1), x, 1u); }
            it[VPI * h + q] = x;
          }
        }
      };
      auto append = [&](const ITEM (&it)[RP]) {
        uint32_t ea[RP], eo[RP];
#pragma unroll
        for(int e = 0; e < RP; ++e) {
          ea[e] = dump; eo[e] = 0;
          if(it[e] != hole) { const uint32_t d = (uint32_t)(it[e] >> tag_bits) & (nb - 1); ea[e] = d * R::kSlots; eo[e] = atomicAdd(&s_fill[d], 1u); }
        }
        uint32_t ghosts = 0;
#pragma unroll
        for(int e = 0; e < RP; ++e) {
          const uint32_t full = eo[e] & R::kFull;
          ghosts |= full;
          const uint32_t at = ea[e] + ((eo[e] + (eo[e] >> 16)) & (R::kSlots - 1));
          s_ring[full ? dump : at] = it[e];
        }
        if(ghosts) {
#pragma unroll 1
          for(int e = 0; e < RP; ++e) if(eo[e] & R::kFull) straggler(ea[e] / R::kSlots, it[e], 1u);
        }
        lds_barrier();                                             // the storers take this round from here
      };
      uint64_t r0 = a;
      if(S.sh[seg] == 0) {                                         // (the same number of rounds as the storers count)
#pragma unroll 1
        for(; r0 < b; r0 += RS) { ITEM it[RP]; load_exact(r0, it); append(it); }
        continue;
      }
      const uint64_t n_full = (b - a) / RS;
      if(n_full) {
        // PD rounds of items are on their way while one is appended: a round lasts ~2 us, about what a load takes under
        // load, so with one round ahead (round 4) the loaders met their own latency every round (round 5: JFGPU_P2_DEPTH)
        ITEM nx[PD][RP];
#pragma unroll
        for(int d = 0; d < PD; ++d) load_full((uint64_t)d < n_full ? a + (uint64_t)d * RS : a, nx[d]);
        uint64_t k = 0;
#pragma unroll 1
        for(; k + PD <= n_full; k += PD) {                          // straight-line body: the waits for nx[d] can be counted
#pragma unroll
          for(int d = 0; d < PD; ++d) {
            ITEM it[RP];
#pragma unroll
            for(int e = 0; e < RP; ++e) it[e] = nx[d][e];
            const uint64_t nxt = k + d + PD;
            load_full(nxt < n_full ? a + nxt * RS : a, nx[d]);      // (past the last round: a load nobody looks at)
            append(it);
          }
        }
#pragma unroll
        for(int d = 0; d < PD - 1; ++d)                             // the last n_full % PD rounds are in nx already
          if(k + d < n_full) append(nx[d]);
        r0 = a + n_full * RS;
      }
      if(r0 < b) { ITEM it[RP]; load_partial(r0, it); append(it); }
    }
    lds_barrier();
  }
  lds_barrier();                                                   // (the final flush may have put items on the list)
  if(t == 0) strag_n[blockIdx.x] = s_nstrag < kP2StragPerBlock ? s_nstrag : kP2StragPerBlock;
  if(my_direct) atomicAdd(ctr_direct, (unsigned long long)my_direct);
}

}  // namespace jfgpu
