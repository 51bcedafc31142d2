// jellyfish_amd/csrc/jfgpu.hip -- C ABI (include/jfgpu.h) over the gfx950 kernels.
//
// Host-side role in the reference: the part of sub_commands/count_main.cc that
// owns the table (mer_hash ctor, :275), drives the counting threads (:331-333)
// and hands the table to the dumper (:349-355).  Here the "threads" are HIP
// kernels enqueued on one stream per table; there is no CPU fallback.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "../../include/jfgpu.h"
#include "gf2_matrix.hpp"
#include "tuning.hpp"
#include "kernels.hip.hpp"
#include "kernels_part.hip.hpp"
#include "kernels_tile.hip.hpp"
#include "kernels_p1ring.hip.hpp"
#include "kernels_bloom.hip.hpp"
#include "kernels_bloom_part.hip.hpp"
#include "kernels_wide.hip.hpp"
#include "kernels_wide_part.hip.hpp"
#include "kernels_nword.hip.hpp"
#include "kernels_parse.hip.hpp"

using namespace jfgpu;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if(e_ != hipSuccess)                                                                      \
      return fail(e_ == hipErrorOutOfMemory ? JFGPU_E_ALLOC : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? JFGPU_E_NO_DEVICE : JFGPU_E_HIP), \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                          \
  } while(0)

constexpr uint64_t kDefaultSeed = 0x6A656C6C79666973ull;  // "jellyfis"
constexpr size_t kStageBytes = 64u << 20;                  // host->device staging chunk
constexpr int kNumProf = 8;   // 0 count 1 add_keys 2 shard-partition 3 lookup 4 P1 5 P2 6 tile-insert 7 items-direct
enum Mode { MODE_AUTO = 0, MODE_DIRECT = 1, MODE_PARTITIONED = 2 };

// Entries of the count-overflow side table of a small table (1 MiB): counts beyond the in-slot field are rare in k-mer
// counting, but hash_counter::add(key, huge value) -- what the reference's unit tests do -- needs one per key.
constexpr uint64_t kMinOvf = 1ull << 16;

struct PendingBatch {
  void* items; uint64_t* off; uint64_t cap_items;
  uint32_t gran_cap = 0;                      // > 0: single-pass ("granule") batch, items per bucket region; off is in pair format
  unsigned long long* tot = nullptr;          // granule batch: exact items per bucket
  uint64_t input_bytes = 0;                   // sequence bytes this batch was made from (0: encoded keys)
  uint64_t bound = 0;                         // what ensure_capacity charged for this batch (an upper bound on its k-mers; 0: not charged)
};

struct ProfSpan { hipEvent_t a, b; int which; uint64_t units; };

}  // namespace

struct jfgpu_table {
  jfgpu_params params{};
  TableGeom g{};
  Gf2Matrix matrix;
  int device = 0;
  int n_cu = 256;
  hipStream_t stream = nullptr;
  DevTable dt{};
  DevTable* d_dt = nullptr;      // a copy of dt in device memory for kernels that call out of line with it (refreshed before every such launch)
  uint64_t* d_fwd = nullptr;
  uint64_t* d_inv = nullptr;
  int failed = 0; std::string failed_msg;   // a flush failed: every call reports it until jfgpu_clear (host_partition.inl: part_flush)
  uint64_t ovf_cap = 0;
  // what the side table may have to hold: an entry needs 2^cnt_bits occurrences of its key, or one add of a large value
  uint64_t occ_bound = 0, bigval_bound = 0;      // upper bounds since the last clear (ensure_ovf)
  uint64_t flushes_plain = 0, flushes_heavy = 0; // tile-kernel instantiation chosen per flush launch (jfgpu_get_counters)
  // which partition kernels ran since the last clear (jfgpu_get_counters 10..15): P2 launches by kind -- loader / storer
  // rings, shared rings, the sort-based single pass, the exact count + scatter -- and P1 launches: ring kernel, any other
  // count --bc: the cache of admitted k-mers (kernels_bloom.hip.hpp) -- 0 undecided, 1 on, -1 off; decided from the first
  // filtered batches (host_partition.inl: bloom_cache_decide), dropped when the counter is detached
  uint64_t* d_bcache = nullptr; int bcache_state = 0; uint64_t mers_seen = 0;
  uint64_t n_p2_roles = 0, n_p2_ring = 0, n_p2_sort = 0, n_p2_exact = 0, n_p1_ring = 0, n_p1_other = 0;
  uint64_t ovf_failed_need = 0;                  // the side-table size whose allocation failed (not retried per batch)
  bool returning = false;
  uint32_t out_counter_len = 4;
  // size doubling (hash_counter::do_size_doubling): occupancy bookkeeping, see ensure_capacity()
  bool grow_on = true;
  uint64_t occ_known = 0, fed_since = 0;
  uint64_t cur_bound = 0;        // the charge of the piece being ingested right now (launch_count -> part_ingest)
  uint64_t direct_seen = 0;      // CTR_DIRECT when the charges were last corrected
  uint64_t grow_seed = 0;
  // two-word keys (33 <= k <= 64): 128-bit slots, kernels_wide.hip.hpp
  bool wide = false;
  // keys of three or four words (65 <= k <= 128): 256-bit slots, kernels_nword.hip.hpp (direct path only)
  bool nword = false;
  NTable nt{};
  WideTable wt{};
  uint32_t key_words = 1;        // 64-bit words of a key in the API (mer_dna::data(): ceil(2k / 64))
  uint32_t slot_words = 1;       // 64-bit words of a table slot (1, 2 or 4)
  // staging for host buffers
  uint8_t* d_stage[2] = {nullptr, nullptr};
  hipEvent_t stage_done[2] = {nullptr, nullptr};
  int stage_next = 0;
  // profiling
  bool prof_on = false;
  std::vector<ProfSpan> prof_pending;
  std::vector<hipEvent_t> ev_pool;
  std::vector<std::pair<int, float>> prof_log;   // every span since the last reset, in launch order (jfgpu_profile_spans)
  double prof_ms[kNumProf] = {};
  uint64_t prof_launches[kNumProf] = {};
  uint64_t prof_units[kNumProf] = {};
  // partitioned insert path (kernels_part.hip.hpp)
  int mode = MODE_AUTO;
  bool part_ok = false;          // geometry admits the partitioned path
  PartGeom pg{};
  bool item32 = false;
  bool item128 = false;          // two-word keys: 128-bit items (kernels_wide_part.hip.hpp)
  bool pristine = true;          // table known all-zero: tile_insert may skip the tile read
  std::vector<PendingBatch> pending;
  uint64_t pending_bytes = 0;
  bool ref_matrix = false;       // default matrix family: the reference's (glibc random() stream kept in `glibc`)
  bool xs_matrix = false;        // the xor-shift family (gf2_xorshift_matrix): doublings stay in it
  GlibcRandom glibc;
  int (*spill_fn)(void*) = nullptr; void* spill_user = nullptr;     // jfgpu_set_spill
  int operation = 0;             // what count_ascii does with a k-mer: 0 add, 1 set (prime), 2 update_add (jfgpu_set_operation)
  Tuning tun;                    // the JFGPU_* switches as they were when the table was created (tuning.hpp)
  hipStream_t stream2 = nullptr; hipEvent_t flush_ev[2] = {nullptr, nullptr}; hipEvent_t flush_done = nullptr;
  double items_per_byte = 0;     // k-mers per sequence byte seen by the last flush (0: unknown yet)
  uint64_t reserved_input = 0;   // sequence bytes the caller announced (jfgpu_reserve): lets forced flushes be spaced evenly
  uint32_t* d_M1 = nullptr; int g1 = 0;
  uint64_t* d_strag2 = nullptr; uint32_t* d_strag2_n = nullptr; uint32_t strag2_lists = 0;      // ... and of the ring P2, one per workgroup
  uint64_t* d_strag = nullptr; uint32_t* d_strag_n = nullptr;      // straggler lists of the ring P1 (kernels_p1ring.hip.hpp), one per workgroup
  uint32_t* d_M2 = nullptr; int g2 = 0;
  // workspace arena for pending batches and flush temporaries: bump-allocated, reset at flush,
  // grown geometrically (allocation of tens of GB costs ~100 ms, so never inside the hot path twice)
  uint8_t* ws = nullptr; size_t ws_cap = 0, ws_used = 0;
  // dump state
  bool dump_open = false;
  uint64_t dump_lower = 0, dump_upper = 0;
  int dump_have_ovf = 0;
  std::vector<uint64_t> dump_prefix;  // per tile exclusive prefix, size n_tiles + 1
  uint64_t dump_tile_cursor = 0;
  uint8_t* d_dump = nullptr; uint64_t dump_cap_records = 0;
  uint64_t* d_tile_off = nullptr; uint64_t tile_off_cap = 0;
};

namespace {

int use(const jfgpu_table* t) {
  if(!t) return fail(JFGPU_E_INVALID, "null table");
  HIP_TRY(hipSetDevice(t->device));
  if(t->failed) return fail(t->failed, t->failed_msg + " (earlier; the table must be cleared)");
  return JFGPU_OK;
}

size_t slot_bytes_of(const jfgpu_table* t) { return t->g.slot32 ? 4 : 8 * (size_t)t->slot_words; }
uint64_t n_tiles_of(const jfgpu_table* t) { return 1ull << (t->g.lsize_l - t->g.tile_bits); }

int grid_for(const jfgpu_table* t, uint64_t work_items) {
  uint64_t g = std::min<uint64_t>(work_items, (uint64_t)t->n_cu * 8);
  return (int)std::max<uint64_t>(g, 1);
}

hipEvent_t get_event(jfgpu_table* t) {
  if(!t->ev_pool.empty()) { hipEvent_t e = t->ev_pool.back(); t->ev_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  hipEventCreate(&e);
  return e;
}

struct ProfScope {  // records a HIP-event pair around a launch on the table's stream
  jfgpu_table* t; int which; uint64_t units; hipEvent_t a = nullptr, b = nullptr;
  ProfScope(jfgpu_table* t_, int w, uint64_t u) : t(t_), which(w), units(u) {
    if(t->prof_on) { a = get_event(t); b = get_event(t); hipEventRecord(a, t->stream); }
  }
  ~ProfScope() {
    if(t->prof_on) { hipEventRecord(b, t->stream); t->prof_pending.push_back({a, b, which, units}); }
  }
};

void prof_collect(jfgpu_table* t) {
  for(auto& s : t->prof_pending) {
    float ms = 0;
    hipEventSynchronize(s.b);
    if(hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      t->prof_ms[s.which] += ms; t->prof_launches[s.which] += 1; t->prof_units[s.which] += s.units;
      if(t->prof_log.size() < (size_t)1 << 16) t->prof_log.emplace_back(s.which, ms);
    }
    t->ev_pool.push_back(s.a); t->ev_pool.push_back(s.b);
  }
  t->prof_pending.clear();
}

int read_counters(jfgpu_table* t, uint64_t* out /* CTR_COUNT */) {
  HIP_TRY(hipMemcpyAsync(out, t->dt.counters, sizeof(uint64_t) * CTR_COUNT, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  return JFGPU_OK;
}

int check_deferred(jfgpu_table* t, uint64_t* ctr_out = nullptr) {
  uint64_t c[CTR_COUNT];
  int rc = read_counters(t, c);
  if(rc) return rc;
  if(ctr_out) memcpy(ctr_out, c, sizeof(c));
  if(c[CTR_MISROUTED])
    return fail(JFGPU_E_INVALID, std::to_string(c[CTR_MISROUTED]) + " k-mers were added to a shard that does not own them");
  if(c[CTR_FULL]) return fail(JFGPU_E_FULL, "Hash full");
  if(c[CTR_OVF_FULL]) return fail(JFGPU_E_FULL, "Hash full (count overflow table exhausted)");
  return JFGPU_OK;
}

bool use_partitioned(const jfgpu_table* t, size_t nbytes);
int part_ingest(jfgpu_table* t, const uint8_t* base, int64_t lo, int64_t hi, bool from_keys, uint64_t max_items);
int part_flush(jfgpu_table* t);
int table_grow(jfgpu_table* t);
int measure_occupancy(jfgpu_table* t);

// Splits a device buffer pointer into a 16-byte aligned base and [lo, hi).
void align_buffer(const char* d, size_t n, const uint8_t*& base, int64_t& lo, int64_t& hi) {
  const uintptr_t p = (uintptr_t)d;
  const uintptr_t a = p & ~(uintptr_t)15;
  base = (const uint8_t*)a; lo = (int64_t)(p - a); hi = lo + (int64_t)n;
}

// ---- the overflow side table grows with what it may have to hold -------------------------------------------------
// A count field of cnt_bits bits wraps into the side table (ovf_add: one entry per key, units of 2^cnt_bits).  An entry
// costs its key 2^cnt_bits occurrences (or one add of a value that large), so (occurrences fed >> cnt_bits) + large adds
// bounds the entries whatever the input; the table is kept at twice that, rebuilt larger before the work that could
// fill it is enqueued (the 32-bit slots of round 2 brought count fields of 8-10 bits: a fixed-size side table then
// fails on repeat-rich input, where the reference just keeps counting: large_hash_array.hpp:887-937).
__global__ void ovf_rehash_kernel(const uint64_t* __restrict__ okey, const uint64_t* __restrict__ ocnt, uint64_t ocap,
                                  uint64_t* __restrict__ nkey, uint64_t* __restrict__ ncnt, uint64_t nmask, unsigned long long* __restrict__ lost) {
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ocap; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t want = okey[i];
    if(!want) continue;
    const uint64_t h = ((want - 1) * 0x9E3779B97F4A7C15ull) >> 20;
    bool placed = false;
    for(uint64_t p = 0; p <= nmask && !placed; ++p) {
      const uint64_t s = (h + p) & nmask;
      const unsigned long long old = atomicCAS((unsigned long long*)&nkey[s], 0ull, (unsigned long long)want);
      if(old == 0ull || old == want) { atomicAdd((unsigned long long*)&ncnt[s], (unsigned long long)ocnt[i]); placed = true; }
    }
    if(!placed) atomicAdd(lost, 1ull);
  }
}

void refresh_views(jfgpu_table* t);     // the two- and N-word views of the table share its side table and counters

int ensure_ovf(jfgpu_table* t, uint64_t more_occurrences, uint64_t more_big_adds) {
  t->occ_bound += more_occurrences; t->bigval_bound += more_big_adds;
  if(!t->returning) return JFGPU_OK;                              // count fields of 40 bits and more
  // entries are keyed by slot: there are never more of them than slots, whatever was fed (round-3 advisor finding: the
  // bound alone asked for 32 GB of side table at 100 Gbp into 10-bit count fields)
  const uint64_t slots = 1ull << t->g.lsize_l;
  const uint64_t need = std::min<uint64_t>(2 * ((t->occ_bound >> t->g.cnt_bits) + t->bigval_bound) + 4096, std::max<uint64_t>(2 * slots, 4096));
  if(need <= t->ovf_cap || need <= t->ovf_failed_need) return JFGPU_OK;      // (a size that could not be had is not asked for again until the table is cleared or grows)
  uint64_t cap2 = t->ovf_cap;
  while(cap2 < 2 * need && cap2 < 4 * slots) cap2 <<= 1;          // (room for the next few batches too)
  while(cap2 < need) cap2 <<= 1;
  HIP_TRY(hipStreamSynchronize(t->stream));
  if(t->stream2) HIP_TRY(hipStreamSynchronize(t->stream2));
  uint64_t *nk = nullptr, *nc = nullptr;
  if(hipMalloc((void**)&nk, cap2 * 8) != hipSuccess || hipMalloc((void**)&nc, cap2 * 8) != hipSuccess) {
    if(nk) hipFree(nk);
    (void)hipGetLastError();
    t->ovf_failed_need = need;
    return JFGPU_OK;                                              // no memory: carry on, CTR_OVF_FULL reports it if it really overflows
  }
  HIP_TRY(hipMemsetAsync(nk, 0, cap2 * 8, t->stream));
  HIP_TRY(hipMemsetAsync(nc, 0, cap2 * 8, t->stream));
  hipLaunchKernelGGL(ovf_rehash_kernel, dim3(grid_for(t, t->ovf_cap / kBlock + 1)), dim3(kBlock), 0, t->stream, t->dt.ovf_key, t->dt.ovf_cnt, t->ovf_cap, nk, nc, cap2 - 1,
                     (unsigned long long*)&t->dt.counters[CTR_OVF_FULL]);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(t->stream));
  hipFree(t->dt.ovf_key); hipFree(t->dt.ovf_cnt);
  t->dt.ovf_key = nk; t->dt.ovf_cnt = nc; t->dt.ovf_mask = cap2 - 1; t->ovf_cap = cap2;
  refresh_views(t);
  return JFGPU_OK;
}

int launch_count_chunk(jfgpu_table* t, const char* d_bases, size_t n) {
  if(n < t->g.k) return JFGPU_OK;
  { int rc = ensure_ovf(t, n, 0); if(rc) return rc; }
  const uint8_t* base; int64_t lo, hi;
  align_buffer(d_bases, n, base, lo, hi);
  if(t->nword) {
    t->pristine = false;
    const int64_t nt = (hi + kTilePos - 1) / kTilePos;
    ProfScope ps(t, 0, n);
    if(t->returning) hipLaunchKernelGGL(count_ascii_nword_kernel<true>, dim3(grid_for(t, (uint64_t)nt)), dim3(kBlock), 0, t->stream, t->nt, base, lo, hi, t->operation);
    else             hipLaunchKernelGGL(count_ascii_nword_kernel<false>, dim3(grid_for(t, (uint64_t)nt)), dim3(kBlock), 0, t->stream, t->nt, base, lo, hi, t->operation);
    HIP_TRY(hipGetLastError());
    return JFGPU_OK;
  }
  if(t->wide) {
    if(t->operation == 0 && use_partitioned(t, n)) {
      const int rc = part_ingest(t, base, lo, hi, false, n);
      if(rc >= 0) return rc;        // < 0: batch too small for the single-pass partition / no memory -> direct kernel
    }
    t->pristine = false;
    const int64_t nt = (hi + kTilePos - 1) / kTilePos;
    ProfScope ps(t, 0, n);
    if(t->returning) hipLaunchKernelGGL(count_ascii_wide_kernel<true>, dim3(grid_for(t, (uint64_t)nt)), dim3(kBlock), 0, t->stream, t->wt, base, lo, hi, t->operation);
    else             hipLaunchKernelGGL(count_ascii_wide_kernel<false>, dim3(grid_for(t, (uint64_t)nt)), dim3(kBlock), 0, t->stream, t->wt, base, lo, hi, t->operation);
    HIP_TRY(hipGetLastError());
    return JFGPU_OK;
  }
  if(t->operation == 0 && use_partitioned(t, n)) {        // PRIME / UPDATE passes of --if run on the direct kernel
    const int rc = part_ingest(t, base, lo, hi, false, n);
    if(rc >= 0) return rc;          // < 0: no memory for the pending batch -> direct kernel below
  }
  t->pristine = false;
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  const int grid = grid_for(t, (uint64_t)n_tiles);
  ProfScope ps(t, 0, n);
  const bool bl = t->dt.bloom.data != nullptr;
#define CA(RT, BL) hipLaunchKernelGGL((count_ascii_kernel<RT, BL>), dim3(grid), dim3(kBlock), 0, t->stream, t->dt, base, lo, hi, t->operation)
  if(t->returning) { if(bl) CA(true, true); else CA(true, false); } else { if(bl) CA(false, true); else CA(false, false); }
#undef CA
  HIP_TRY(hipGetLastError());
  return JFGPU_OK;
}


// How many more k-mers may be enqueued before the table could exceed 80 % load, assuming every one
// of them is new (an upper bound: duplicates are only discovered by inserting).
uint64_t capacity_limit(const jfgpu_table* t) { return ((1ull << t->g.lsize_l) / 10) * 8; }
bool capacity_managed(const jfgpu_table* t) { return (t->grow_on || t->spill_fn) && t->g.shard_bits == 0 && t->g.lsize_g < t->g.key_bits && (!(t->wide || t->nword) || t->g.lsize_g < 48); }

// The size passed at creation is a hint (doc/Readme.md:67-72; hash_counter::handle_full_ary,
// hash_counter.hpp:178-198): before enqueuing `incoming` potential new keys make sure they cannot
// overflow the table -- measure the true occupancy when the running upper bound runs out, double the
// table (cooperative rehash on the device) when it is really more than half full.  Returns the
// number of k-mers that may be enqueued now (<= incoming, > 0).
extern "C" int jfgpu_clear(jfgpu_table* t);
int refine_pending_charges(jfgpu_table* t);
int ensure_capacity(jfgpu_table* t, uint64_t incoming, uint64_t* allowed) {
  *allowed = incoming;
  if(!capacity_managed(t)) return JFGPU_OK;
  uint64_t limit = capacity_limit(t);
  if(t->occ_known + t->fed_since + incoming <= limit) { t->fed_since += incoming; return JFGPU_OK; }
  // the charges are one k-mer per input byte; what is pending knows its exact item count (no flush needed to read it)
  int rc = refine_pending_charges(t); if(rc) return rc;
  if(t->occ_known + t->fed_since + incoming <= limit) { t->fed_since += incoming; return JFGPU_OK; }
  rc = measure_occupancy(t); if(rc) return rc;
  if(!t->grow_on) {
    // --disk (count_main.cc:276-277, hash_counter.hpp:178-198 with a dumper): when what is left would not take a
    // useful piece, the caller writes the table out as one sorted run, the table is emptied and counting goes on
    // (the runs are merged at the end).  Pieces never exceed the room left, so nothing is ever dropped.
    uint64_t room = limit > t->occ_known ? limit - t->occ_known : 0;
    if(room < std::min<uint64_t>(incoming, std::max<uint64_t>(limit / 8, 1))) {
      rc = check_deferred(t); if(rc) return rc;
      if(t->spill_fn(t->spill_user) != 0) return fail(JFGPU_E_INVALID, "the spill callback failed");
      rc = jfgpu_clear(t); if(rc) return rc;
      room = capacity_limit(t);
    }
    *allowed = std::max<uint64_t>(1, std::min<uint64_t>(incoming, room));
    t->fed_since += *allowed;
    return JFGPU_OK;
  }
  const uint64_t min_piece = std::min<uint64_t>(incoming, 65536);      // never enqueue less than this at a time
  while(true) {
    limit = capacity_limit(t);
    const uint64_t headroom = limit > t->occ_known ? limit - t->occ_known : 0;
    const bool really_full = t->occ_known > (1ull << t->g.lsize_l) / 2;
    if(!really_full && headroom >= min_piece) break;
    if(t->g.lsize_g >= t->g.key_bits) break;                             // 4^k positions: cannot fill up
    rc = table_grow(t);
    if(rc < 0) break;            // no memory for a bigger table: carry on, "Hash full" if it really overflows
    if(rc) return rc;
  }
  limit = capacity_limit(t);
  const uint64_t headroom = limit > t->occ_known ? limit - t->occ_known : 0;
  *allowed = std::min<uint64_t>(incoming, std::max<uint64_t>(headroom, min_piece));
  t->fed_since += *allowed;
  return JFGPU_OK;
}

int launch_count(jfgpu_table* t, const char* d_bases, size_t n) {
  if(n < t->g.k) return JFGPU_OK;
  if(!capacity_managed(t) || t->operation == 2) return launch_count_chunk(t, d_bases, n);   // an update pass adds no keys
  size_t off = 0;
  while(true) {
    uint64_t take = 0;
    int rc = ensure_capacity(t, n - off, &take); if(rc) return rc;
    // a piece holds at least one window and moves forward (spill mode may grant less than k characters of room; the few
    // extra k-mers cannot take the table past its 80 % bound by more than 2k)
    if(take < 2 * (uint64_t)t->g.k) take = std::min<uint64_t>(n - off, 2 * (uint64_t)t->g.k);
    t->cur_bound = take;
    rc = launch_count_chunk(t, d_bases + off, (size_t)take);
    t->cur_bound = 0;
    if(rc) return rc;
    if(off + take >= n) return JFGPU_OK;
    off += take - (t->g.k - 1);          // next piece re-reads the last k-1 characters: every window exactly once
  }
}

int ensure_stage(jfgpu_table* t) {
  for(int i = 0; i < 2; ++i) {
    if(!t->d_stage[i]) HIP_TRY(hipMalloc((void**)&t->d_stage[i], kStageBytes));
    if(!t->stage_done[i]) HIP_TRY(hipEventCreateWithFlags(&t->stage_done[i], hipEventDisableTiming));
  }
  return JFGPU_OK;
}


#include "host_partition.inl"

int measure_occupancy(jfgpu_table* t) {
  int rc = part_flush(t); if(rc) return rc;
  uint64_t c[CTR_COUNT];
  rc = check_deferred(t, c); if(rc) return rc;
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, 4 * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), t->stream));
  const int grid = grid_for(t, (1ull << t->g.lsize_l) / kBlock + 1);
  if(t->nword) hipLaunchKernelGGL(scan_nword_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->nt, 0, 0ull, ~0ull, 0, 0ull, 0ull, 1ull, 1ull, d, (uint32_t*)nullptr);
  else if(t->wide) hipLaunchKernelGGL(scan_wide_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->wt, 0, 0ull, ~0ull, 0, 0ull, 0ull, 1ull, 1ull, d, (uint32_t*)nullptr);
  else hipLaunchKernelGGL(stats_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, 0ull, ~0ull, 0, d);
  unsigned long long h[4];
  hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, t->stream);
  if(e == hipSuccess) e = hipStreamSynchronize(t->stream);
  hipFree(d);
  if(e != hipSuccess) return fail(JFGPU_E_HIP, hipGetErrorString(e));
  t->occ_known = h[1]; t->fed_since = 0;
  return JFGPU_OK;
}

// Double the table: one more matrix row (so old positions are the low bits of new ones), new slots,
// rehash on the device, swap.  < 0: not possible (memory) -- the caller carries on with the old table.
// In three parts, because a SHARD grows together with the other shards of its table (abi_comm.inl: comm_grow): the new
// geometry, matrix and allocations (grow_prepare), what moves the entries (the rehash kernels here; a redistribution
// between the ranks there), and the exchange of old for new (grow_swap).
struct GrowNew {
  TableGeom g2; WideGeom w2; NGeom ng2;
  Gf2Matrix m2;
  DevTable nd; WideTable nw; NTable nn;
  uint64_t *nf = nullptr, *ni = nullptr;
  uint64_t cap2 = 0;
  uint64_t ctr[CTR_COUNT];
};

int grow_prepare(jfgpu_table* t, GrowNew& N) {
  int rc = part_flush(t); if(rc) return rc;
  rc = check_deferred(t, N.ctr); if(rc) return rc;
  const uint32_t r = t->g.lsize_g, c = t->g.key_bits;
  if(r >= c) return -1;
  Gf2Matrix& m2 = N.m2;
  m2 = t->matrix; m2.r = r + 1; m2.identity = false;
  std::vector<uint64_t> fwd, inv;
  if(t->xs_matrix) {
    m2 = gf2_xorshift_matrix(r + 1, c);
    if(!gf2_build_tables(m2, fwd, inv)) return fail(JFGPU_E_INVALID, "xor-shift matrix: singular low block");
  } else if(t->ref_matrix) {
    // like the reference: a brand-new matrix for the doubled table, next in the same random() stream
    for(int tries = 0; ; ++tries) {
      m2 = r + 1 >= c ? gf2_identity(r + 1, c) : gf2_reference_matrix(r + 1, c, t->glibc);
      if(gf2_build_tables(m2, fwd, inv)) break;
      if(tries > 1000) return fail(JFGPU_E_INVALID, "could not draw a hash matrix");
    }
  } else {
    // seeded family: extend the matrix by a random top row until the (r+1) x (r+1) low block is invertible again
    uint64_t st = (t->params.matrix_seed ? t->params.matrix_seed : kDefaultSeed) ^ (0xD6E8FEB86659FD93ull * (r + 1)) ^ t->grow_seed;
    for(int tries = 0; ; ++tries) {
      for(uint32_t j = 0; j < c; ++j) m2.columns[j] = (t->matrix.columns[j] & ((1ull << r) - 1)) | ((splitmix64(st) & 1ull) << r);
      if(gf2_build_tables(m2, fwd, inv)) break;
      if(tries > 1000) return fail(JFGPU_E_INVALID, "could not extend the hash matrix");
    }
  }
  TableGeom& g2 = N.g2;
  if(t->nword) { if(!nword_geom_init(N.ng2, t->g.k, r + 1, t->g.canonical)) return -1; g2 = N.ng2.g; }
  else if(t->wide) { if(!wide_geom_init(N.w2, t->g.k, r + 1, t->g.canonical, t->g.shard_bits, t->g.shard_id)) return -1; g2 = N.w2.g; }
  else if(!geom_init(g2, t->g.k, r + 1, t->g.shard_bits, t->g.shard_id, t->g.canonical, !t->tun.slot64)) return -1;
  if(!t->nword) { g2.hash_xs = gf2_is_xorshift(m2) ? 1 : 0; if(t->wide) N.w2.g.hash_xs = g2.hash_xs; }
  const uint64_t n2 = 1ull << g2.lsize_l;
  const size_t slot_bytes = g2.slot32 ? 4 : 8 * (size_t)t->slot_words;
  uint64_t cap2 = std::max<uint64_t>(kMinOvf, std::min<uint64_t>(n2 / 256, 1ull << 26));
  if(g2.cnt_bits < 40) cap2 = std::max<uint64_t>(cap2, std::min<uint64_t>(2 * ((t->occ_bound >> g2.cnt_bits) + t->bigval_bound) + 4096, 2 * n2));      // (ensure_ovf's rule)
  cap2 = std::max<uint64_t>(cap2, t->ovf_cap);
  { uint64_t x = 1; while(x < cap2) x <<= 1; cap2 = x; }
  N.cap2 = cap2;
  DevTable& nd = N.nd;
  nd = t->dt;
  nd.g = g2; nd.slots = nullptr; nd.ovf_key = nd.ovf_cnt = nullptr; nd.dirty = nullptr;
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  if(n2 * slot_bytes + cap2 * 16 + ((size_t)1 << 30) > free_b) return -1;
  bool ok = hipMalloc((void**)&nd.slots, n2 * slot_bytes) == hipSuccess && hipMalloc((void**)&nd.ovf_key, cap2 * 8) == hipSuccess &&
            hipMalloc((void**)&nd.ovf_cnt, cap2 * 8) == hipSuccess && hipMalloc((void**)&nd.dirty, (size_t)1 << (g2.lsize_l - g2.tile_bits)) == hipSuccess &&
            hipMalloc((void**)&N.nf, fwd.size() * 8) == hipSuccess && hipMalloc((void**)&N.ni, inv.size() * 8) == hipSuccess;
  if(!ok) {
    hipFree(nd.slots); hipFree(nd.ovf_key); hipFree(nd.ovf_cnt); hipFree(nd.dirty); hipFree(N.nf); hipFree(N.ni);
    (void)hipGetLastError();
    return -1;
  }
  nd.fwd_tbl = N.nf; nd.inv_tbl = N.ni; nd.ovf_mask = cap2 - 1;
  nd.max_probe = (uint32_t)std::min<uint64_t>(g2.tile_mask, 1023);
  HIP_TRY(hipMemsetAsync(nd.slots, 0, n2 * slot_bytes, t->stream));
  HIP_TRY(hipMemsetAsync(nd.ovf_key, 0, cap2 * 8, t->stream));
  HIP_TRY(hipMemsetAsync(nd.ovf_cnt, 0, cap2 * 8, t->stream));
  HIP_TRY(hipMemsetAsync(nd.dirty, 0, (size_t)1 << (g2.lsize_l - g2.tile_bits), t->stream));
  HIP_TRY(hipMemcpyAsync(N.nf, fwd.data(), fwd.size() * 8, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipMemcpyAsync(N.ni, inv.data(), inv.size() * 8, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));                  // (fwd / inv are this function's)
  N.nw = t->wt; N.nn = t->nt;
  if(t->nword) {
    NTable& nn = N.nn;
    nn.N = N.ng2; nn.slots = nd.slots; nn.fwd_tbl = N.nf; nn.inv_tbl = N.ni; nn.ovf_key = nd.ovf_key; nn.ovf_cnt = nd.ovf_cnt;
    nn.ovf_mask = nd.ovf_mask; nn.counters = nd.counters; nn.max_probe = nd.max_probe;
  } else if(t->wide) {
    WideTable& nw = N.nw;
    nw.W = N.w2; nw.slots = nd.slots; nw.fwd_tbl = N.nf; nw.inv_tbl = N.ni; nw.ovf_key = nd.ovf_key; nw.ovf_cnt = nd.ovf_cnt;
    nw.ovf_mask = nd.ovf_mask; nw.counters = nd.counters; nw.max_probe = nd.max_probe; nw.dirty = nd.dirty;
  }
  return JFGPU_OK;
}

// the new table takes the old one's place (the entries have been moved and the stream is idle)
int grow_swap(jfgpu_table* t, GrowNew& N) {
  hipFree(t->dt.slots); hipFree(t->dt.ovf_key); hipFree(t->dt.ovf_cnt); hipFree(t->dt.dirty); hipFree(t->d_fwd); hipFree(t->d_inv);
  t->dt = N.nd; t->g = N.g2; t->matrix = N.m2; t->d_fwd = N.nf; t->d_inv = N.ni; t->ovf_cap = N.cap2;
  t->returning = t->g.cnt_bits < 40; t->ovf_failed_need = 0;
  if(t->d_M2) { hipFree(t->d_M2); t->d_M2 = nullptr; }
  if(t->d_strag) { hipFree(t->d_strag); hipFree(t->d_strag_n); t->d_strag = nullptr; t->d_strag_n = nullptr; }     // (the item width may change with the geometry)
  if(t->nword) {
    t->nt = N.nn;
    t->pristine = false;
    ++t->grow_seed;
    return check_deferred(t);
  }
  if(t->wide) {
    t->wt = N.nw;
    const int wl = (int)(((size_t)16 << t->g.tile_bits) + ((size_t)2 << t->g.tile_bits));
    HIP_TRY(hipFuncSetAttribute((const void*)dump_tiles_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wl));
  }
  part_geom_init(t);
  if(t->mode == MODE_PARTITIONED && !t->part_ok) t->mode = MODE_AUTO;
  t->pristine = false;
  ++t->grow_seed;
  return check_deferred(t);
}

int table_grow(jfgpu_table* t) {
  GrowNew N;
  int rc = grow_prepare(t, N); if(rc) return rc;
  const int have_ovf = (int)(N.ctr[CTR_OVF_USED] != 0);
  const dim3 grid(grid_for(t, (1ull << t->g.lsize_l) / kBlock + 1)), block(kBlock);
  if(t->nword) hipLaunchKernelGGL(rehash_nword_kernel, grid, block, 0, t->stream, t->nt, N.nn, have_ovf);
  else if(t->wide) hipLaunchKernelGGL(rehash_wide_kernel, grid, block, 0, t->stream, t->wt, N.nw, have_ovf);
  else hipLaunchKernelGGL(rehash_kernel, grid, block, 0, t->stream, t->dt, N.nd, have_ovf);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(t->stream));
  return grow_swap(t, N);
}

void refresh_views(jfgpu_table* t) {
  if(t->nword) { t->nt.ovf_key = t->dt.ovf_key; t->nt.ovf_cnt = t->dt.ovf_cnt; t->nt.ovf_mask = t->dt.ovf_mask; }
  if(t->wide) { t->wt.ovf_key = t->dt.ovf_key; t->wt.ovf_cnt = t->dt.ovf_cnt; t->wt.ovf_mask = t->dt.ovf_mask; }
}

bool use_partitioned(const jfgpu_table* t, size_t nbytes) {
  if(!t->part_ok || t->mode == MODE_DIRECT) return false;
  if(t->mode == MODE_PARTITIONED) return true;
  return nbytes >= kPartMinBytes || !t->pending.empty();
}

}  // namespace

extern "C" {

const char* jfgpu_last_error(void) { return g_err.c_str(); }
int jfgpu_abi_version(void) { return JFGPU_ABI_VERSION; }

int jfgpu_device_count(void) {
  int n = 0;
  if(hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int jfgpu_create(const jfgpu_params* p, jfgpu_table** out) {
  if(!p || !out) return fail(JFGPU_E_INVALID, "null argument");
  *out = nullptr;
  if(p->k < 1) return fail(JFGPU_E_INVALID, "mer length must be >= 1");
  if(p->k > 128) return fail(JFGPU_E_UNSUPPORTED, "mer length > 128 (more than four key words) is not built");
  if(p->k > 64 && p->shard_bits) return fail(JFGPU_E_UNSUPPORTED, "sharded tables with mer length > 64 are not built yet");
  if(p->shard_bits > 8) return fail(JFGPU_E_INVALID, "at most 256 shards");
  if(p->shard_id >= (1u << p->shard_bits)) return fail(JFGPU_E_INVALID, "shard_id out of range");
  int ndev = 0;
  if(hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(JFGPU_E_NO_DEVICE, "no HIP device: the engine has no CPU fallback");
  int dev = p->device;
  if(dev < 0) HIP_TRY(hipGetDevice(&dev));
  if(dev >= ndev) return fail(JFGPU_E_NO_DEVICE, "device ordinal out of range");
  HIP_TRY(hipSetDevice(dev));

  // size -> lsize (large_hash_array.hpp:156-157 rounds up to a power of two, :997-1000 caps at 4^k)
  const bool nword = p->k > 64, wide = p->k > 32 && !nword;
  uint32_t lsize = 0;
  while(lsize < 63 && (1ull << lsize) < p->size) ++lsize;
  if(nword) {
    lsize = std::max(lsize, nword_min_lsize(p->k));
    lsize = std::min<uint32_t>(lsize, 48);
  } else if(wide) {
    lsize = std::max(lsize, wide_min_lsize(p->k));
    lsize = std::max<uint32_t>(lsize, kMaxTileBits + p->shard_bits);       // a shard holds at least one tile
    lsize = std::min<uint32_t>(lsize, 48);
  } else {
    lsize = std::max(lsize, geom_min_lsize(p->k, p->shard_bits));
    lsize = std::max(lsize, p->shard_bits);
    lsize = std::max<uint32_t>(lsize, 1);
    lsize = std::min<uint32_t>(lsize, 2 * p->k);
    if(lsize < p->shard_bits) return fail(JFGPU_E_INVALID, "more shards than 4^k table positions");
  }

  std::unique_ptr<jfgpu_table> t(new jfgpu_table);
  t->tun = Tuning::from_env();
  t->tun.p2_cap = t->tun.p2_cap / kGran * kGran;
  t->params = *p; t->params.matrix_columns = nullptr;
  t->device = dev;
  t->out_counter_len = p->out_counter_len ? p->out_counter_len : 4;
  if(t->out_counter_len > 8) return fail(JFGPU_E_INVALID, "out_counter_len must be <= 8");
  t->wide = wide; t->nword = nword; t->key_words = (2 * p->k + 63) / 64; t->slot_words = nword ? kNWords : wide ? 2 : 1;
  if(nword) {
    if(!nword_geom_init(t->nt.N, p->k, lsize, p->canonical ? 1 : 0)) return fail(JFGPU_E_INVALID, "table geometry does not fit a 256-bit slot");
    t->g = t->nt.N.g;
  } else if(wide) {
    if(!wide_geom_init(t->wt.W, p->k, lsize, p->canonical ? 1 : 0, p->shard_bits, p->shard_id)) return fail(JFGPU_E_INVALID, "table geometry does not fit a 128-bit slot");
    t->g = t->wt.W.g;
  } else if(!geom_init(t->g, p->k, lsize, p->shard_bits, p->shard_id, p->canonical ? 1 : 0, !t->tun.slot64))
    return fail(JFGPU_E_INVALID, "table geometry does not fit a 64-bit slot");

  // hash matrix (large_hash_array.hpp:992-1001)
  if(p->matrix_columns) {
    t->matrix.r = lsize; t->matrix.c = 2 * p->k;
    t->matrix.columns.assign(p->matrix_columns, p->matrix_columns + 2 * p->k);
    t->matrix.identity = gf2_is_low_identity(t->matrix);
  } else if(p->matrix_seed) {
    t->matrix = gf2_random(lsize, 2 * p->k, p->matrix_seed);
  } else if(p->k <= 64 && lsize < 2 * p->k && (p->matrix_kind ? p->matrix_kind : (uint32_t)t->tun.matrix) == JFGPU_MATRIX_XORSHIFT) {
    t->xs_matrix = true;
    t->matrix = gf2_xorshift_matrix(lsize, 2 * p->k);
  } else {
    // the reference's default: first matrix of an unseeded glibc random() stream; the stream stays with the
    // table so that doublings draw their matrices like hash_counter::double_size does (hash_counter.hpp:210-214)
    t->ref_matrix = true;
    t->matrix = lsize >= 2 * p->k ? gf2_identity(lsize, 2 * p->k) : gf2_reference_matrix(lsize, 2 * p->k, t->glibc);
  }
  if(!nword) {                                                                 // (also a matrix given by its columns: a file header's)
    t->g.hash_xs = gf2_is_xorshift(t->matrix) ? 1 : 0;
    if(wide) t->wt.W.g.hash_xs = t->g.hash_xs;
  }
  std::vector<uint64_t> fwd, inv;
  while(!gf2_build_tables(t->matrix, fwd, inv)) {
    if(!t->ref_matrix || t->matrix.identity) return fail(JFGPU_E_INVALID, "hash matrix: low r x r block is singular");
    t->matrix = gf2_reference_matrix(lsize, 2 * p->k, t->glibc);        // cannot happen for a pseudo-inverse; belt and braces
  }

  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  t->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_TRY(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));

  const uint64_t n_slots = 1ull << t->g.lsize_l;
  t->ovf_cap = std::max<uint64_t>(kMinOvf, std::min<uint64_t>(n_slots / 256, 1ull << 26));
  { uint64_t c = 1; while(c < t->ovf_cap) c <<= 1; t->ovf_cap = c; }
  t->returning = t->g.cnt_bits < 40;

  DevTable& d = t->dt;
  d.g = t->g;
  HIP_TRY(hipMalloc((void**)&d.slots, n_slots * slot_bytes_of(t.get())));
  HIP_TRY(hipMalloc((void**)&t->d_fwd, fwd.size() * sizeof(uint64_t)));
  HIP_TRY(hipMalloc((void**)&t->d_inv, inv.size() * sizeof(uint64_t)));
  HIP_TRY(hipMalloc((void**)&d.ovf_key, t->ovf_cap * sizeof(uint64_t)));
  HIP_TRY(hipMalloc((void**)&d.ovf_cnt, t->ovf_cap * sizeof(uint64_t)));
  HIP_TRY(hipMalloc((void**)&d.counters, CTR_COUNT * sizeof(uint64_t)));
  HIP_TRY(hipMalloc((void**)&d.dirty, (size_t)1 << (t->g.lsize_l - t->g.tile_bits)));
  d.fwd_tbl = t->d_fwd; d.inv_tbl = t->d_inv; d.ovf_mask = t->ovf_cap - 1;
  // Give up on a tile after 1024 probes (load > 99.8%); the reference gives up after 126
  // (count_main_cmdline.yaggo -p).  Small tiles are probed exhaustively.
  d.max_probe = (uint32_t)std::min<uint64_t>(t->g.tile_mask, 1023);
  HIP_TRY(hipMemcpy(t->d_fwd, fwd.data(), fwd.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(t->d_inv, inv.data(), inv.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  // dump kernel needs > 64 KiB of dynamic LDS
  const size_t dump_lds = ((size_t)8 << t->g.tile_bits) + ((size_t)2 << t->g.tile_bits) + (size_t)t->g.nbytes * 2048;
  HIP_TRY(hipFuncSetAttribute((const void*)dump_tiles_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dump_lds));
  if(nword) {
    NTable& w = t->nt;
    w.slots = d.slots; w.fwd_tbl = d.fwd_tbl; w.inv_tbl = d.inv_tbl; w.ovf_key = d.ovf_key; w.ovf_cnt = d.ovf_cnt;
    w.ovf_mask = d.ovf_mask; w.counters = d.counters; w.max_probe = d.max_probe;
    t->part_ok = false; t->mode = MODE_DIRECT;
    const int nl = (int)(((size_t)32 << kNTileBits) + ((size_t)2 << kNTileBits));
    HIP_TRY(hipFuncSetAttribute((const void*)dump_tiles_nword_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, nl));
  } else if(wide) {
    WideTable& w = t->wt;
    w.slots = d.slots; w.fwd_tbl = d.fwd_tbl; w.inv_tbl = d.inv_tbl; w.ovf_key = d.ovf_key; w.ovf_cnt = d.ovf_cnt;
    w.ovf_mask = d.ovf_mask; w.counters = d.counters; w.max_probe = d.max_probe; w.dirty = d.dirty;
    memset(&w.bloom, 0, sizeof w.bloom);
    part_geom_init(t.get());
    const int wl = (int)(((size_t)16 << t->g.tile_bits) + ((size_t)2 << t->g.tile_bits));
    HIP_TRY(hipFuncSetAttribute((const void*)dump_tiles_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wl));
  } else part_geom_init(t.get());
  if(!nword) {
    if(t->tun.mode == 1) t->mode = MODE_DIRECT;
    else if(t->tun.mode == 2 && t->part_ok) t->mode = MODE_PARTITIONED;
  }
  {
#define TATTR1(I, R, S, P, H, M) HIP_TRY(hipFuncSetAttribute((const void*)tile_rank_insert_kernel<I, R, S, P, kTileBlock, H, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_rank_lds(sizeof(S), kMaxTileBits, P)))
#define TATTR(I, S, P) TATTR1(I, true, S, P, false, false); TATTR1(I, false, S, P, false, false); TATTR1(I, true, S, P, true, false); TATTR1(I, false, S, P, true, false); \
                       TATTR1(I, true, S, P, false, true); TATTR1(I, false, S, P, false, true)
    TATTR(uint32_t, unsigned int, 1); TATTR(uint32_t, unsigned int, 2); TATTR(uint32_t, unsigned long long, 1);
    TATTR(uint64_t, unsigned int, 1); TATTR(uint64_t, unsigned int, 2); TATTR(uint64_t, unsigned long long, 1);
#undef TATTR
#undef TATTR1
    HIP_TRY(hipFuncSetAttribute((const void*)p1_scatter_sorted_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kPTilePos * 6));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_scatter_sorted_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kPTilePos * 6));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_scatter_sorted_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kPTilePos * 6));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_scatter_sorted_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kPTilePos * 6));
#define PATTR(N) HIP_TRY(hipFuncSetAttribute((const void*)p1_keys_scatter_sorted_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, kPTilePos * 6)); \
                 HIP_TRY(hipFuncSetAttribute((const void*)p1_scatter_sorted_kernel<false, false, N>, hipFuncAttributeMaxDynamicSharedMemorySize, kPTilePos * 6))
    PATTR(0); PATTR(6); PATTR(7); PATTR(8);
#undef PATTR
    HIP_TRY(hipFuncSetAttribute((const void*)p1_keys_granule_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kPTilePos * 6));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_keys_granule_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kPTilePos * 6));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_keys_granule_kernel<false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, kPTilePos * 6));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_keys_granule_kernel<false, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, kPTilePos * 6));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_keys_granule_kernel<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, kPTilePos * 6));
    {
      const int rl = kGranMaxB * 128 + 128;
#define RATTR(IT, BL, N, CN) HIP_TRY(hipFuncSetAttribute((const void*)p1_ring_kernel<IT, BL, N, CN>, hipFuncAttributeMaxDynamicSharedMemorySize, rl))
      HIP_TRY(hipFuncSetAttribute((const void*)p2_ring_kernel<P2RingDirect>, hipFuncAttributeMaxDynamicSharedMemorySize, rl));
      HIP_TRY(hipFuncSetAttribute((const void*)p2_ring_roles_kernel<uint32_t, 2, P2RingDirect, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, rl));
      HIP_TRY(hipFuncSetAttribute((const void*)p2_ring_roles_kernel<uint32_t, 1, P2RingDirect, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, rl));
      HIP_TRY(hipFuncSetAttribute((const void*)p2_ring_roles_kernel<uint32_t, 2, P2RingDirect, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, rl));
      HIP_TRY(hipFuncSetAttribute((const void*)p2_ring_roles_kernel<uint32_t, 1, P2RingDirect, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, rl));
      HIP_TRY(hipFuncSetAttribute((const void*)p2_ring_roles_kernel<uint32_t, 2, P2RingDirect, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, rl));
      HIP_TRY(hipFuncSetAttribute((const void*)p2_ring_roles_kernel<uint32_t, 1, P2RingDirect, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, rl));
      RATTR(uint32_t, false, 6, 1); RATTR(uint32_t, false, 6, 0); RATTR(uint32_t, true, 0, 2); RATTR(uint32_t, false, 0, 2);
      RATTR(uint32_t, false, kHashXS, 1); RATTR(uint32_t, false, kHashXS, 0); RATTR(uint32_t, false, kHashXSLow, 2);
#undef RATTR
    }
    HIP_TRY(hipFuncSetAttribute((const void*)p2_scatter_sorted_kernel<uint32_t, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * 16 * 4));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_ring_kernel<uint32_t, false, 6, 1, RouteListDirect>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kGranMaxB * 128 + 128)));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_ring_kernel<uint32_t, false, 6, 0, RouteListDirect>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kGranMaxB * 128 + 128)));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_ring_kernel<uint32_t, false, 0, 2, RouteListDirect>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kGranMaxB * 128 + 128)));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_ring_kernel<uint32_t, true, 0, 2, RouteListDirect>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kGranMaxB * 128 + 128)));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_ring_kernel<uint32_t, false, kHashXS, 1, RouteListDirect>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kGranMaxB * 128 + 128)));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_ring_kernel<uint32_t, false, kHashXS, 0, RouteListDirect>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kGranMaxB * 128 + 128)));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_ring_kernel<uint32_t, false, kHashXSLow, 2, RouteListDirect>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kGranMaxB * 128 + 128)));
#define SATTR(SW) HIP_TRY(hipFuncSetAttribute((const void*)p2_granule_kernel<uint32_t, TableDirect<true>, kP2PairPer, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * kP2PairPer * 4)); \
                  HIP_TRY(hipFuncSetAttribute((const void*)p2_granule_kernel<uint32_t, TableDirect<false>, kP2PairPer, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * kP2PairPer * 4))
    SATTR(1); SATTR(2); SATTR(4);
#undef SATTR
    HIP_TRY(hipFuncSetAttribute((const void*)p2_granule_kernel<uint64_t, TableDirect<true>, kP2MidPer>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * kP2MidPer * 8));
    HIP_TRY(hipFuncSetAttribute((const void*)p2_granule_kernel<uint64_t, TableDirect<false>, kP2MidPer>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * kP2MidPer * 8));
    HIP_TRY(hipFuncSetAttribute((const void*)p2_granule_kernel<uint32_t, TableDirect<true>, kP2PairPer>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * kP2PairPer * 4));
    HIP_TRY(hipFuncSetAttribute((const void*)p2_granule_kernel<uint32_t, TableDirect<false>, kP2PairPer>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * kP2PairPer * 4));
    HIP_TRY(hipFuncSetAttribute((const void*)p2_scatter_sorted_kernel<uint32_t, kP2PairPer>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * kP2PairPer * 4));
    HIP_TRY(hipFuncSetAttribute((const void*)p2_scatter_sorted_kernel<uint64_t, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * 14 * 8));
    const int gl = kG64Chunk * 10;
    HIP_TRY(hipFuncSetAttribute((const void*)p1_granule64_kernel<true, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, gl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_granule64_kernel<false, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, gl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_granule64_kernel<true, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, gl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_granule64_kernel<false, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, gl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_granule64_kernel<false, false, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, gl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_granule64_kernel<false, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, gl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_granule64_kernel<false, false, kHashXS>, hipFuncAttributeMaxDynamicSharedMemorySize, gl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_granule64_kernel<true, false, kHashXS>, hipFuncAttributeMaxDynamicSharedMemorySize, gl));
    HIP_TRY(hipFuncSetAttribute((const void*)p2_granule_kernel<u128, WideDirect<true>, kP2WidePer>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * kP2WidePer * 16));
    HIP_TRY(hipFuncSetAttribute((const void*)p2_granule_kernel<u128, WideDirect<false>, kP2WidePer>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * kP2WidePer * 16));
    HIP_TRY(hipFuncSetAttribute((const void*)p2_scatter_sorted_kernel<u128, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * 7 * 16));
    const int wl = kWideChunk * 18 + 16 * 2048, wt = 16 << kMaxTileBits;
    HIP_TRY(hipFuncSetAttribute((const void*)p1_wide_granule_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_wide_granule_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, wl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_wide_granule_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_wide_granule_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, wl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_wide_granule_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_wide_granule_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wl));
    HIP_TRY(hipFuncSetAttribute((const void*)tile_insert_wide_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, wt));
    HIP_TRY(hipFuncSetAttribute((const void*)tile_insert_wide_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, wt));
    HIP_TRY(hipFuncSetAttribute((const void*)tile_insert_wide_pipe_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, wt));
    HIP_TRY(hipFuncSetAttribute((const void*)tile_insert_wide_pipe_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, wt));
  }
  jfgpu_table* raw = t.release();
  int rc = jfgpu_clear(raw);
  if(rc) { jfgpu_destroy(raw); return rc; }
  *out = raw;
  return JFGPU_OK;
}

void jfgpu_destroy(jfgpu_table* t) {
  if(!t) return;
  hipSetDevice(t->device);
  if(t->stream) hipStreamSynchronize(t->stream);
  prof_collect(t);
  for(auto e : t->ev_pool) hipEventDestroy(e);
  hipFree(t->dt.slots); hipFree(t->d_fwd); hipFree(t->d_inv);
  if(t->d_dt) hipFree(t->d_dt);
  hipFree(t->dt.ovf_key); hipFree(t->dt.ovf_cnt); hipFree(t->dt.counters); hipFree(t->dt.dirty);
  for(int i = 0; i < 2; ++i) { if(t->d_stage[i]) hipFree(t->d_stage[i]); if(t->stage_done[i]) hipEventDestroy(t->stage_done[i]); }
  if(t->d_dump) hipFree(t->d_dump);
  if(t->d_bcache) hipFree(t->d_bcache);
  if(t->d_tile_off) hipFree(t->d_tile_off);
  part_discard(t);
  if(t->d_M1) hipFree(t->d_M1);
  if(t->d_strag) hipFree(t->d_strag);
  if(t->d_strag2) hipFree(t->d_strag2);
  if(t->d_strag2_n) hipFree(t->d_strag2_n);
  if(t->d_strag_n) hipFree(t->d_strag_n);
  if(t->d_M2) hipFree(t->d_M2);
  if(t->ws) hipFree(t->ws);
  if(t->stream2) hipStreamDestroy(t->stream2);
  for(auto ev : t->flush_ev) if(ev) hipEventDestroy(ev);
  if(t->flush_done) hipEventDestroy(t->flush_done);
  if(t->stream) hipStreamDestroy(t->stream);
  delete t;
}

int jfgpu_get_info(const jfgpu_table* t, jfgpu_info* o) {
  if(!t || !o) return fail(JFGPU_E_INVALID, "null argument");
  memset(o, 0, sizeof(*o));
  o->k = t->g.k; o->key_len = t->g.key_bits; o->canonical = t->g.canonical;
  o->lsize = t->g.lsize_g; o->size = 1ull << t->g.lsize_g; o->local_size = 1ull << t->g.lsize_l;
  o->shard_bits = t->g.shard_bits; o->shard_id = t->g.shard_id;
  o->val_len = t->g.cnt_bits; o->slot_bytes = (uint32_t)slot_bytes_of(t); o->tile_slots = 1u << t->g.tile_bits;
  o->matrix_identity = t->matrix.identity ? 1 : 0;
  o->out_counter_len = t->out_counter_len;
  o->max_reprobe = t->dt.max_probe;
  o->table_bytes = (1ull << t->g.lsize_l) * slot_bytes_of(t);
  return JFGPU_OK;
}

int jfgpu_get_matrix(const jfgpu_table* t, uint64_t* columns) {
  if(!t || !columns) return fail(JFGPU_E_INVALID, "null argument");
  memcpy(columns, t->matrix.columns.data(), sizeof(uint64_t) * t->matrix.c);
  return JFGPU_OK;
}

int jfgpu_clear(jfgpu_table* t) {
  if(t) { t->failed = 0; t->failed_msg.clear(); }              // (what a failed flush left behind goes with everything else)
  int rc = use(t); if(rc) return rc;
  part_discard(t);
  HIP_TRY(hipMemsetAsync(t->dt.slots, 0, (1ull << t->g.lsize_l) * slot_bytes_of(t), t->stream));
  HIP_TRY(hipMemsetAsync(t->dt.ovf_key, 0, t->ovf_cap * sizeof(uint64_t), t->stream));
  HIP_TRY(hipMemsetAsync(t->dt.ovf_cnt, 0, t->ovf_cap * sizeof(uint64_t), t->stream));
  HIP_TRY(hipMemsetAsync(t->dt.counters, 0, CTR_COUNT * sizeof(uint64_t), t->stream));
  HIP_TRY(hipMemsetAsync(t->dt.dirty, 0, (size_t)1 << (t->g.lsize_l - t->g.tile_bits), t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  t->pristine = true; t->occ_known = 0; t->fed_since = 0; t->direct_seen = 0; t->occ_bound = 0; t->bigval_bound = 0; t->ovf_failed_need = 0;
  t->flushes_plain = 0; t->flushes_heavy = 0;
  t->n_p2_roles = t->n_p2_ring = t->n_p2_sort = t->n_p2_exact = t->n_p1_ring = t->n_p1_other = 0;
  return JFGPU_OK;
}

int jfgpu_sync(jfgpu_table* t) {
  int rc = use(t); if(rc) return rc;
  rc = part_flush(t); if(rc) return rc;
  HIP_TRY(hipStreamSynchronize(t->stream));
#ifdef JFGPU_PHASE_PROF
  { unsigned long long c[24], z[24] = {0};
    if(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_phase_prof), sizeof c) == hipSuccess) {
      fprintf(stderr, "[phase prof] P1 (ring): stage %llu  encode+hash+append %llu  (barrier) %llu  units out %llu  finish %llu\n",
              c[0], c[1], c[2], c[3], c[4]);
      fprintf(stderr, "[phase prof] P2: (barrier) %llu  load %llu  hist %llu  scan %llu  lds-scatter %llu  write-out %llu  finish %llu\n",
              c[8], c[9], c[10], c[11], c[12], c[13], c[14]);
      fprintf(stderr, "[phase prof] T: loop+offsets %llu  rank adds + fill %llu  place %llu  merge %llu  queue (wave 0's own) %llu  (its wait for the others) %llu  store %llu\n",
              c[16], c[17], c[18], c[22], c[21], c[19], c[20]);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_prof), z, sizeof z);
    } }
#endif
#ifdef JFGPU_TILE_PROF
  { uint64_t c[CTR_COUNT]; if(read_counters(t, c) == JFGPU_OK)
      fprintf(stderr, "[tile prof] wait+prefetch %llu  fill %llu  insert %llu  store %llu (shader clocks, summed over blocks)\n",
              (unsigned long long)c[CTR_PROF0], (unsigned long long)c[CTR_PROF0 + 1], (unsigned long long)c[CTR_PROF0 + 2], (unsigned long long)c[CTR_PROF0 + 3]); }
#endif
  return check_deferred(t);
}

int jfgpu_wait(jfgpu_table* t) {
  int rc = use(t); if(rc) return rc;
  HIP_TRY(hipStreamSynchronize(t->stream));
  return JFGPU_OK;
}

int jfgpu_count_ascii_dev(jfgpu_table* t, const char* d_bases, size_t n) {
  int rc = use(t); if(rc) return rc;
  if(t->g.shard_bits) return fail(JFGPU_E_INVALID, "sharded table: count through its communicator (jfgpu_comm_count_ascii_dev)");
  if(!d_bases && n) return fail(JFGPU_E_INVALID, "null buffer");
  return launch_count(t, d_bases, n);
}

int jfgpu_count_ascii(jfgpu_table* t, const char* bases, size_t n) {
  int rc = use(t); if(rc) return rc;
  if(t->g.shard_bits) return fail(JFGPU_E_INVALID, "sharded table: count through its communicator (jfgpu_comm_count_ascii_dev)");
  if(!bases && n) return fail(JFGPU_E_INVALID, "null buffer");
  if(n < t->g.k) return JFGPU_OK;
  rc = ensure_stage(t); if(rc) return rc;
  // Chunks overlap by k-1 bytes so that every window is seen exactly once (the same
  // "seam" idea as mer_overlap_sequence_parser.hpp:164-167,182-184).
  const size_t step = kStageBytes - (t->g.k - 1);
  for(size_t o = 0; o < n; o += step) {
    const size_t len = std::min(kStageBytes, n - o);
    const int b = t->stage_next; t->stage_next ^= 1;
    HIP_TRY(hipEventSynchronize(t->stage_done[b]));  // kernel that last read this buffer has finished
    HIP_TRY(hipMemcpyAsync(t->d_stage[b], bases + o, len, hipMemcpyHostToDevice, t->stream));
    rc = launch_count(t, (const char*)t->d_stage[b], len); if(rc) return rc;
    HIP_TRY(hipEventRecord(t->stage_done[b], t->stream));
    if(o + len >= n) break;
  }
  return JFGPU_OK;
}

// One piece of an add_keys batch that is known to fit (the capacity check was made by the caller).
static int add_keys_piece(jfgpu_table* t, const uint64_t* d_keys, size_t n, uint64_t val, uint8_t* d_is_new) {
  int rc = JFGPU_OK;
  if(!n) return JFGPU_OK;
  { const bool big = t->g.cnt_bits < 64 && (val >> t->g.cnt_bits) != 0;
    rc = ensure_ovf(t, big ? 0 : (uint64_t)n * val, big ? n : 0); if(rc) return rc; }
  if(t->nword) {
    ProfScope ps(t, 1, n);
    t->pristine = false;
    hipLaunchKernelGGL(add_keys_nword_kernel, dim3(grid_for(t, (n + kBlock - 1) / kBlock)), dim3(kBlock), 0, t->stream, t->nt, d_keys, (uint64_t)n, t->key_words, val, d_is_new);
    HIP_TRY(hipGetLastError());
    return JFGPU_OK;
  }
  if(t->wide) {
    ProfScope ps(t, 1, n);
    hipLaunchKernelGGL(add_keys_wide_kernel, dim3(grid_for(t, (n + kBlock - 1) / kBlock)), dim3(kBlock), 0, t->stream, t->wt, d_keys, (uint64_t)n, val, d_is_new);
    HIP_TRY(hipGetLastError());
    return JFGPU_OK;
  }
  if(val == 1 && !d_is_new && use_partitioned(t, n * 8)) {
    const int prc = part_ingest(t, (const uint8_t*)d_keys, 0, (int64_t)n, true, n);
    if(prc >= 0) return prc;
  }
  if(d_is_new || val != 1) { rc = part_flush(t); if(rc) return rc; }   // is_new / set() must see earlier adds
  t->pristine = false;
  const int grid = grid_for(t, (n + kBlock - 1) / kBlock);
  ProfScope ps(t, 1, n);
  if(val == 1 && !d_is_new) {
    if(t->returning) hipLaunchKernelGGL(add_keys_one_kernel<true>, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, d_keys, (uint64_t)n);
    else             hipLaunchKernelGGL(add_keys_one_kernel<false>, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, d_keys, (uint64_t)n);
  } else {
    hipLaunchKernelGGL(add_keys_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, d_keys, (uint64_t)n, val, d_is_new);
  }
  HIP_TRY(hipGetLastError());
  return JFGPU_OK;
}

int jfgpu_add_keys_dev(jfgpu_table* t, const uint64_t* d_keys, size_t n, uint64_t val, uint8_t* d_is_new) {
  int rc = use(t); if(rc) return rc;
  if(!n) return JFGPU_OK;
  if(!d_keys) return fail(JFGPU_E_INVALID, "null keys");
  if(!capacity_managed(t)) return add_keys_piece(t, d_keys, n, val, d_is_new);
  // In order, piece by piece: each piece is enqueued before the occupancy is looked at again, so a batch several
  // times the table's size doubles (or spills) the table as often as it has to (hash_counter::add, hash_counter.hpp:91-115)
  size_t off = 0;
  while(off < n) {
    uint64_t take = 0;
    rc = ensure_capacity(t, n - off, &take); if(rc) return rc;
    rc = add_keys_piece(t, d_keys + off * t->key_words, (size_t)take, val, d_is_new ? d_is_new + off : nullptr); if(rc) return rc;
    off += (size_t)take;
  }
  return JFGPU_OK;
}

int jfgpu_add_keys(jfgpu_table* t, const uint64_t* keys, size_t n, uint64_t val, uint8_t* is_new) {
  int rc = use(t); if(rc) return rc;
  if(!n) return JFGPU_OK;
  uint64_t* d_k = nullptr; uint8_t* d_n = nullptr;
  HIP_TRY(hipMalloc((void**)&d_k, n * sizeof(uint64_t) * t->key_words));
  if(is_new && hipMalloc((void**)&d_n, n) != hipSuccess) { hipFree(d_k); return fail(JFGPU_E_ALLOC, "hipMalloc is_new"); }
  hipError_t e = hipMemcpyAsync(d_k, keys, n * sizeof(uint64_t) * t->key_words, hipMemcpyHostToDevice, t->stream);
  if(e == hipSuccess) { rc = jfgpu_add_keys_dev(t, d_k, n, val, d_n); }
  if(e == hipSuccess && !rc && is_new) e = hipMemcpyAsync(is_new, d_n, n, hipMemcpyDeviceToHost, t->stream);
  hipStreamSynchronize(t->stream);
  hipFree(d_k); if(d_n) hipFree(d_n);
  if(e != hipSuccess) return fail(JFGPU_E_HIP, hipGetErrorString(e));
  if(rc) return rc;
  return check_deferred(t);
}

int jfgpu_lookup_dev(jfgpu_table* t, const uint64_t* d_keys, size_t n, uint64_t* d_vals, uint8_t* d_found) {
  int rc = use(t); if(rc) return rc;
  rc = part_flush(t); if(rc) return rc;
  if(!n) return JFGPU_OK;
  if(!d_keys || !d_vals) return fail(JFGPU_E_INVALID, "null argument");
  uint64_t c[CTR_COUNT];
  rc = check_deferred(t, c); if(rc) return rc;
  const int grid = grid_for(t, (n + kBlock - 1) / kBlock);
  ProfScope ps(t, 3, n);
  if(t->nword) {
    hipLaunchKernelGGL(lookup_nword_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->nt, d_keys, (uint64_t)n, t->key_words, d_vals, d_found, (int)(c[CTR_OVF_USED] != 0));
    HIP_TRY(hipGetLastError());
    return JFGPU_OK;
  }
  if(t->wide) {
    hipLaunchKernelGGL(lookup_wide_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->wt, d_keys, (uint64_t)n, d_vals, d_found, (int)(c[CTR_OVF_USED] != 0));
    HIP_TRY(hipGetLastError());
    return JFGPU_OK;
  }
  hipLaunchKernelGGL(lookup_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, d_keys, (uint64_t)n, d_vals, d_found,
                     (int)(c[CTR_OVF_USED] != 0));
  HIP_TRY(hipGetLastError());
  return JFGPU_OK;
}

int jfgpu_lookup(jfgpu_table* t, const uint64_t* keys, size_t n, uint64_t* vals, uint8_t* found) {
  int rc = use(t); if(rc) return rc;
  if(!n) return JFGPU_OK;
  uint64_t *d_k = nullptr, *d_v = nullptr; uint8_t* d_f = nullptr;
  HIP_TRY(hipMalloc((void**)&d_k, n * sizeof(uint64_t) * t->key_words));
  if(hipMalloc((void**)&d_v, n * sizeof(uint64_t)) != hipSuccess || hipMalloc((void**)&d_f, n) != hipSuccess) {
    hipFree(d_k); if(d_v) hipFree(d_v);
    return fail(JFGPU_E_ALLOC, "hipMalloc lookup buffers");
  }
  hipError_t e = hipMemcpyAsync(d_k, keys, n * sizeof(uint64_t) * t->key_words, hipMemcpyHostToDevice, t->stream);
  if(e == hipSuccess) rc = jfgpu_lookup_dev(t, d_k, n, d_v, d_f);
  if(e == hipSuccess && !rc) e = hipMemcpyAsync(vals, d_v, n * sizeof(uint64_t), hipMemcpyDeviceToHost, t->stream);
  if(e == hipSuccess && !rc && found) e = hipMemcpyAsync(found, d_f, n, hipMemcpyDeviceToHost, t->stream);
  hipStreamSynchronize(t->stream);
  hipFree(d_k); hipFree(d_v); hipFree(d_f);
  if(e != hipSuccess) return fail(JFGPU_E_HIP, hipGetErrorString(e));
  return rc;
}

int jfgpu_partition_ascii_dev(jfgpu_table* t, const char* d_bases, size_t n, uint64_t* d_keys_out, size_t capacity,
                              uint64_t* counts_out) {
  int rc = use(t); if(rc) return rc;
  if(!counts_out) return fail(JFGPU_E_INVALID, "null counts_out");
  if(t->wide || t->nword) return fail(JFGPU_E_UNSUPPORTED, "hash-prefix partition with mer length > 32 is not built yet");
  const uint32_t n_shards = 1u << t->g.shard_bits;
  for(uint32_t i = 0; i < n_shards; ++i) counts_out[i] = 0;
  if(n < t->g.k) return JFGPU_OK;
  if(!d_bases || !d_keys_out) return fail(JFGPU_E_INVALID, "null buffer");
  unsigned long long* d_cnt = nullptr;
  HIP_TRY(hipMalloc((void**)&d_cnt, sizeof(unsigned long long) * n_shards));
  HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long) * n_shards, t->stream));
  const uint8_t* base; int64_t lo, hi;
  align_buffer(d_bases, n, base, lo, hi);
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  const int grid = grid_for(t, (uint64_t)n_tiles);
  {
    ProfScope ps(t, 2, n);
    hipLaunchKernelGGL(partition_count_kernel<false>, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, base, lo, hi, d_cnt);
  }
  std::vector<unsigned long long> h(n_shards);
  hipError_t e = hipMemcpyAsync(h.data(), d_cnt, sizeof(unsigned long long) * n_shards, hipMemcpyDeviceToHost, t->stream);
  if(e == hipSuccess) e = hipStreamSynchronize(t->stream);
  if(e != hipSuccess) { hipFree(d_cnt); return fail(JFGPU_E_HIP, hipGetErrorString(e)); }
  uint64_t total = 0;
  std::vector<unsigned long long> offs(n_shards);
  for(uint32_t i = 0; i < n_shards; ++i) { offs[i] = total; total += h[i]; counts_out[i] = h[i]; }
  if(total > capacity) { hipFree(d_cnt); return fail(JFGPU_E_INVALID, "partition buffer too small: need " + std::to_string(total) + " keys"); }
  e = hipMemcpyAsync(d_cnt, offs.data(), sizeof(unsigned long long) * n_shards, hipMemcpyHostToDevice, t->stream);
  if(e == hipSuccess) {
    ProfScope ps(t, 2, 0);
    hipLaunchKernelGGL(partition_scatter_kernel<false>, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, base, lo, hi, d_cnt, d_keys_out);
    e = hipGetLastError();
  }
  if(e == hipSuccess) e = hipStreamSynchronize(t->stream);
  hipFree(d_cnt);
  if(e != hipSuccess) return fail(JFGPU_E_HIP, hipGetErrorString(e));
  return JFGPU_OK;
}

int jfgpu_stats_compute(jfgpu_table* t, uint64_t lower, uint64_t upper, jfgpu_stats* out) {
  int rc = use(t); if(rc) return rc;
  rc = part_flush(t); if(rc) return rc;
  if(!out) return fail(JFGPU_E_INVALID, "null out");
  uint64_t c[CTR_COUNT];
  rc = check_deferred(t, c); if(rc) return rc;
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, 4 * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), t->stream));
  const int grid = grid_for(t, (1ull << t->g.lsize_l) / kBlock + 1);
  if(t->nword) hipLaunchKernelGGL(scan_nword_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->nt, 0, lower, upper, (int)(c[CTR_OVF_USED] != 0), 0ull, 0ull, 1ull, 1ull, d, (uint32_t*)nullptr);
  else if(t->wide) hipLaunchKernelGGL(scan_wide_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->wt, 0, lower, upper, (int)(c[CTR_OVF_USED] != 0), 0ull, 0ull, 1ull, 1ull, d, (uint32_t*)nullptr);
  else hipLaunchKernelGGL(stats_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, lower, upper, (int)(c[CTR_OVF_USED] != 0), d);
  unsigned long long h[4];
  hipError_t e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, t->stream);
  if(e == hipSuccess) e = hipStreamSynchronize(t->stream);
  hipFree(d);
  if(e != hipSuccess) return fail(JFGPU_E_HIP, hipGetErrorString(e));
  out->unique = h[0]; out->distinct = h[1]; out->total = h[2]; out->max_count = h[3];
  out->occupied = h[1]; out->mers_fed = c[CTR_MERS];
  return JFGPU_OK;
}

int jfgpu_digest(jfgpu_table* t, uint64_t lower, uint64_t upper, uint64_t* out4) {
  int rc = use(t); if(rc) return rc;
  rc = part_flush(t); if(rc) return rc;
  if(!out4) return fail(JFGPU_E_INVALID, "null out");
  uint64_t c[CTR_COUNT];
  rc = check_deferred(t, c); if(rc) return rc;
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, 4 * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), t->stream));
  const int grid = grid_for(t, (1ull << t->g.lsize_l) / kBlock + 1);
  if(t->nword) hipLaunchKernelGGL(scan_nword_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->nt, 3, lower, upper, (int)(c[CTR_OVF_USED] != 0), 0ull, 0ull, 1ull, 1ull, d, (uint32_t*)nullptr);
  else if(t->wide) hipLaunchKernelGGL(digest_wide_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->wt, lower, upper, (int)(c[CTR_OVF_USED] != 0), d);
  else hipLaunchKernelGGL(digest_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, lower, upper, (int)(c[CTR_OVF_USED] != 0), d);
  hipError_t e = hipMemcpyAsync(out4, d, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, t->stream);
  if(e == hipSuccess) e = hipStreamSynchronize(t->stream);
  hipFree(d);
  if(e != hipSuccess) return fail(JFGPU_E_HIP, hipGetErrorString(e));
  return JFGPU_OK;
}

int jfgpu_histo(jfgpu_table* t, uint64_t base, uint64_t ceil, uint64_t inc, uint64_t* histo, uint64_t nb) {
  int rc = use(t); if(rc) return rc;
  rc = part_flush(t); if(rc) return rc;
  if(!histo || !nb || !inc) return fail(JFGPU_E_INVALID, "bad histogram arguments");
  uint64_t c[CTR_COUNT];
  rc = check_deferred(t, c); if(rc) return rc;
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, nb * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(d, 0, nb * sizeof(unsigned long long), t->stream));
  const int grid = grid_for(t, (1ull << t->g.lsize_l) / kBlock + 1);
  if(t->nword) hipLaunchKernelGGL(scan_nword_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->nt, 1, 0ull, ~0ull, (int)(c[CTR_OVF_USED] != 0), base, ceil, inc, nb, d, (uint32_t*)nullptr);
  else if(t->wide) hipLaunchKernelGGL(scan_wide_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->wt, 1, 0ull, ~0ull, (int)(c[CTR_OVF_USED] != 0), base, ceil, inc, nb, d, (uint32_t*)nullptr);
  else hipLaunchKernelGGL(histo_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, base, ceil, inc, nb, (int)(c[CTR_OVF_USED] != 0), d);
  hipError_t e = hipMemcpyAsync(histo, d, nb * sizeof(uint64_t), hipMemcpyDeviceToHost, t->stream);
  if(e == hipSuccess) e = hipStreamSynchronize(t->stream);
  hipFree(d);
  if(e != hipSuccess) return fail(JFGPU_E_HIP, hipGetErrorString(e));
  return JFGPU_OK;
}

int jfgpu_dump_begin(jfgpu_table* t, uint64_t lower, uint64_t upper, uint64_t* n_records, uint32_t* record_bytes) {
  int rc = use(t); if(rc) return rc;
  rc = part_flush(t); if(rc) return rc;
  uint64_t c[CTR_COUNT];
  rc = check_deferred(t, c); if(rc) return rc;
  const uint64_t nt = n_tiles_of(t);
  uint32_t* d_cnt = nullptr;
  HIP_TRY(hipMalloc((void**)&d_cnt, nt * sizeof(uint32_t)));
  t->dump_have_ovf = c[CTR_OVF_USED] != 0;
  if(t->nword) {
    HIP_TRY(hipMemsetAsync(d_cnt, 0, nt * sizeof(uint32_t), t->stream));
    hipLaunchKernelGGL(scan_nword_kernel, dim3(grid_for(t, (1ull << t->g.lsize_l) / kBlock + 1)), dim3(kBlock), 0, t->stream, t->nt, 2, lower, upper,
                       t->dump_have_ovf, 0ull, 0ull, 1ull, 1ull, (unsigned long long*)nullptr, d_cnt);
  } else if(t->wide) {
    HIP_TRY(hipMemsetAsync(d_cnt, 0, nt * sizeof(uint32_t), t->stream));
    hipLaunchKernelGGL(scan_wide_kernel, dim3(grid_for(t, (1ull << t->g.lsize_l) / kBlock + 1)), dim3(kBlock), 0, t->stream, t->wt, 2, lower, upper,
                       t->dump_have_ovf, 0ull, 0ull, 1ull, 1ull, (unsigned long long*)nullptr, d_cnt);
  } else
  hipLaunchKernelGGL(tile_count_kernel, dim3(grid_for(t, nt)), dim3(kBlock), 0, t->stream, t->dt, lower, upper,
                     t->dump_have_ovf, nt, d_cnt);
  std::vector<uint32_t> h(nt);
  hipError_t e = hipMemcpyAsync(h.data(), d_cnt, nt * sizeof(uint32_t), hipMemcpyDeviceToHost, t->stream);
  if(e == hipSuccess) e = hipStreamSynchronize(t->stream);
  hipFree(d_cnt);
  if(e != hipSuccess) return fail(JFGPU_E_HIP, hipGetErrorString(e));
  t->dump_prefix.assign(nt + 1, 0);
  for(uint64_t i = 0; i < nt; ++i) t->dump_prefix[i + 1] = t->dump_prefix[i] + h[i];
  t->dump_lower = lower; t->dump_upper = upper; t->dump_tile_cursor = 0; t->dump_open = true;
  if(n_records) *n_records = t->dump_prefix[nt];
  if(record_bytes) *record_bytes = (t->g.key_bits + 7) / 8 + t->out_counter_len;
  return JFGPU_OK;
}

int jfgpu_dump_next(jfgpu_table* t, void* out, uint64_t capacity_records, uint64_t* n_read) {
  int rc = use(t); if(rc) return rc;
  if(!t->dump_open) return fail(JFGPU_E_INVALID, "jfgpu_dump_begin not called");
  if(!out || !n_read) return fail(JFGPU_E_INVALID, "null argument");
  *n_read = 0;
  const uint64_t nt = n_tiles_of(t);
  const uint64_t tsz = 1ull << t->g.tile_bits;
  if(capacity_records < tsz) return fail(JFGPU_E_INVALID, "dump buffer must hold at least one tile (" + std::to_string(tsz) + " records)");
  uint64_t t0 = t->dump_tile_cursor;
  while(t0 < nt && t->dump_prefix[t0 + 1] == t->dump_prefix[t0]) ++t0;  // skip empty tiles
  if(t0 >= nt) { t->dump_tile_cursor = nt; return JFGPU_OK; }
  // largest t1 with prefix[t1] - prefix[t0] <= capacity
  const uint64_t limit = t->dump_prefix[t0] + capacity_records;
  uint64_t t1 = std::upper_bound(t->dump_prefix.begin() + t0, t->dump_prefix.end(), limit) - t->dump_prefix.begin() - 1;
  t1 = std::min(t1, nt);
  const uint64_t ntile = t1 - t0, nrec = t->dump_prefix[t1] - t->dump_prefix[t0];
  const uint32_t key_bytes = (t->g.key_bits + 7) / 8, rec = key_bytes + t->out_counter_len;
  if(t->dump_cap_records < nrec) {
    if(t->d_dump) hipFree(t->d_dump);
    t->d_dump = nullptr; t->dump_cap_records = 0;
    HIP_TRY(hipMalloc((void**)&t->d_dump, std::max<uint64_t>(nrec, capacity_records) * rec));
    t->dump_cap_records = std::max<uint64_t>(nrec, capacity_records);
  }
  if(t->tile_off_cap < ntile) {
    if(t->d_tile_off) hipFree(t->d_tile_off);
    t->d_tile_off = nullptr; t->tile_off_cap = 0;
    HIP_TRY(hipMalloc((void**)&t->d_tile_off, ntile * sizeof(uint64_t)));
    t->tile_off_cap = ntile;
  }
  std::vector<uint64_t> offs(ntile);
  for(uint64_t i = 0; i < ntile; ++i) offs[i] = t->dump_prefix[t0 + i] - t->dump_prefix[t0];
  HIP_TRY(hipMemcpyAsync(t->d_tile_off, offs.data(), ntile * sizeof(uint64_t), hipMemcpyHostToDevice, t->stream));
  if(t->nword) {
    const size_t nl = ((size_t)32 << t->g.tile_bits) + ((size_t)2 << t->g.tile_bits);
    hipLaunchKernelGGL(dump_tiles_nword_kernel, dim3(grid_for(t, ntile)), dim3(kBlock), nl, t->stream, t->nt, t->dump_lower, t->dump_upper,
                       t->dump_have_ovf, t0, ntile, (const uint64_t*)t->d_tile_off, t->d_dump, key_bytes, t->out_counter_len);
  } else if(t->wide) {
    const size_t wl = ((size_t)16 << t->g.tile_bits) + ((size_t)2 << t->g.tile_bits);
    hipLaunchKernelGGL(dump_tiles_wide_kernel, dim3(grid_for(t, ntile)), dim3(kBlock), wl, t->stream, t->wt, t->dump_lower, t->dump_upper,
                       t->dump_have_ovf, t0, ntile, (const uint64_t*)t->d_tile_off, t->d_dump, key_bytes, t->out_counter_len);
  } else {
  const size_t lds = ((size_t)8 << t->g.tile_bits) + ((size_t)2 << t->g.tile_bits) + (size_t)t->g.nbytes * 2048;
  hipLaunchKernelGGL(dump_tiles_kernel, dim3(grid_for(t, ntile)), dim3(kBlock), lds, t->stream, t->dt, t->dump_lower,
                     t->dump_upper, t->dump_have_ovf, t0, ntile, (const uint64_t*)t->d_tile_off, t->d_dump, key_bytes,
                     t->out_counter_len);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, t->d_dump, nrec * rec, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  t->dump_tile_cursor = t1;
  *n_read = nrec;
  return JFGPU_OK;
}

int jfgpu_dump_end(jfgpu_table* t) {
  int rc = use(t); if(rc) return rc;
  t->dump_open = false;
  t->dump_prefix.clear(); t->dump_prefix.shrink_to_fit();
  if(t->d_dump) { hipFree(t->d_dump); t->d_dump = nullptr; t->dump_cap_records = 0; }
  if(t->d_tile_off) { hipFree(t->d_tile_off); t->d_tile_off = nullptr; t->tile_off_cap = 0; }
  return JFGPU_OK;
}

int jfgpu_set_growth(jfgpu_table* t, int on) {
  int rc = use(t); if(rc) return rc;
  t->grow_on = on != 0;
  return JFGPU_OK;
}

int jfgpu_reserve(jfgpu_table* t, uint64_t input_bytes) {
  int rc = use(t); if(rc) return rc;
  if(!t->part_ok || t->mode == MODE_DIRECT) return JFGPU_OK;
  rc = part_flush(t); if(rc) return rc;
  t->reserved_input = input_bytes;
  const uint32_t nb1 = 1u << t->pg.b1, nb2 = 1u << t->pg.b2;
  // pending items (upper bound: one per input byte) + the P2 output of the same size + offsets
  // (single-pass P1 batches are regions with head-room: slack + one stranded reservation per block and bucket)
  const size_t pend = align_up(input_bytes * item_size(t), 256);
  const size_t strand = (size_t)nb1 * (2 * (size_t)t->n_cu) * kGran * item_size(t);              // per batch
  const size_t headroom = t->pg.b2 ? (size_t)(pend * ((t->tun.p1_slack > 0 ? t->tun.p1_slack : 0.0) + 0.05)) + 2 * strand : 0;
  const size_t need = 2 * pend + headroom + (n_tiles_of(t) + 1 + nb1) * sizeof(uint64_t) +
                      (size_t)kMaxSeg * (align_up((2 * nb1 + 1) * sizeof(uint64_t), 256) + align_up(nb1 * 16, 256) + 1280) + ((size_t)1 << 20);
  size_t want = need;
  if(t->item128) {      // 16-byte items: what a whole input would need may exceed the device; the arena is then flushed more than once
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const size_t avail = free_b + t->ws_cap > ((size_t)8 << 30) ? free_b + t->ws_cap - ((size_t)8 << 30) : 0;
    want = std::min(want, avail);
  }
  rc = ws_grow(t, want);
  if(rc < 0) return fail(JFGPU_E_ALLOC, "not enough device memory to reserve the partition workspace");
  if(rc) return rc;
  // touch every page once now: first-touch of fresh device pages costs ~40% on the first pass over them
  HIP_TRY(hipMemsetAsync(t->ws, 0, t->ws_cap, t->stream));
  if(!t->d_M1) { t->g1 = 2 * t->n_cu; HIP_TRY(hipMalloc((void**)&t->d_M1, (size_t)t->g1 * kMaxBuckets * sizeof(uint32_t))); }
  if(t->pg.b2 && !t->d_M2) HIP_TRY(hipMalloc((void**)&t->d_M2, (size_t)nb1 * 32 * nb2 * sizeof(uint32_t)));
  return JFGPU_OK;
}

int jfgpu_table_bytes(uint32_t k, uint64_t size, uint64_t* slots, uint64_t* bytes) {
  if(k < 1 || k > 128) return fail(JFGPU_E_UNSUPPORTED, "mer length must be in [1, 128]");
  uint32_t lsize = 0;
  while(lsize < 63 && (1ull << lsize) < size) ++lsize;
  const bool nword = k > 64, wide = k > 32 && !nword;
  if(nword) { lsize = std::max(lsize, nword_min_lsize(k)); lsize = std::min<uint32_t>(lsize, 48); }
  else if(wide) { lsize = std::max(lsize, wide_min_lsize(k)); lsize = std::min<uint32_t>(lsize, 48); }
  else { lsize = std::max(lsize, geom_min_lsize(k, 0)); lsize = std::max<uint32_t>(lsize, 1); lsize = std::min<uint32_t>(lsize, 2 * k); }
  const uint64_t n = 1ull << lsize;
  uint64_t ovf = std::max<uint64_t>(kMinOvf, std::min<uint64_t>(n / 256, 1ull << 26));
  { uint64_t c = 1; while(c < ovf) c <<= 1; ovf = c; }
  if(slots) *slots = n;
  TableGeom gg;
  const bool s32 = !nword && !wide && geom_init(gg, k, lsize, 0, 0, 1, !Tuning::from_env().slot64) && gg.slot32;
  if(bytes) *bytes = n * (nword ? 32 : wide ? 16 : s32 ? 4 : 8) + ovf * 16 + (n >> std::min<uint32_t>(lsize, kMaxTileBits)) + (size_t)(2 * k + 7) / 8 * 256 * 8 * 2;
  return JFGPU_OK;
}

int jfgpu_set_spill(jfgpu_table* t, int (*fn)(void*), void* user) {
  int rc = use(t); if(rc) return rc;
  t->spill_fn = fn; t->spill_user = user;
  return JFGPU_OK;
}

int jfgpu_reference_matrix(uint32_t lsize, uint32_t key_len, uint64_t* columns) {
  if(!columns || lsize < 1 || lsize > 64 || key_len < 2 || key_len > 256) return fail(JFGPU_E_INVALID, "bad matrix dimensions");
  GlibcRandom rng;
  const Gf2Matrix m = lsize >= key_len ? gf2_identity(lsize, key_len) : gf2_reference_matrix(lsize, key_len, rng);
  for(uint32_t i = 0; i < key_len; ++i) columns[i] = m.columns[i];
  return JFGPU_OK;
}

int jfgpu_set_operation(jfgpu_table* t, int op) {
  int rc = use(t); if(rc) return rc;
  if(op < 0 || op > 2) return fail(JFGPU_E_INVALID, "operation must be 0 (count), 1 (prime) or 2 (update)");
  if(op != t->operation) { rc = part_flush(t); if(rc) return rc; }     // pending adds belong to the old operation
  t->operation = op;
  return JFGPU_OK;
}

int jfgpu_set_mode(jfgpu_table* t, int mode) {
  int rc = use(t); if(rc) return rc;
  if(mode < 0 || mode > 2) return fail(JFGPU_E_INVALID, "mode must be 0 (auto), 1 (direct) or 2 (partitioned)");
  if(mode == MODE_PARTITIONED && !t->part_ok) return fail(JFGPU_E_UNSUPPORTED, "table geometry has no partitioned path");
  rc = part_flush(t); if(rc) return rc;
  t->mode = mode;
  return JFGPU_OK;
}

int jfgpu_profile_enable(jfgpu_table* t, int on) {
  int rc = use(t); if(rc) return rc;
  t->prof_on = on != 0;
  return JFGPU_OK;
}

int jfgpu_profile_get(jfgpu_table* t, int which, double* ms, uint64_t* launches, uint64_t* units) {
  int rc = use(t); if(rc) return rc;
  if(which < 0 || which >= kNumProf) return fail(JFGPU_E_INVALID, "bad profile slot");
  HIP_TRY(hipStreamSynchronize(t->stream));
  prof_collect(t);
  if(ms) *ms = t->prof_ms[which];
  if(launches) *launches = t->prof_launches[which];
  if(units) *units = t->prof_units[which];
  return JFGPU_OK;
}

int jfgpu_get_counters(jfgpu_table* t, uint64_t* out, uint32_t n) {
  int rc = use(t); if(rc) return rc;
  if(!out) return fail(JFGPU_E_INVALID, "null argument");
  uint64_t c[CTR_COUNT];
  rc = read_counters(t, c); if(rc) return rc;
  const uint64_t v[JFGPU_N_COUNTERS] = {c[CTR_FULL], c[CTR_MERS], c[CTR_OVF_FULL], c[CTR_OVF_USED], c[CTR_MISROUTED], c[CTR_DIRECT],
                                        c[CTR_T_ITEMS], c[CTR_T_QUEUED], t->flushes_plain, t->flushes_heavy,
                                        t->n_p2_roles, t->n_p2_ring, t->n_p2_sort, t->n_p2_exact, t->n_p1_ring, t->n_p1_other};
  for(uint32_t i = 0; i < n; ++i) out[i] = i < JFGPU_N_COUNTERS ? v[i] : 0;
  return JFGPU_OK;
}

int jfgpu_profile_reset(jfgpu_table* t) {
  int rc = use(t); if(rc) return rc;
  HIP_TRY(hipStreamSynchronize(t->stream));
  prof_collect(t);
  for(int i = 0; i < kNumProf; ++i) { t->prof_ms[i] = 0; t->prof_launches[i] = 0; t->prof_units[i] = 0; }
  t->prof_log.clear();
  return JFGPU_OK;
}

int jfgpu_profile_spans(jfgpu_table* t, int* which, double* ms, size_t cap, size_t* n) {
  int rc = use(t); if(rc) return rc;
  if(!n) return fail(JFGPU_E_INVALID, "null argument");
  HIP_TRY(hipStreamSynchronize(t->stream));
  prof_collect(t);
  *n = t->prof_log.size();
  for(size_t i = 0; i < t->prof_log.size() && i < cap; ++i) { if(which) which[i] = t->prof_log[i].first; if(ms) ms[i] = t->prof_log[i].second; }
  return JFGPU_OK;
}

int jfgpu_gen_reads_dev(jfgpu_table* t, char* d_out, uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t seed) {
  int rc = use(t); if(rc) return rc;
  if(!n_reads) return JFGPU_OK;
  if(!d_out) return fail(JFGPU_E_INVALID, "null buffer");
  if(read_len < 1 || read_len > 2048) return fail(JFGPU_E_INVALID, "read_len must be in [1, 2048]");
  if(((uintptr_t)d_out & 15) != 0) return fail(JFGPU_E_INVALID, "gen_reads: output must be 16-byte aligned");
  const uint64_t vecs = (n_reads * ((uint64_t)read_len + 1) + 15) / 16;
  hipLaunchKernelGGL(gen_reads_kernel, dim3(grid_for(t, (vecs + kBlock - 1) / kBlock)), dim3(kBlock), 0, t->stream,
                     (uint8_t*)d_out, first_read, n_reads, read_len, seed);
  HIP_TRY(hipGetLastError());
  return JFGPU_OK;
}

int jfgpu_gen_genome_reads_dev(jfgpu_table* t, char* d_out, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                               uint64_t genome_len, double substitution_rate, uint64_t seed) {
  int rc = use(t); if(rc) return rc;
  if(!n_reads) return JFGPU_OK;
  if(!d_out) return fail(JFGPU_E_INVALID, "null buffer");
  if(read_len < 1 || read_len > 2048 || genome_len < read_len) return fail(JFGPU_E_INVALID, "read_len must be in [1, 2048] and <= genome_len");
  if(substitution_rate < 0 || substitution_rate > 0.5) return fail(JFGPU_E_INVALID, "substitution rate must be in [0, 0.5]");
  if(((uintptr_t)d_out & 15) != 0) return fail(JFGPU_E_INVALID, "gen_genome_reads: output must be 16-byte aligned");
  const uint64_t vecs = (n_reads * ((uint64_t)read_len + 1) + 15) / 16;
  hipLaunchKernelGGL(gen_genome_reads_kernel, dim3(grid_for(t, (vecs + kBlock - 1) / kBlock)), dim3(kBlock), 0, t->stream,
                     (uint8_t*)d_out, first_read, n_reads, read_len, genome_len, (uint32_t)lrint(substitution_rate * 65536.0), seed);
  HIP_TRY(hipGetLastError());
  return JFGPU_OK;
}

int jfgpu_gups(jfgpu_table* t, uint64_t n_updates, int mode, double* ups) {
  int rc = use(t); if(rc) return rc;
  if(!ups) return fail(JFGPU_E_INVALID, "null out");
  unsigned long long* d_sink = nullptr;
  HIP_TRY(hipMalloc((void**)&d_sink, sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(d_sink, 0, sizeof(unsigned long long), t->stream));
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
  const uint64_t mask = (((1ull << t->g.lsize_l) * slot_bytes_of(t)) >> 3) - 1;    // 64-bit words of the table's allocation
  t->pristine = false;
  HIP_TRY(hipMemsetAsync(t->dt.dirty, 1, (size_t)1 << (t->g.lsize_l - t->g.tile_bits), t->stream));
  const int grid = grid_for(t, (n_updates + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(gups_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->dt.slots, mask, std::min<uint64_t>(n_updates, 1u << 20), mode, 1ull, d_sink);  // warm-up
  HIP_TRY(hipEventRecord(a, t->stream));
  hipLaunchKernelGGL(gups_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->dt.slots, mask, n_updates, mode, 42ull, d_sink);
  HIP_TRY(hipEventRecord(b, t->stream));
  HIP_TRY(hipEventSynchronize(b));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, a, b));
  hipEventDestroy(a); hipEventDestroy(b); hipFree(d_sink);
  *ups = ms > 0 ? (double)n_updates / (ms * 1e-3) : 0;
  return JFGPU_OK;
}

int jfgpu_malloc_host(size_t bytes, void** out) {       // pinned: device copies to / from it run at PCIe speed
  if(!out) return fail(JFGPU_E_INVALID, "null out");
  *out = nullptr;
  HIP_TRY(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
  return JFGPU_OK;
}
int jfgpu_free_host(void* p) {
  if(p) HIP_TRY(hipHostFree(p));
  return JFGPU_OK;
}
int jfgpu_malloc_dev(jfgpu_table* t, size_t bytes, void** out) {
  int rc = use(t); if(rc) return rc;
  if(!out) return fail(JFGPU_E_INVALID, "null out");
  HIP_TRY(hipMalloc(out, bytes ? bytes : 16));
  return JFGPU_OK;
}
int jfgpu_free_dev(jfgpu_table* t, void* p) {
  int rc = use(t); if(rc) return rc;
  HIP_TRY(hipStreamSynchronize(t->stream));
  HIP_TRY(hipFree(p));
  return JFGPU_OK;
}
int jfgpu_memcpy_h2d(jfgpu_table* t, void* d_dst, const void* src, size_t bytes) {
  int rc = use(t); if(rc) return rc;
  HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  return JFGPU_OK;
}
int jfgpu_memcpy_d2h(jfgpu_table* t, void* dst, const void* d_src, size_t bytes) {
  int rc = use(t); if(rc) return rc;
  HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  return JFGPU_OK;
}

}  // extern "C"

#include "abi_bloom.inl"
#include "abi_parser.inl"
#include "abi_comm.inl"
