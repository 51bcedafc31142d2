// jellyfish_amd/csrc/tuning.hpp -- every JFGPU_* environment switch of the ENGINE LIBRARY (libjfgpu.so), read in ONE place
// (the CLI and the facade headers above the C ABI read their own few where they use them: INTEGRATION.md lists both).
//
// The switches are A/B and test knobs (nothing a user of the reference's CLI needs): which insert path, the head-room of
// the partition regions, forcing rare code paths so that the parity tests reach them.  An object (table, Bloom counter,
// communicator) takes a snapshot when it is created -- Tuning::from_env() -- and nothing reads the environment after
// that, so two objects of one process can be created under different settings (tests) and a setting cannot change under
// a running job.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>

namespace jfgpu {

struct Tuning {
  // ---- tables (jfgpu_create)
  int mode = 0;                  // JFGPU_MODE=direct|partitioned            0: not set, 1 direct, 2 partitioned
  bool slot64 = false;           // JFGPU_SLOT64=1         never use 32-bit slots (A/B against the 4-byte layout)
  int p1_single = -1;            // JFGPU_P1_SINGLE        single-pass P1: -1 auto, 0 never, 1 whenever the geometry allows
  double p1_slack = 0.03;        // JFGPU_P1_SLACK         head-room of a P1 bucket region over the mean (negative: forces the exhausted path)
  int flush_groups = 1;          // JFGPU_FLUSH_GROUPS     P2 / tile-insert pipeline depth of a flush
  bool tile_pair = true;         // JFGPU_TILE_PAIR=0      32-bit slots: single tiles instead of pairs (A/B)
  int p2_single = 1;             // JFGPU_P2_SINGLE        single-pass P2: 0 exact count + scatter, 1 when regions would be mostly items, 2 always
  uint32_t p2_cap = 0;           // JFGPU_P2_CAP           items per P2 region (tests: forces region overflow)
  double p2_slack = -1;          // JFGPU_P2_SLACK         head-room of a P2 region; < 0: 30 % (4- and 8-byte items), 8 % (16-byte)
  int tile_adapt = 1;            // JFGPU_TILE_ADAPT       tile kernel instantiation: 1 sampled per flush, 0 always plain, 2 always HEAVY
  uint32_t flush_share = 0;      // JFGPU_FLUSH_SHARE      force a flush into this many bucket groups sharing one P2 buffer (tests)
  bool flush_trace = false;      // JFGPU_FLUSH_TRACE      one stderr line per flush
  int p2_ring = 1;               // JFGPU_P2_RING          single-pass P2 of 4-byte items through per-destination rings (0: the sort-based kernel, 3: never the loader / storer kernel; A/B)
  bool wide_pipe = true;        // JFGPU_WIDE_PIPE=0      two-word keys: the round-2 tile kernel instead of the pipelined one (A/B)
  int p2_depth = 3;              // JFGPU_P2_DEPTH         rounds of items the loader waves of p2_ring_roles_kernel keep in flight (1, 2 or 3: 19.5 / 19.25 / 19.1 ms on the metric's job; A/B)
  int bloom_cache = -1;          // JFGPU_BLOOM_CACHE      count --bc: remember admitted k-mers (-1: when the first batches admit > 15 % of their windows, 0 never, 1 always)
  int matrix = 0;                // JFGPU_MATRIX=xs|reference   the matrix family of tables created with matrix_kind 0 (0: not set -> the reference's)
  uint32_t bloom_cache_log2 = 28;// JFGPU_BLOOM_CACHE_LOG2 two-way sets of that cache (2^28 sets = 4 GB; tests: tiny caches evict all the time)
  // ---- Bloom counters (jfgpu_bloom_create)
  int bloom_mode = 0;            // JFGPU_BLOOM_MODE=direct|partitioned      0: not set
  int bloom_p1_two = 1;          // JFGPU_BLOOM_P1_TWO     P1b as two workgroups per CU (rounds of 5 cells, nibble tables); 0: one, rounds of 10 (A/B)
  int bloom_p1_ring = 1;         // JFGPU_BLOOM_P1_RING    P1b through rings of 256 bytes where the filter gives it >= 200 buckets of at most 512 (0: the sort-based kernels; A/B)
  // ---- communicators (jfgpu_comm_create*)
  bool comm_trace = false;       // JFGPU_COMM_TRACE       every rank reports where it is in a step
  bool comm_ipc = false;         // JFGPU_COMM_TRANSPORT=ipc   rank processes exchange through hipIpc* copies instead of RCCL
  int comm_ipc_keep = 1;         // JFGPU_IPC_KEEP         ipc transport: 1 = re-allocated send buffers and the peers' mappings of them live until the communicator goes (0: closed / freed as soon as replaced; A/B)
  uint64_t comm_max_msg = 0;     // JFGPU_COMM_MAX_MSG     keys per message round (0: default)
  int comm_self_rccl = -1;       // JFGPU_COMM_SELF_RCCL   world 1: send the rank's own share through RCCL too (-1: default)
  int comm_items = -1;           // JFGPU_COMM_ITEMS       item path: -1 default, 0 off, 1 on, 2 forced
  int comm_strag = -1;           // JFGPU_COMM_STRAG       straggler list capacity (tests; -1: default)
  double comm_slack = 0.03;      // JFGPU_COMM_SLACK       item path: head-room of a routed region over the mean, applied to the k-mers-per-byte estimate and to the mean (round 6: 0.03; 0.10 before)
  int comm_gbits = 10;           // JFGPU_COMM_GBITS       bits of (owner, coarse bucket) the sender routes by (tests: fewer, so that small shards get the wide receive split of full-size ones)
  uint32_t comm_split_cap = 0;   // JFGPU_COMM_SPLIT_CAP   items per region of the receive split (tests: forces region overflow there)
  int comm_split = 1;            // JFGPU_COMM_SPLIT=0     the receive split by round 4's sort-based kernel instead of the wave-per-stream one (A/B)

  static Tuning from_env() {
    Tuning u;
    auto str = [](const char* name) -> const char* { const char* e = getenv(name); return e && *e ? e : nullptr; };
    if(const char* e = str("JFGPU_MODE")) u.mode = !strcmp(e, "direct") ? 1 : !strcmp(e, "partitioned") ? 2 : 0;
    if(const char* e = str("JFGPU_SLOT64")) u.slot64 = atoi(e) != 0;
    if(const char* e = str("JFGPU_MATRIX")) u.matrix = (!strcmp(e, "xs") || !strcmp(e, "xorshift")) ? 1 : (!strcmp(e, "reference") || !strcmp(e, "ref")) ? 2 : 0;
    if(const char* e = str("JFGPU_P1_SINGLE")) u.p1_single = atoi(e) ? 1 : 0;
    if(const char* e = str("JFGPU_P1_SLACK")) u.p1_slack = atof(e);
    if(const char* e = str("JFGPU_FLUSH_GROUPS")) u.flush_groups = std::max(1, atoi(e));
    if(const char* e = str("JFGPU_TILE_PAIR")) u.tile_pair = atoi(e) != 0;
    if(const char* e = str("JFGPU_P2_SINGLE")) u.p2_single = atoi(e);
    if(const char* e = str("JFGPU_P2_CAP")) u.p2_cap = (uint32_t)atoi(e);
    if(const char* e = str("JFGPU_P2_SLACK")) u.p2_slack = atof(e);
    if(const char* e = str("JFGPU_TILE_ADAPT")) u.tile_adapt = atoi(e);
    if(const char* e = str("JFGPU_FLUSH_SHARE")) u.flush_share = (uint32_t)atoi(e);
    u.flush_trace = str("JFGPU_FLUSH_TRACE") != nullptr;
    if(const char* e = str("JFGPU_P2_RING")) u.p2_ring = atoi(e);
    if(const char* e = str("JFGPU_WIDE_PIPE")) u.wide_pipe = atoi(e) != 0;
    if(const char* e = str("JFGPU_P2_DEPTH")) u.p2_depth = std::min(3, std::max(1, atoi(e)));
    if(const char* e = str("JFGPU_BLOOM_CACHE")) u.bloom_cache = atoi(e);
    if(const char* e = str("JFGPU_BLOOM_CACHE_LOG2")) u.bloom_cache_log2 = (uint32_t)std::min(30, std::max(2, atoi(e)));
    if(const char* e = str("JFGPU_BLOOM_P1_TWO")) u.bloom_p1_two = atoi(e);
    if(const char* e = str("JFGPU_BLOOM_P1_RING")) u.bloom_p1_ring = atoi(e);
    if(const char* e = str("JFGPU_BLOOM_MODE")) u.bloom_mode = !strcmp(e, "direct") ? 1 : !strcmp(e, "partitioned") ? 2 : 0;
    u.comm_trace = str("JFGPU_COMM_TRACE") != nullptr;
    if(const char* e = str("JFGPU_COMM_TRANSPORT")) u.comm_ipc = !strcmp(e, "ipc");
    if(const char* e = str("JFGPU_IPC_KEEP")) u.comm_ipc_keep = atoi(e);
    if(const char* e = str("JFGPU_COMM_MAX_MSG")) u.comm_max_msg = std::max<uint64_t>(1, strtoull(e, 0, 10));
    if(const char* e = str("JFGPU_COMM_SELF_RCCL")) u.comm_self_rccl = atoi(e) != 0;
    if(const char* e = str("JFGPU_COMM_ITEMS")) u.comm_items = atoi(e);
    if(const char* e = str("JFGPU_COMM_STRAG")) u.comm_strag = std::max(1, atoi(e));
    if(const char* e = str("JFGPU_COMM_SPLIT")) u.comm_split = atoi(e);
    if(const char* e = str("JFGPU_COMM_SPLIT_CAP")) u.comm_split_cap = (uint32_t)std::max(64, atoi(e));
    if(const char* e = str("JFGPU_COMM_SLACK")) u.comm_slack = std::min(1.0, std::max(0.0, atof(e)));
    if(const char* e = str("JFGPU_COMM_GBITS")) u.comm_gbits = std::min(10, std::max(1, atoi(e)));
    return u;
  }
};

}  // namespace jfgpu
