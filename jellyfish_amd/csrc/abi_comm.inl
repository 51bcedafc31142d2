// jellyfish_amd/csrc/abi_comm.inl -- multi-GPU exchange under the C ABI (jfgpu_comm_*), included by jfgpu.hip.
//
// the global table has 2^lsize_g positions, rank r owns the positions whose top shard_bits bits are r,
// every rank routes the k-mers of its own input to their owners and inserts what it receives; no other collective.
// One process per GPU, RCCL (ncclSend / ncclRecv inside one group per round) over xGMI.  The step is software-pipelined
// by one: while the keys of step i travel on the exchange stream, the device routes step i+1 and partitions (P1 from
// keys) what arrived for step i-1; stream order is carried by events, the host only waits for the per-destination
// counts of its own routing pass (they size the messages).
//
// A second transport, "local", keeps all ranks' shards in ONE process on one device and moves the messages with
// device copies: same routing, bookkeeping, rounds and insert code, no RCCL.  It exists so that the sharded path is
// testable on a single GPU (and under tests/host/hip_emu) at world sizes 2 and 4.
//
// A third transport, "ipc", has one process per rank like RCCL but moves the messages with copies between the processes'
// device allocations (hipIpcGetMemHandle / hipIpcOpenMemHandle) and keeps the small host-level agreements (counts,
// barriers, allreduce / allgather) in a POSIX shared-memory block.  Every step ends on a host barrier, so nothing
// overlaps: it exists so that the multi-PROCESS code around the exchange -- `count --gpus N`: rank start-up, rendezvous,
// file parts, the collectives, the sharded writer, a failing rank -- runs and is tested on a box with a single GPU
// (all ranks on device 0).  Chosen with JFGPU_COMM_TRANSPORT=ipc when the id is made (jfgpu_comm_unique_id).
#if !defined(JFGPU_EMU)
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <thread>
#endif

namespace { struct IpcShared; }

struct jfgpu_comm {
  int world = 1, rank = 0, device = 0;
  Tuning tun = Tuning::from_env();               // the JFGPU_* switches as they were at creation (tuning.hpp)
  bool local = false;
#if !defined(JFGPU_EMU)
  ncclComm_t nccl = nullptr;
#endif
  hipStream_t xstream = nullptr;                 // exchange stream
  uint64_t* d_coll = nullptr;                    // [512] staging of the small host-level collectives
  uint64_t max_msg_keys = (uint64_t)1 << 27;     // 1 GiB per peer per round (a 6.9 GB self-message was dropped by RCCL 2.26)
  bool self_rccl = false;                        // JFGPU_COMM_SELF_RCCL=1: a rank's own share travels through ncclSend/ncclRecv as
                                                 // well (default: a device copy) -- lets a single-GPU box exercise every RCCL call
  struct Rank {
    jfgpu_table* t = nullptr;
    uint64_t* send[2] = {nullptr, nullptr}; size_t send_cap[2] = {0, 0};
    uint64_t* recv[2] = {nullptr, nullptr}; size_t recv_cap[2] = {0, 0};
    unsigned long long* d_cnt = nullptr;         // [world] device counters of the routing pass
    uint64_t* d_xc = nullptr;                    // [2 * world] staging of the counts exchange (send | recv)
    std::vector<uint64_t> scount[2], soff[2], rcount[2], roff[2];
    hipEvent_t routed[2] = {nullptr, nullptr}, exchanged[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    int turn = 0; bool inflight = false;
    uint64_t sent = 0, received = 0;
    // item path (see comm_route_items): this step's region capacity (0: the step went as 8-byte keys), the P1 cursors,
    // k-mers per input byte seen so far (sizes the regions)
    uint32_t icap[2] = {0, 0};
    unsigned int* d_gcur = nullptr;              // gcur[2 * 1024] (u32) then tot[1024] (u64)
    unsigned long long* d_claimed = nullptr;     // k-mers the senders said they sent here (item path), summed on the device
    unsigned long long* d_arrived = nullptr;     // [2]: k-mers the receive side actually found in what arrived (regions split + direct inserts + stragglers kept); scratch word
    double ipb = 0;
    uint64_t strag_seen = 0;
    uint64_t* d_route = nullptr;                 // [2] items stored / stragglers of the routing pass in flight
    uint64_t* h_route = nullptr;                 // ... pinned copy, read after route_done
    hipEvent_t route_done = nullptr;
    size_t route_n = 0;                          // its input bytes
    const uint32_t* self_items[2] = {nullptr, nullptr};   // item path, RCCL transport: the rank's own share is read where the routing left it (send[turn]), not copied
    // exchange timing (jfgpu_comm_exchange_times): a pair of timing events per turn, open while its exchange may be running
    hipEvent_t x_begin[2] = {nullptr, nullptr}, x_end[2] = {nullptr, nullptr};
    bool x_open[2] = {false, false}; uint64_t x_seq[2] = {0, 0}, x_bytes[2] = {0, 0};
  };
  uint64_t x_next = 0;                           // exchanges started so far
  std::vector<std::pair<double, uint64_t>> x_log;   // (device ms, bytes this rank sent over the wires) per exchange, in order
#if !defined(JFGPU_EMU)
  // "ipc" transport (see the head of this file)
  IpcShared* shm = nullptr; std::string shm_name;
  struct PeerMap { void* ptr = nullptr; uint64_t epoch = 0; };
  std::vector<PeerMap> peer_send[2];             // peers' send buffers as mapped here, per turn
  std::vector<void*> stale_maps;                 // mappings of send buffers the peers have replaced since (closed with the communicator)
  uint64_t send_epoch[2] = {0, 0}; void* send_exported[2] = {nullptr, nullptr}; size_t send_exported_cap[2] = {0, 0};
  uint32_t barrier_gen = 0;
#endif
  bool ipc = false;
  // ipc transport: send buffers given up while a peer may still have them mapped (freed after the next exchange of their turn,
  // by when every peer has let go of its mapping: see comm_reserve_send)
  std::vector<void*> send_retired[2];
  bool items_on = true; int items_mode = 1;      // JFGPU_COMM_ITEMS: 0 always send 8-byte keys, 1 items when the step is large enough, 2 items always
  uint32_t strag_cap = 1u << 16;                 // stragglers per rank and step (JFGPU_COMM_STRAG)
  std::vector<Rank> ranks;                       // RCCL transport: one; local transport: `world`
};

namespace {

// JFGPU_COMM_TRACE=1: every rank says on stderr where it is in the exchange (a stuck run shows who waits for whom).
#define IPC_TRACE(c, ...) do { if((c)->tun.comm_trace) { fprintf(stderr, "[comm rank %d] ", (c)->rank); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while(0)

int comm_reserve(uint64_t*& buf, size_t& cap, size_t need, hipStream_t s1, hipStream_t s2);
int comm_reserve_send(jfgpu_comm* c, jfgpu_comm::Rank& R, int turn, size_t need, hipStream_t s1);

// the communicator's share of the JFGPU_* switches (tuning.hpp), applied once at creation
void comm_apply_tuning(jfgpu_comm* c) {
  if(c->tun.comm_max_msg) c->max_msg_keys = c->tun.comm_max_msg;
  if(c->tun.comm_items >= 0) { c->items_mode = c->tun.comm_items; c->items_on = c->items_mode != 0; }
  if(c->tun.comm_strag > 0) c->strag_cap = (uint32_t)c->tun.comm_strag;
}

int comm_init_rank(jfgpu_comm* c, jfgpu_comm::Rank& R) {
  for(int i = 0; i < 2; ++i) {
    HIP_TRY(hipEventCreateWithFlags(&R.routed[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&R.exchanged[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&R.consumed[i], hipEventDisableTiming));
    R.scount[i].assign(c->world, 0); R.soff[i].assign(c->world + 1, 0); R.rcount[i].assign(c->world, 0); R.roff[i].assign(c->world + 1, 0);
    HIP_TRY(hipEventCreate(&R.x_begin[i]));
    HIP_TRY(hipEventCreate(&R.x_end[i]));
  }
  HIP_TRY(hipMalloc((void**)&R.d_cnt, sizeof(unsigned long long) * c->world));
  HIP_TRY(hipMalloc((void**)&R.d_xc, sizeof(uint64_t) * 2 * c->world));
  HIP_TRY(hipMalloc((void**)&R.d_gcur, 1024 * 16));
  HIP_TRY(hipMalloc((void**)&R.d_claimed, 8));
  HIP_TRY(hipMemset(R.d_claimed, 0, 8));
  HIP_TRY(hipMalloc((void**)&R.d_arrived, 16));
  HIP_TRY(hipMemset(R.d_arrived, 0, 16));
  HIP_TRY(hipMalloc((void**)&R.d_route, 2 * sizeof(uint64_t)));
  HIP_TRY(hipHostMalloc((void**)&R.h_route, 2 * sizeof(uint64_t), hipHostMallocDefault));
  HIP_TRY(hipEventCreateWithFlags(&R.route_done, hipEventDisableTiming));
  return JFGPU_OK;
}

// ---- the item path: 4 bytes per k-mer, already grouped for the receiver ------------------------------------------
// Sender: p1_ring_kernel<.., RouteListDirect> = the single-pass P1 over the global table (1024 buckets = owner x coarse bucket), so
// an owner's share is one contiguous run of regions -- equal-sized messages, no counts to exchange.  Receiver: the regions
// of its coarse buckets from all W senders go through one more split (fan-out W, p2_granule_kernel) into the regions of
// its own 1024 P1 buckets: a pending batch like any other, applied by the next flush.  Region capacity is agreed per
// step (the maximum of what the ranks want); what cannot travel this way (a region overflows on skewed input, the item
// that looks like a hole) goes on a short list every rank receives.  A step falls back to 8-byte keys on all ranks when
// any rank says so: geometry (2k - 10 > 32 bits, fewer than 2^23 slots per shard), steps too small for fixed regions,
// or more stragglers than the list holds.
struct ItemLayout {
  uint32_t cap, nbg, nbc, gbits, cbits, split_bits, S;
  size_t items_bytes, offs_at, claims_at, strag_at, send_bytes;   // send buffer: items[nbg * cap] | offs[2 * nbg] | claims[W] | strag[1 + S]
  size_t r_offs_at, r_claims_at, r_strag_at, recv_bytes;          // recv buffer: items[nbg * cap] | offs[2 * nbg] | claims[W] | W x strag[1 + S]
};                                                                // (claims[p]: k-mers the sender says it sends to rank p: bookkeeping for sent == received)
// nbg = 2^gbits sender buckets = W owners x nbc = 2^cbits coarse buckets each; the receiver splits a coarse bucket into
// 2^split_bits of its own 2^b1 P1 buckets (b1 = cbits + split_bits).  gbits = min(10, shard_bits + b1).
ItemLayout item_layout(const jfgpu_comm* c, const jfgpu_table* t, uint32_t cap) {
  ItemLayout L;
  uint32_t sb = 0; while((1 << sb) < c->world) ++sb;
  L.gbits = std::min<uint32_t>((uint32_t)c->tun.comm_gbits, sb + t->pg.b1); L.cbits = L.gbits - sb; L.split_bits = t->pg.b1 - L.cbits;
  L.cap = cap; L.nbg = 1u << L.gbits; L.nbc = 1u << L.cbits; L.S = c->strag_cap;
  L.items_bytes = align_up((size_t)L.nbg * cap * 4, 256);
  L.offs_at = L.items_bytes; L.claims_at = L.offs_at + (size_t)2 * L.nbg * 8; L.strag_at = L.claims_at + 1024 * 8;
  L.send_bytes = L.strag_at + (size_t)(1 + L.S) * 8;
  L.r_offs_at = L.items_bytes; L.r_claims_at = L.r_offs_at + (size_t)2 * L.nbg * 8; L.r_strag_at = L.r_claims_at + 1024 * 8;
  L.recv_bytes = L.r_strag_at + (size_t)c->world * (1 + L.S) * 8;
  return L;
}

bool items_geometry_ok(const jfgpu_comm* c, const jfgpu_table* t) {
  if(!c->items_on || t->wide || t->nword || c->world > 512) return false;
  if(!t->part_ok || !t->item32 || t->pg.b2 == 0) return false;              // the shard inserts through two partition levels, 32-bit items
  uint32_t sb = 0; while((1 << sb) < c->world) ++sb;
  const uint32_t gbits = std::min<uint32_t>((uint32_t)c->tun.comm_gbits, sb + t->pg.b1);
  if(gbits < sb || t->g.key_bits < gbits || t->g.key_bits - gbits > 32) return false;   // the routed item is 2k - gbits bits
  if(t->pg.b1 - (gbits - sb) > 4) return false;                                          // the receiver splits a coarse bucket at most 16 ways
  return t->g.lsize_g >= t->g.tile_bits + gbits;
}

// Region capacity this rank wants for a step of n bytes (0: it would rather send keys).
uint32_t items_cap_wanted(const jfgpu_comm* c, const jfgpu_comm::Rank& R, size_t n) {
  const jfgpu_table* t = R.t;
  if(!items_geometry_ok(c, t) || t->operation != 0) return 0;              // (the PRIME / UPDATE passes of count --if travel as keys)
  const ItemLayout L = item_layout(c, t, 64);
  // Head-room (round 6: JFGPU_COMM_SLACK, 3 % on the k-mers-per-byte figure the rank measured on its earlier steps and 3 % on
  // the mean; it was 10 % + 10 %).  Regions travel whole, so head-room is wire bytes: 1.21 x + strand was 24 % over the mean
  // at the metric's step, this is 9 %.  A bucket of ~10^6 items deviates from the mean by 0.1 % (sigma); what the slack has
  // to absorb is the input changing its k-mers per byte between steps -- and what does not fit a region goes on the
  // stragglers' list, a step that overflows the list is redone with keys: slower, never wrong.
  const double slack = c->tun.comm_slack;
  const double ipb = R.ipb > 0 ? std::min(1.0, R.ipb * (1.0 + slack) + 0.002) : 1.0;
  const uint64_t items = (uint64_t)((double)n * ipb) + 4096;
  const uint64_t strand = (uint64_t)(2 * t->n_cu) * kGran, mean = (items + L.nbg - 1) / L.nbg;
  if(mean < 4 * strand && c->items_mode < 2) return 0;                      // regions would be mostly holes (2: forced, for tests)
  const uint64_t cap = ((uint64_t)((double)mean * (1.0 + slack)) + strand + kGran - 1) / kGran * kGran;
  if(cap > 0x7FFF0000ull) return 0;
#if !defined(JFGPU_EMU)
  // (inter-process test transport, observed and not understood: with item-path send buffers of 2.6 GB and more the first
  //  hipIpcOpenMemHandle of a peer's buffer never returns -- 1.8 GB is fine, and so are 5 GB buffers of the key path and
  //  any size in tools/probes/r03_ipc_size_probe.hip.  Steps that large go as keys there.)
  if(c->ipc && (uint64_t)L.nbg * cap * 5 > ((uint64_t)1 << 31)) return 0;
#endif
  return (uint32_t)cap;
}

// What every owner is being sent by this rank (its regions' items + its stragglers), written where the exchange takes it
// from (claims[W] of the send buffer), and the two numbers the host wants back (items stored, stragglers): on the device,
// so that the host has nothing to compute between the routing kernel and the exchange.
__global__ __launch_bounds__(1024) void comm_claims_kernel(const unsigned long long* __restrict__ tot, uint32_t nbg, uint32_t nbc, const uint64_t* __restrict__ strag,
                                                            uint32_t S, uint32_t world, uint64_t* __restrict__ claims, uint64_t* __restrict__ summary) {
  __shared__ unsigned long long s_claims[512];
  __shared__ unsigned long long s_stored;
  for(uint32_t p = threadIdx.x; p < world; p += blockDim.x) s_claims[p] = 0;
  if(threadIdx.x == 0) s_stored = 0;
  lds_barrier();
  for(uint32_t j = threadIdx.x; j < nbg; j += blockDim.x) { const unsigned long long v = tot[j]; if(v) { atomicAdd(&s_claims[j / nbc], v); atomicAdd(&s_stored, v); } }
  const uint64_t ns = strag[0];
  const uint64_t nl = ns < S ? ns : S;
  for(uint64_t i = threadIdx.x; i < nl; i += blockDim.x) atomicAdd(&s_claims[(uint32_t)(strag[1 + i] >> 32) / nbc], 1ull);
  lds_barrier();
  for(uint32_t p = threadIdx.x; p < world; p += blockDim.x) claims[p] = s_claims[p];
  if(threadIdx.x == 0) { summary[0] = s_stored; summary[1] = ns; }
}

// A Bloom counter attached to a shard (count --bc) is asked on the SENDING side, every rank holding the whole read-only
// counter.  A one-pass filter (--bf-size) changes as it is asked and would see only its rank's reads; two-word keys: not built.
int comm_filter_ok(const jfgpu_table* t) {
  if(t->operation != 0 && t->nword) return fail(JFGPU_E_UNSUPPORTED, "count --if with --gpus: keys longer than two words are not built yet");
  if(t->wide && t->wt.bloom.data && t->wt.bloom.kind != 0) return fail(JFGPU_E_UNSUPPORTED, "--bf-size with --gpus: a one-pass filter cannot be sharded by input");
  if(!t->wide && t->dt.bloom.data && t->dt.bloom.kind != 0) return fail(JFGPU_E_UNSUPPORTED, "--bf-size with --gpus: a one-pass filter cannot be sharded by input");
  return JFGPU_OK;
}

// P1 over the global table into send[cur]: enqueued here, looked at in comm_route_items_complete -- between the two the
// caller enqueues the insert of what arrived for the previous step, so the device has work while the host waits for the
// two numbers it needs (stragglers, exact item count).
int comm_route_items_enqueue(jfgpu_comm* c, jfgpu_comm::Rank& R, const char* d_bases, size_t n, uint32_t cap) {
  jfgpu_table* t = R.t;
  const int cur = R.turn;
  const ItemLayout L = item_layout(c, t, cap);
  { const int rc_ = comm_filter_ok(t); if(rc_) return rc_; }
  if(R.used[cur]) HIP_TRY(hipEventSynchronize(R.exchanged[cur]));       // send[cur] has left (step - 2)
  int rc = comm_reserve_send(c, R, cur, (L.send_bytes + 7) / 8, t->stream); if(rc) return rc;
  if(!R.used[cur ^ 1]) { rc = comm_reserve_send(c, R, cur ^ 1, (L.send_bytes + 7) / 8, t->stream); if(rc) return rc; }   // (both buffers of the pair at once)
  uint8_t* sb = reinterpret_cast<uint8_t*>(R.send[cur]);
  uint32_t* items = reinterpret_cast<uint32_t*>(sb);
  uint64_t* offs = reinterpret_cast<uint64_t*>(sb + L.offs_at);
  uint64_t* strag = reinterpret_cast<uint64_t*>(sb + L.strag_at);
  unsigned long long* tot = reinterpret_cast<unsigned long long*>(R.d_gcur + 2 * L.nbg);
  HIP_TRY(hipMemsetAsync(R.d_gcur, 0, 1024 * 16, t->stream));
  HIP_TRY(hipMemsetAsync(strag, 0, 8, t->stream));
  if(n >= t->g.k) {
    DevTable gv = t->dt;                                   // the shard's table seen as one table of 2^lsize_g slots
    gv.g.lsize_l = gv.g.lsize_g; gv.g.shard_bits = 0; gv.g.shard_id = 0;
    gv.g.local_mask = gv.g.lsize_g >= 64 ? ~0ull : ((1ull << gv.g.lsize_g) - 1);
    PartGeom pg; memset(&pg, 0, sizeof pg);
    pg.b1 = L.gbits; pg.b2 = gv.g.lsize_g - gv.g.tile_bits - L.gbits; pg.rest_shift = gv.g.lsize_g - L.gbits; pg.item_bits = pg.rest_shift + gv.g.rem_bits;
    const uint8_t* base; int64_t lo, hi;
    align_buffer(d_bases, n, base, lo, hi);
    const RouteListDirect rd{StragList{reinterpret_cast<unsigned long long*>(strag), strag + 1, L.S, 0}};
    { const int rc2 = ensure_strag(t); if(rc2) return rc2; }
    const dim3 grid(t->n_cu), block(kPBlock);
    const size_t lds = ((size_t)1 << pg.b1) * 128 + 128;
    ProfScope ps(t, 2, n);
    // the count path's ring P1 (kernels_p1ring.hip.hpp) over the global geometry; what it cannot store goes on the list
#define PR(N, CN) hipLaunchKernelGGL((p1_ring_kernel<uint32_t, false, N, CN, RouteListDirect>), grid, block, lds, t->stream, gv, rd, pg, base, lo, hi, cap, R.d_gcur, tot, items, t->d_strag, t->d_strag_n)
    if(t->dt.bloom.data)      // count --bc: the sender asks its copy of the Bloom counter (read-only: the same answer on every rank)
      hipLaunchKernelGGL((p1_ring_kernel<uint32_t, true, 0, 2, RouteListDirect>), grid, block, lds, t->stream, gv, rd, pg, base, lo, hi, cap, R.d_gcur, tot, items, t->d_strag, t->d_strag_n);
    else if(t->g.hash_xs && gv.g.lsize_g > 32 && gv.g.lsize_g < 64 && gv.g.lsize_g - pg.b1 < 32) { if(t->g.canonical) PR(kHashXS, 1); else PR(kHashXS, 0); }
    else if(t->g.hash_xs) PR(kHashXSLow, 2);
    else if(t->g.nbytes == 6) { if(t->g.canonical) PR(6, 1); else PR(6, 0); } else PR(0, 2);
#undef PR
    hipLaunchKernelGGL((p1_stragglers_kernel<uint32_t, RouteListDirect>), dim3(t->n_cu), dim3(256), 0, t->stream, rd, (unsigned long long*)nullptr, (const uint64_t*)t->d_strag,
                       (const uint32_t*)t->d_strag_n, (uint32_t)t->n_cu, cap, R.d_gcur, tot, items, kStragPerBlock);
  }
  hipLaunchKernelGGL(granule_finish_kernel, dim3(4), dim3(256), 0, t->stream, R.d_gcur, cap, L.nbg, offs);
  hipLaunchKernelGGL(comm_claims_kernel, dim3(1), dim3(1024), 0, t->stream, tot, L.nbg, L.nbc, (const uint64_t*)strag, L.S, (uint32_t)c->world,
                     reinterpret_cast<uint64_t*>(sb + L.claims_at), R.d_route);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(R.h_route, R.d_route, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipEventRecord(R.route_done, t->stream));
  R.route_n = n;
  return JFGPU_OK;
}

// *overflow: more stragglers than the list holds -- the step has to be redone with keys.
int comm_route_items_complete(jfgpu_comm* c, jfgpu_comm::Rank& R, uint32_t cap, bool* overflow, uint64_t* routed) {
  jfgpu_table* t = R.t;
  const ItemLayout L = item_layout(c, t, cap);
  *overflow = false; *routed = 0;
  HIP_TRY(hipEventSynchronize(R.route_done));
  const uint64_t stored = R.h_route[0], ns = R.h_route[1];
  if(ns > L.S) { *overflow = true; return JFGPU_OK; }
  if(R.route_n >= ((size_t)1 << 20)) R.ipb = (double)(stored + ns) / (double)R.route_n;
  *routed = stored + ns;
  if(t->tun.flush_trace)
    fprintf(stderr, "[comm] item path: %zu bytes -> %llu items in %u regions of %u, %llu stragglers\n", R.route_n, (unsigned long long)stored, L.nbg, cap, (unsigned long long)ns);
  return JFGPU_OK;
}

__global__ void comm_add_claims_kernel(const uint64_t* __restrict__ claims, int n, unsigned long long* __restrict__ total) {
  if(blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long s = 0; for(int i = 0; i < n; ++i) s += claims[i]; *total += s; }
}

// The receiver's own count of what arrived (the claims above are the senders' word): before the split, remember the
// table's direct-insert counter; after it, add what the split stored in its regions (tot[]) and what it inserted directly.
__global__ void comm_mark_direct_kernel(const uint64_t* __restrict__ counters, unsigned long long* __restrict__ arrived) {
  if(blockIdx.x == 0 && threadIdx.x == 0) arrived[1] = counters[CTR_DIRECT];
}
__global__ void comm_count_arrived_kernel(const unsigned long long* __restrict__ tot, uint32_t nb, const uint64_t* __restrict__ counters,
                                          unsigned long long* __restrict__ arrived) {
  unsigned long long s = 0;
  for(uint32_t j = threadIdx.x; j < nb; j += blockDim.x) s += tot[j];
  for(int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if((threadIdx.x & 63) == 0 && s) atomicAdd(&arrived[0], s);
  if(threadIdx.x == 0) atomicAdd(&arrived[0], (unsigned long long)(counters[CTR_DIRECT] - arrived[1]));
}

// What arrived for the previous step (item path): one split of every coarse bucket into the shard's own P1 buckets, then
// the stragglers; the result is a pending batch.
int comm_insert_prev_items(jfgpu_comm* c, jfgpu_comm::Rank& R, int rank) {
  const int prev = R.turn ^ 1, W = c->world;
  jfgpu_table* t = R.t;
  const ItemLayout L = item_layout(c, t, R.icap[prev]);
  const uint32_t sb = L.split_bits;                        // bits that finish the shard's own bucket index
  const uint32_t nb = 1u << t->pg.b1;
  // blocks per coarse bucket: about two workgroups per CU in all (world 8: 4 x 128 buckets; world 1: 1 x 1024)
  const uint32_t kBlocksPerBucket = std::max<uint32_t>(1, std::min<uint32_t>(4, (2u * (uint32_t)t->n_cu) / std::max<uint32_t>(1, L.nbc)));
  // a fine bucket takes its share (1 / 2^sb) of a coarse bucket's W incoming regions.  (Round 4: this said L.cap, which is
  // right only when the fan-out equals W -- a shard with 2^9 P1 buckets at world 4 splits two ways, its regions overflowed
  // and a third of the items went in by global atomics: tools/local_world_stage_times.py.)
  const uint64_t per_fine = ((uint64_t)W * L.cap + ((uint64_t)1 << sb) - 1) >> sb;
  // the wave-per-stream split (recv_split_kernel; JFGPU_COMM_SPLIT=0: round 4's sort-based kernel): about four workgroups
  // per CU in all; a wave reserves `res` items at a time and may leave two reservations per destination partly unused
  const bool wave_split = t->tun.comm_split != 0;
  const uint32_t split_wgs = std::max<uint32_t>(1, std::min<uint32_t>(16, (4u * (uint32_t)t->n_cu) / std::max<uint32_t>(1, L.nbc)));
  const uint32_t split_waves = split_wgs * (uint32_t)kSplitWaves;
  const uint32_t row = split_row(sb);                      // items a wave writes at a time per destination
  const uint32_t res = row * (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(1024 / row, per_fine / ((uint64_t)row * split_waves * 16)));
  const uint64_t strand = wave_split ? 2ull * split_waves * res : (uint64_t)kBlocksPerBucket * kGran;
  if(per_fine + strand + kGran > 0xFFFF0000ull) return fail(JFGPU_E_UNSUPPORTED, "item exchange: a step too large for its regions");
  uint32_t cap2 = (uint32_t)((per_fine + strand + kGran - 1) / kGran * kGran);
  if(t->tun.comm_split_cap) cap2 = (t->tun.comm_split_cap + kGran - 1) / kGran * kGran;      // (tests: regions that overflow)
  const size_t bytes = (size_t)nb * cap2 * 4;
  const size_t need = align_up(bytes, 256) + align_up((2 * nb + 1) * sizeof(uint64_t), 256) + align_up(nb * 16, 256) + 1024;
  if(t->pending.size() >= kMaxSeg) { int rc = part_flush(t); if(rc) return rc; }
  if(t->ws_used + need > t->ws_cap) {
    if(!t->pending.empty()) { int rc = part_flush(t); if(rc) return rc; }
    if(need > t->ws_cap) { int rc = ws_grow(t, need); if(rc) return rc < 0 ? fail(JFGPU_E_ALLOC, "no device memory for the partition workspace of a shard") : rc; }
  }
  HIP_TRY(hipStreamWaitEvent(t->stream, R.exchanged[prev], 0));
  PendingBatch b{nullptr, nullptr, (uint64_t)nb * cap2};
  b.items = ws_alloc(t, bytes);
  b.off = (uint64_t*)ws_alloc(t, (2 * nb + 1) * sizeof(uint64_t));
  unsigned int* gcur = (unsigned int*)ws_alloc(t, nb * 16);             // gcur[2 nb] (u32) then tot[nb] (u64)
  if(!b.items || !b.off || !gcur) return fail(JFGPU_E_ALLOC, "partition workspace exhausted");
  b.gran_cap = cap2; b.tot = (unsigned long long*)(gcur + 2 * nb);
  HIP_TRY(hipMemsetAsync(gcur, 0, nb * 16, t->stream));
  const uint8_t* rb = reinterpret_cast<const uint8_t*>(R.recv[prev]);
  const uint32_t* r_items = reinterpret_cast<const uint32_t*>(rb);
  const uint64_t* r_offs = reinterpret_cast<const uint64_t*>(rb + L.r_offs_at);
  const uint64_t* r_strag = reinterpret_cast<const uint64_t*>(rb + L.r_strag_at);
  hipLaunchKernelGGL(comm_add_claims_kernel, dim3(1), dim3(64), 0, t->stream, reinterpret_cast<const uint64_t*>(rb + L.r_claims_at), W, R.d_claimed);
  hipLaunchKernelGGL(comm_mark_direct_kernel, dim3(1), dim3(64), 0, t->stream, t->dt.counters, R.d_arrived);
  // sender p's regions of my coarse buckets sit at r_items + p * nbc * cap, its offsets speak of its own whole output
  // (bucket j of 1024 at j * cap): bases shifted so that "bucket = rank * nbc + coarse" finds both
  SegList S; memset(&S, 0, sizeof S);
  S.n = (uint32_t)W;
  const int64_t first = (int64_t)rank * L.nbc;
  for(int p = 0; p < W; ++p) {
    S.items[p] = r_items + (int64_t)p * L.nbc * L.cap - first * (int64_t)L.cap;
    if(p == rank && !c->local && R.self_items[prev]) S.items[p] = R.self_items[prev];      // (its offsets speak of the sender's whole output: no shift)
    S.off[p] = r_offs + (int64_t)p * 2 * L.nbc - 2 * first;
    S.sh[p] = 1;
  }
  { int rc = refresh_d_dt(t); if(rc) return rc; }
  PartGeom pd = t->pg;                                     // direct inserts of (coarse bucket, item)
  pd.b1 = L.cbits; pd.b2 = t->g.lsize_l - t->g.tile_bits - L.cbits;
  const uint32_t split_at = t->g.key_bits - L.gbits - sb;  // where those bits sit in a routed item (its top bits)
  const int64_t dshift = first * (int64_t)(1u << sb);      // destinations are numbered from this rank's first bucket
  {
    ProfScope ps(t, 1, 0);
    const dim3 grid(kBlocksPerBucket, L.nbc), block(kPBlock);
    const size_t lds = (size_t)kPBlock * kP2PairPer * 4;
    unsigned int* gc_v = gcur - dshift; unsigned int* gs_v = gcur + nb - dshift;
    uint32_t* out_v = reinterpret_cast<uint32_t*>(b.items) - dshift * (int64_t)cap2;
    unsigned long long* tot_v = b.tot - dshift;
    // (fan-out 2^sb <= 16: the packed-counter ranking of p2_granule_kernel, with 1, 2 or 4 words of four destinations)
#define SPLIT(RT, SW) hipLaunchKernelGGL((p2_granule_kernel<uint32_t, TableDirect<RT>, kP2PairPer, SW>), grid, block, lds, t->stream, \
                        TableDirect<RT>{t->d_dt, pd, (unsigned long long*)&t->dt.counters[CTR_DIRECT]}, sb, split_at, S, cap2, gc_v, gs_v, out_v, (uint32_t)first, tot_v, L.nbc - 1)
    if(sb > 4) return fail(JFGPU_E_UNSUPPORTED, "item exchange: more than 16 fine buckets per coarse bucket");
    if(wave_split) {
      const dim3 gridw(split_wgs, L.nbc), blockw(64 * kSplitWaves);
#define WSPLIT(RT, LF) hipLaunchKernelGGL((recv_split_kernel<TableDirect<RT>, LF>), gridw, blockw, 0, t->stream, \
                         TableDirect<RT>{t->d_dt, pd, (unsigned long long*)&t->dt.counters[CTR_DIRECT]}, split_at, S, cap2, res, gc_v, gs_v, out_v, (uint32_t)first, tot_v, L.nbc - 1)
#define WSPLIT_LF(RT) do { switch(sb) { case 0: WSPLIT(RT, 0); break; case 1: WSPLIT(RT, 1); break; case 2: WSPLIT(RT, 2); break; case 3: WSPLIT(RT, 3); break; default: WSPLIT(RT, 4); } } while(0)
      if(t->returning) WSPLIT_LF(true); else WSPLIT_LF(false);
#undef WSPLIT_LF
#undef WSPLIT
    }
    else if(t->returning) { if(sb <= 2) SPLIT(true, 1); else if(sb == 3) SPLIT(true, 2); else SPLIT(true, 4); }
    else             { if(sb <= 2) SPLIT(false, 1); else if(sb == 3) SPLIT(false, 2); else SPLIT(false, 4); }
#undef SPLIT
    hipLaunchKernelGGL(granule_finish_kernel, dim3(4), dim3(256), 0, t->stream, gcur, cap2, nb, b.off);
    hipLaunchKernelGGL(comm_count_arrived_kernel, dim3(1), dim3(256), 0, t->stream, b.tot, nb, t->dt.counters, R.d_arrived);
    for(int p = 0; p < W; ++p) {
      const uint64_t* lst = r_strag + (size_t)p * (1 + L.S);
      if(t->returning) hipLaunchKernelGGL(straggler_insert_kernel<true>, dim3(64), dim3(kBlock), 0, t->stream, t->dt, pd, lst + 1, (const unsigned long long*)lst, L.S, (uint32_t)rank, L.cbits, R.d_arrived);
      else             hipLaunchKernelGGL(straggler_insert_kernel<false>, dim3(64), dim3(kBlock), 0, t->stream, t->dt, pd, lst + 1, (const unsigned long long*)lst, L.S, (uint32_t)rank, L.cbits, R.d_arrived);
    }
  }
  HIP_TRY(hipGetLastError());
  if(t->tun.flush_trace) {          // what the split stored per fine bucket
    std::vector<unsigned long long> ht(nb);
    HIP_TRY(hipStreamSynchronize(t->stream));
    HIP_TRY(hipMemcpy(ht.data(), b.tot, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long sum = 0, mn = ~0ull, mx = 0;
    for(unsigned long long v : ht) { sum += v; mn = std::min(mn, v); mx = std::max(mx, v); }
    fprintf(stderr, "[comm] split of rank %d: %llu items into %u regions of %u (per region: min %llu, max %llu), fan-out %u, %u workgroups per coarse bucket; first regions:",
            rank, sum, nb, cap2, mn, mx, 1u << sb, kBlocksPerBucket);
    for(uint32_t j = 0; j < 8 && j < nb; ++j) fprintf(stderr, " %llu", ht[j]);
    fprintf(stderr, "\n");
  }
  t->pending.push_back(b);
  t->pending_bytes += bytes;
  t->pristine = false;
  HIP_TRY(hipEventRecord(R.consumed[prev], t->stream));
  R.inflight = false;
  return JFGPU_OK;
}

int comm_reserve(uint64_t*& buf, size_t& cap, size_t need, hipStream_t s1, hipStream_t s2) {
  if(need <= cap) return JFGPU_OK;
  HIP_TRY(hipStreamSynchronize(s1)); HIP_TRY(hipStreamSynchronize(s2));
  if(buf) hipFree(buf);
  buf = nullptr; cap = 0;
  const size_t want = need + need / 4 + 1024;      // steps differ in size: head-room, so that the buffers are not re-allocated mid-job
  HIP_TRY(hipMalloc((void**)&buf, want * sizeof(uint64_t)));
  cap = want;
  return JFGPU_OK;
}

// A SEND buffer of the ipc transport is mapped by the peers.  Freeing it under their mappings and allocating the larger one
// right away made the peers read stale memory through the handle of the new buffer (round 6: shards that grow with most of
// their entries leaving -- the xor-shift family, world 4 -- re-allocate both send buffers inside comm_grow, and the pairs
// arrived wrong: profiles/r06_ipc_realloc.log).  So the old buffer is only RETIRED here; it is freed after the next exchange
// of its turn, whose import phase makes every peer close its mapping of it before the new handle is opened.
int comm_reserve_send(jfgpu_comm* c, jfgpu_comm::Rank& R, int turn, size_t need, hipStream_t s1) {
  if(need <= R.send_cap[turn]) return JFGPU_OK;
  if(c->ipc && R.send[turn]) {
    HIP_TRY(hipStreamSynchronize(s1)); HIP_TRY(hipStreamSynchronize(c->xstream));
    c->send_retired[turn].push_back(R.send[turn]);
    R.send[turn] = nullptr; R.send_cap[turn] = 0;
  }
  return comm_reserve(R.send[turn], R.send_cap[turn], need, s1, c->xstream);
}
void comm_free_retired(jfgpu_comm* c, int turn, bool closing = false) {
  if(c->tun.comm_ipc_keep && !closing) return;                 // (kept until the communicator goes: see tuning.hpp, JFGPU_IPC_KEEP)
  for(void* p : c->send_retired[turn]) hipFree(p);
  c->send_retired[turn].clear();
}

// Route one contract buffer of rank R into send[cur], grouped by owner; fills scount / soff.  The host waits for the
// per-owner counts (one small copy) -- they place the groups and size the messages.
int comm_route(jfgpu_comm* c, jfgpu_comm::Rank& R, const char* d_bases, size_t n) {
  jfgpu_table* t = R.t;
  const int cur = R.turn, W = c->world;
  if((int)(1u << t->g.shard_bits) != W) return fail(JFGPU_E_INVALID, "table shard_bits does not match the communicator's world size");
  if(t->nword) return fail(JFGPU_E_UNSUPPORTED, "sharded tables with mer length > 64 are not built yet");
  { const int rc_ = comm_filter_ok(t); if(rc_) return rc_; }
  const uint64_t kw = t->wide ? 2 : 1;                                   // 64-bit words per routed k-mer (counts and offsets below are in words)
  if(R.used[cur]) HIP_TRY(hipEventSynchronize(R.exchanged[cur]));       // send[cur] has left (step - 2)
  std::fill(R.scount[cur].begin(), R.scount[cur].end(), 0);
  std::fill(R.soff[cur].begin(), R.soff[cur].end(), 0);
  if(n < t->g.k) return JFGPU_OK;
  int rc = comm_reserve_send(c, R, cur, n * kw, t->stream); if(rc) return rc;
  if(!R.used[cur ^ 1]) { rc = comm_reserve_send(c, R, cur ^ 1, n * kw, t->stream); if(rc) return rc; }   // (both buffers of the pair at once)
  const uint8_t* base; int64_t lo, hi;
  align_buffer(d_bases, n, base, lo, hi);
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  const int grid = grid_for(t, (uint64_t)n_tiles);
  HIP_TRY(hipMemsetAsync(R.d_cnt, 0, sizeof(unsigned long long) * W, t->stream));
  {
    ProfScope ps(t, 2, n);
    if(t->wide && t->wt.bloom.data) hipLaunchKernelGGL(partition_count_wide_kernel<true>, dim3(grid), dim3(kBlock), 0, t->stream, t->wt, base, lo, hi, R.d_cnt);
    else if(t->wide) hipLaunchKernelGGL(partition_count_wide_kernel<false>, dim3(grid), dim3(kBlock), 0, t->stream, t->wt, base, lo, hi, R.d_cnt);
    else if(t->dt.bloom.data) hipLaunchKernelGGL(partition_count_kernel<true>, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, base, lo, hi, R.d_cnt);
    else hipLaunchKernelGGL(partition_count_kernel<false>, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, base, lo, hi, R.d_cnt);
  }
  std::vector<unsigned long long> h(W);
  HIP_TRY(hipMemcpyAsync(h.data(), R.d_cnt, sizeof(unsigned long long) * W, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  uint64_t total = 0;
  for(int p = 0; p < W; ++p) { R.scount[cur][p] = h[p] * kw; R.soff[cur][p] = total * kw; const uint64_t first = total; total += h[p]; h[p] = first; }      // (cursors in k-mers, messages in words)
  R.soff[cur][W] = total * kw;
  HIP_TRY(hipMemcpyAsync(R.d_cnt, h.data(), sizeof(unsigned long long) * W, hipMemcpyHostToDevice, t->stream));   // cursors = offsets
  {
    ProfScope ps(t, 2, 0);
    if(t->wide && t->wt.bloom.data) hipLaunchKernelGGL(partition_scatter_wide_kernel<true>, dim3(grid), dim3(kBlock), 0, t->stream, t->wt, base, lo, hi, R.d_cnt, R.send[cur]);
    else if(t->wide) hipLaunchKernelGGL(partition_scatter_wide_kernel<false>, dim3(grid), dim3(kBlock), 0, t->stream, t->wt, base, lo, hi, R.d_cnt, R.send[cur]);
    else if(t->dt.bloom.data) hipLaunchKernelGGL(partition_scatter_kernel<true>, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, base, lo, hi, R.d_cnt, R.send[cur]);
    else hipLaunchKernelGGL(partition_scatter_kernel<false>, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, base, lo, hi, R.d_cnt, R.send[cur]);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(t->stream));       // the host vector h is read by the copy above
  R.sent += total;
  return JFGPU_OK;
}

// What arrived for the previous step goes into the table (P1 from keys: pending batch applied at the next flush).
int comm_insert_prev_items(jfgpu_comm* c, jfgpu_comm::Rank& R, int rank);
int comm_insert_prev(jfgpu_comm* c, jfgpu_comm::Rank& R) {
  if(!R.inflight) return JFGPU_OK;
  const int prev = R.turn ^ 1;
  if(R.icap[prev]) return comm_insert_prev_items(c, R, c->local ? (int)(&R - c->ranks.data()) : c->rank);
  jfgpu_table* t = R.t;
  HIP_TRY(hipStreamWaitEvent(t->stream, R.exchanged[prev], 0));
  const uint64_t n = R.roff[prev][c->world] / (t->wide ? 2 : 1);          // k-mers that arrived (the offsets are in words)
  int rc = JFGPU_OK;
  // what the table does with a k-mer (jfgpu_set_operation: the passes of count --if) holds for what arrives, too
  if(n && t->operation == 2) {
    rc = part_flush(t); if(rc) return rc;
    ProfScope ps(t, 1, n);
    const int grid = grid_for(t, (n + kBlock - 1) / kBlock);
    if(t->wide) {
      if(t->returning) hipLaunchKernelGGL(update_keys_wide_kernel<true>, dim3(grid), dim3(kBlock), 0, t->stream, t->wt, (const uint64_t*)R.recv[prev], (uint64_t)n);
      else             hipLaunchKernelGGL(update_keys_wide_kernel<false>, dim3(grid), dim3(kBlock), 0, t->stream, t->wt, (const uint64_t*)R.recv[prev], (uint64_t)n);
    }
    else if(t->returning) hipLaunchKernelGGL(update_keys_one_kernel<true>, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, (const uint64_t*)R.recv[prev], (uint64_t)n);
    else             hipLaunchKernelGGL(update_keys_one_kernel<false>, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, (const uint64_t*)R.recv[prev], (uint64_t)n);
    HIP_TRY(hipGetLastError());
  } else if(n) rc = add_keys_piece(t, R.recv[prev], (size_t)n, t->operation == 1 ? 0 : 1, nullptr);
  HIP_TRY(hipEventRecord(R.consumed[prev], t->stream));
  R.received += n;
  R.inflight = false;
  return rc;
}

#if !defined(JFGPU_EMU)
#define NCCL_TRY(expr) do { ncclResult_t r_ = (expr); if(r_ != ncclSuccess) return fail(JFGPU_E_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_)); } while(0)
#endif

// The exchange of the current step for the RCCL transport: counts (host-visible), then the keys in rounds.
// exchange timing: harvest a turn's finished pair (the caller knows its exchange is over), open it for the next one
void comm_x_harvest(jfgpu_comm* c, jfgpu_comm::Rank& R, int turn) {
  if(!R.x_open[turn]) return;
  float ms = 0;
  if(hipEventSynchronize(R.x_end[turn]) == hipSuccess && hipEventElapsedTime(&ms, R.x_begin[turn], R.x_end[turn]) == hipSuccess && c->x_log.size() < ((size_t)1 << 16)) {
    if(c->x_log.size() <= R.x_seq[turn]) c->x_log.resize(R.x_seq[turn] + 1, std::make_pair(-1.0, (uint64_t)0));
    c->x_log[R.x_seq[turn]] = std::make_pair((double)ms, R.x_bytes[turn]);
  }
  R.x_open[turn] = false;
}
void comm_x_begin(jfgpu_comm* c, jfgpu_comm::Rank& R, int turn) {
  comm_x_harvest(c, R, turn);
  (void)hipEventRecord(R.x_begin[turn], c->xstream);
  R.x_seq[turn] = c->x_next++; R.x_bytes[turn] = 0; R.x_open[turn] = true;
}
void comm_x_end(jfgpu_comm* c, jfgpu_comm::Rank& R, int turn, uint64_t wire_bytes) {
  (void)hipEventRecord(R.x_end[turn], c->xstream);
  R.x_bytes[turn] = wire_bytes;
}

int comm_exchange_rccl(jfgpu_comm* c) {
#if defined(JFGPU_EMU)
  (void)c;
  return fail(JFGPU_E_UNSUPPORTED, "no RCCL in the emulated build");
#else
  jfgpu_comm::Rank& R = c->ranks[0];
  const int cur = R.turn, W = c->world;
  // counts: W words out, W words in (what a rank keeps for itself never leaves the device: plain copies below)
  const bool via_rccl = W > 1 || c->self_rccl;
  const int skip = c->self_rccl ? -1 : c->rank;          // the peer served by a plain device copy
  R.rcount[cur][c->rank] = R.scount[cur][c->rank];
  if(via_rccl) {
    HIP_TRY(hipMemcpyAsync(R.d_xc, R.scount[cur].data(), sizeof(uint64_t) * W, hipMemcpyHostToDevice, c->xstream));
    NCCL_TRY(ncclGroupStart());
    for(int p = 0; p < W; ++p) {
      if(p == skip) continue;
      NCCL_TRY(ncclSend(R.d_xc + p, 1, ncclUint64, p, c->nccl, c->xstream));
      NCCL_TRY(ncclRecv(R.d_xc + W + p, 1, ncclUint64, p, c->nccl, c->xstream));
    }
    NCCL_TRY(ncclGroupEnd());
    std::vector<uint64_t> got(W);
    HIP_TRY(hipMemcpyAsync(got.data(), R.d_xc + W, sizeof(uint64_t) * W, hipMemcpyDeviceToHost, c->xstream));
    HIP_TRY(hipStreamSynchronize(c->xstream));
    for(int p = 0; p < W; ++p) if(p != skip) R.rcount[cur][p] = got[p];
  }
  uint64_t total = 0, gmax = 0;
  for(int p = 0; p < W; ++p) { R.roff[cur][p] = total; total += R.rcount[cur][p]; gmax = std::max(gmax, std::max(R.rcount[cur][p], R.scount[cur][p])); }
  R.roff[cur][W] = total;
  IPC_TRACE(c, "keys: %llu to pull; waiting for recv[%d] to be consumed", (unsigned long long)total, cur);
  if(R.used[cur]) HIP_TRY(hipEventSynchronize(R.consumed[cur]));          // recv[cur] was read by the insert of step - 2
  int rc = comm_reserve(R.recv[cur], R.recv_cap[cur], total, R.t->stream, c->xstream); if(rc) return rc;
  if(!R.used[cur ^ 1]) { rc = comm_reserve(R.recv[cur ^ 1], R.recv_cap[cur ^ 1], total, R.t->stream, c->xstream); if(rc) return rc; }
  HIP_TRY(hipEventRecord(R.routed[cur], R.t->stream));
  HIP_TRY(hipStreamWaitEvent(c->xstream, R.routed[cur], 0));
  comm_x_begin(c, R, cur);
  // this rank's own share: a device copy (1/W of the keys; everything, for a world of one)
  if(skip >= 0 && R.scount[cur][c->rank])
    HIP_TRY(hipMemcpyAsync(R.recv[cur] + R.roff[cur][c->rank], R.send[cur] + R.soff[cur][c->rank], R.scount[cur][c->rank] * sizeof(uint64_t),
                           hipMemcpyDeviceToDevice, c->xstream));
  if(via_rccl) {
    // every rank runs the same number of rounds: the largest peer-to-peer message of this step decides
    unsigned long long lmax = 0, *d_m = (unsigned long long*)R.d_xc;
    for(int p = 0; p < W; ++p) if(p != skip) lmax = std::max<unsigned long long>(lmax, std::max(R.rcount[cur][p], R.scount[cur][p]));
    HIP_TRY(hipMemcpyAsync(d_m, &lmax, sizeof lmax, hipMemcpyHostToDevice, c->xstream));
    NCCL_TRY(ncclAllReduce(d_m, d_m, 1, ncclUint64, ncclMax, c->nccl, c->xstream));
    HIP_TRY(hipMemcpyAsync(&lmax, d_m, sizeof lmax, hipMemcpyDeviceToHost, c->xstream));
    HIP_TRY(hipStreamSynchronize(c->xstream));
    const uint64_t rounds = std::max<uint64_t>(1, (lmax + c->max_msg_keys - 1) / c->max_msg_keys);
    for(uint64_t r = 0; r < rounds; ++r) {
      const uint64_t lo = r * c->max_msg_keys;
      NCCL_TRY(ncclGroupStart());
      for(int p = 0; p < W; ++p) {
        if(p == skip) continue;
        const uint64_t sc = R.scount[cur][p] > lo ? std::min(R.scount[cur][p] - lo, c->max_msg_keys) : 0;
        const uint64_t rcn = R.rcount[cur][p] > lo ? std::min(R.rcount[cur][p] - lo, c->max_msg_keys) : 0;
        if(sc) NCCL_TRY(ncclSend(R.send[cur] + R.soff[cur][p] + lo, sc, ncclUint64, p, c->nccl, c->xstream));
        if(rcn) NCCL_TRY(ncclRecv(R.recv[cur] + R.roff[cur][p] + lo, rcn, ncclUint64, p, c->nccl, c->xstream));
      }
      NCCL_TRY(ncclGroupEnd());
    }
  }
  { uint64_t wire = 0; for(int p = 0; p < W; ++p) if(p != skip && via_rccl) wire += R.scount[cur][p] * 8; comm_x_end(c, R, cur, wire); }
  HIP_TRY(hipEventRecord(R.exchanged[cur], c->xstream));
  R.used[cur] = true;
  return JFGPU_OK;
#endif
}

// The same for the local transport: all ranks live here, messages are device copies on the exchange stream.
int comm_exchange_local(jfgpu_comm* c) {
  const int W = c->world;
  for(int d = 0; d < W; ++d) {                    // receive side bookkeeping of rank d
    jfgpu_comm::Rank& D = c->ranks[d];
    const int cur = D.turn;
    uint64_t total = 0;
    for(int s = 0; s < W; ++s) { D.rcount[cur][s] = c->ranks[s].scount[c->ranks[s].turn][d]; D.roff[cur][s] = total; total += D.rcount[cur][s]; }
    D.roff[cur][W] = total;
    if(D.used[cur]) HIP_TRY(hipEventSynchronize(D.consumed[cur]));
    int rc = comm_reserve(D.recv[cur], D.recv_cap[cur], total, D.t->stream, c->xstream); if(rc) return rc;
  }
  for(int s = 0; s < W; ++s) {
    jfgpu_comm::Rank& S = c->ranks[s];
    HIP_TRY(hipEventRecord(S.routed[S.turn], S.t->stream));
    HIP_TRY(hipStreamWaitEvent(c->xstream, S.routed[S.turn], 0));
  }
  uint64_t gmax = 0;
  for(int s = 0; s < W; ++s) for(int d = 0; d < W; ++d) gmax = std::max(gmax, c->ranks[s].scount[c->ranks[s].turn][d]);
  const uint64_t rounds = std::max<uint64_t>(1, (gmax + c->max_msg_keys - 1) / c->max_msg_keys);
  for(uint64_t r = 0; r < rounds; ++r) {
    const uint64_t lo = r * c->max_msg_keys;
    for(int s = 0; s < W; ++s)
      for(int d = 0; d < W; ++d) {
        jfgpu_comm::Rank &S = c->ranks[s], &D = c->ranks[d];
        const uint64_t n = S.scount[S.turn][d] > lo ? std::min(S.scount[S.turn][d] - lo, c->max_msg_keys) : 0;
        if(n) HIP_TRY(hipMemcpyAsync(D.recv[D.turn] + D.roff[D.turn][s] + lo, S.send[S.turn] + S.soff[S.turn][d] + lo, n * sizeof(uint64_t),
                                     hipMemcpyDeviceToDevice, c->xstream));
      }
  }
  for(int d = 0; d < W; ++d) {
    jfgpu_comm::Rank& D = c->ranks[d];
    HIP_TRY(hipEventRecord(D.exchanged[D.turn], c->xstream));
    D.used[D.turn] = true;
  }
  return JFGPU_OK;
}

// The exchange of an item-path step, RCCL transport: equal-sized messages, nothing to agree on.
int comm_exchange_items_rccl(jfgpu_comm* c) {
#if defined(JFGPU_EMU)
  (void)c;
  return fail(JFGPU_E_UNSUPPORTED, "no RCCL in the emulated build");
#else
  jfgpu_comm::Rank& R = c->ranks[0];
  const int cur = R.turn, W = c->world;
  const ItemLayout L = item_layout(c, R.t, R.icap[cur]);
  if(R.used[cur]) HIP_TRY(hipEventSynchronize(R.consumed[cur]));          // recv[cur] was read by the insert of step - 2
  int rc = comm_reserve(R.recv[cur], R.recv_cap[cur], (L.recv_bytes + 7) / 8, R.t->stream, c->xstream); if(rc) return rc;
  if(!R.used[cur ^ 1]) { rc = comm_reserve(R.recv[cur ^ 1], R.recv_cap[cur ^ 1], (L.recv_bytes + 7) / 8, R.t->stream, c->xstream); if(rc) return rc; }
  HIP_TRY(hipEventRecord(R.routed[cur], R.t->stream));
  HIP_TRY(hipStreamWaitEvent(c->xstream, R.routed[cur], 0));
  comm_x_begin(c, R, cur);
  uint8_t* sb = reinterpret_cast<uint8_t*>(R.send[cur]);
  uint8_t* rb = reinterpret_cast<uint8_t*>(R.recv[cur]);
  const size_t blk = (size_t)L.nbc * L.cap;                               // items per (sender, receiver) message
  const int skip = c->self_rccl ? -1 : c->rank;
  auto s_items = [&](int p) { return reinterpret_cast<uint32_t*>(sb) + (size_t)p * blk; };
  auto r_items = [&](int p) { return reinterpret_cast<uint32_t*>(rb) + (size_t)p * blk; };
  auto s_offs = [&](int p) { return reinterpret_cast<uint64_t*>(sb + L.offs_at) + (size_t)p * 2 * L.nbc; };
  auto r_offs = [&](int p) { return reinterpret_cast<uint64_t*>(rb + L.r_offs_at) + (size_t)p * 2 * L.nbc; };
  uint64_t* s_claims = reinterpret_cast<uint64_t*>(sb + L.claims_at);
  uint64_t* r_claims = reinterpret_cast<uint64_t*>(rb + L.r_claims_at);
  uint64_t* s_strag = reinterpret_cast<uint64_t*>(sb + L.strag_at);
  auto r_strag = [&](int p) { return reinterpret_cast<uint64_t*>(rb + L.r_strag_at) + (size_t)p * (1 + L.S); };
  R.self_items[cur] = nullptr;
  if(skip >= 0) {
    const int p = c->rank;
    // The rank's own share does not move: the split reads it from the send buffer (stream order keeps it there: the next
    // routing into send[cur] is enqueued behind that split, on the table's stream).  At world 8 an eighth of the items, on
    // one GPU all of them (a 3.5 GB copy per step that ran beside the next step's routing kernel and slowed it).
    R.self_items[cur] = reinterpret_cast<const uint32_t*>(sb);
    HIP_TRY(hipMemcpyAsync(r_offs(p), s_offs(p), (size_t)2 * L.nbc * 8, hipMemcpyDeviceToDevice, c->xstream));
    HIP_TRY(hipMemcpyAsync(r_claims + p, s_claims + p, 8, hipMemcpyDeviceToDevice, c->xstream));
    HIP_TRY(hipMemcpyAsync(r_strag(p), s_strag, (size_t)(1 + L.S) * 8, hipMemcpyDeviceToDevice, c->xstream));
  }
  if(W > 1 || c->self_rccl) {
    // the regions in rounds of at most max_msg_keys * 8 bytes per peer (sizes are the same everywhere: nothing to agree on)
    const size_t per = std::max<size_t>(1, (size_t)c->max_msg_keys * 2);
    for(size_t lo = per; lo < blk; lo += per) {
      NCCL_TRY(ncclGroupStart());
      for(int p = 0; p < W; ++p) {
        if(p == skip) continue;
        const size_t len = std::min(per, blk - lo);
        NCCL_TRY(ncclSend(s_items(p) + lo, len, ncclUint32, p, c->nccl, c->xstream));
        NCCL_TRY(ncclRecv(r_items(p) + lo, len, ncclUint32, p, c->nccl, c->xstream));
      }
      NCCL_TRY(ncclGroupEnd());
    }
    NCCL_TRY(ncclGroupStart());
    for(int p = 0; p < W; ++p) {
      if(p == skip) continue;
      NCCL_TRY(ncclSend(s_items(p), std::min(per, blk), ncclUint32, p, c->nccl, c->xstream));
      NCCL_TRY(ncclRecv(r_items(p), std::min(per, blk), ncclUint32, p, c->nccl, c->xstream));
      NCCL_TRY(ncclSend(s_offs(p), (size_t)2 * L.nbc, ncclUint64, p, c->nccl, c->xstream));
      NCCL_TRY(ncclRecv(r_offs(p), (size_t)2 * L.nbc, ncclUint64, p, c->nccl, c->xstream));
      NCCL_TRY(ncclSend(s_claims + p, 1, ncclUint64, p, c->nccl, c->xstream));
      NCCL_TRY(ncclRecv(r_claims + p, 1, ncclUint64, p, c->nccl, c->xstream));
      NCCL_TRY(ncclSend(s_strag, (size_t)(1 + L.S), ncclUint64, p, c->nccl, c->xstream));
      NCCL_TRY(ncclRecv(r_strag(p), (size_t)(1 + L.S), ncclUint64, p, c->nccl, c->xstream));
    }
    NCCL_TRY(ncclGroupEnd());
  }
  { const int peers = (W > 1 || c->self_rccl) ? W - (skip >= 0 ? 1 : 0) : 0;
    comm_x_end(c, R, cur, (uint64_t)peers * (blk * 4 + (uint64_t)2 * L.nbc * 8 + 8 + (uint64_t)(1 + L.S) * 8)); }
  HIP_TRY(hipEventRecord(R.exchanged[cur], c->xstream));
  R.used[cur] = true;
  return JFGPU_OK;
#endif
}

// The same for the local transport.
int comm_exchange_items_local(jfgpu_comm* c) {
  const int W = c->world;
  const ItemLayout L = item_layout(c, c->ranks[0].t, c->ranks[0].icap[c->ranks[0].turn]);
  for(int d = 0; d < W; ++d) {
    jfgpu_comm::Rank& D = c->ranks[d];
    if(D.used[D.turn]) HIP_TRY(hipEventSynchronize(D.consumed[D.turn]));
    int rc = comm_reserve(D.recv[D.turn], D.recv_cap[D.turn], (L.recv_bytes + 7) / 8, D.t->stream, c->xstream); if(rc) return rc;
  }
  for(int s = 0; s < W; ++s) {
    jfgpu_comm::Rank& S = c->ranks[s];
    HIP_TRY(hipEventRecord(S.routed[S.turn], S.t->stream));
    HIP_TRY(hipStreamWaitEvent(c->xstream, S.routed[S.turn], 0));
  }
  const size_t blk = (size_t)L.nbc * L.cap;
  for(int s = 0; s < W; ++s)
    for(int d = 0; d < W; ++d) {
      const uint8_t* sb = reinterpret_cast<const uint8_t*>(c->ranks[s].send[c->ranks[s].turn]);
      uint8_t* rb = reinterpret_cast<uint8_t*>(c->ranks[d].recv[c->ranks[d].turn]);
      HIP_TRY(hipMemcpyAsync(rb + (size_t)s * blk * 4, sb + (size_t)d * blk * 4, blk * 4, hipMemcpyDeviceToDevice, c->xstream));
      HIP_TRY(hipMemcpyAsync(rb + L.r_offs_at + (size_t)s * 2 * L.nbc * 8, sb + L.offs_at + (size_t)d * 2 * L.nbc * 8, (size_t)2 * L.nbc * 8, hipMemcpyDeviceToDevice, c->xstream));
      HIP_TRY(hipMemcpyAsync(rb + L.r_claims_at + (size_t)s * 8, sb + L.claims_at + (size_t)d * 8, 8, hipMemcpyDeviceToDevice, c->xstream));
      HIP_TRY(hipMemcpyAsync(rb + L.r_strag_at + (size_t)s * (1 + L.S) * 8, sb + L.strag_at, (size_t)(1 + L.S) * 8, hipMemcpyDeviceToDevice, c->xstream));
    }
  for(int d = 0; d < W; ++d) {
    jfgpu_comm::Rank& D = c->ranks[d];
    HIP_TRY(hipEventRecord(D.exchanged[D.turn], c->xstream));
    D.used[D.turn] = true;
  }
  return JFGPU_OK;
}


#if !defined(JFGPU_EMU)
// ---- the "ipc" transport -----------------------------------------------------------------------------------------
constexpr int kIpcMaxWorld = 16;
struct IpcShared {
  std::atomic<uint32_t> arrived, generation, failed, attached;
  uint64_t coll[kIpcMaxWorld][64];                          // a rank's contribution to the running collective
  struct Pub {
    hipIpcMemHandle_t send[2]; uint64_t send_epoch[2];      // send buffers of both turns (epoch changes when one is re-allocated)
    uint64_t scount[kIpcMaxWorld], soff[kIpcMaxWorld + 1];  // key path: what this rank holds for every owner, and where
  } pub[kIpcMaxWorld];
};

int ipc_fail(jfgpu_comm* c, const char* what) {
  if(c->shm) c->shm->failed.store(1);
  return fail(JFGPU_E_HIP, std::string("ipc transport: ") + what);
}

// Sense-reversing barrier over the shared block; gives up when a rank reported a failure or nobody moves for 10 minutes.
int ipc_barrier(jfgpu_comm* c) {
  IpcShared* S = c->shm;
  const uint32_t gen = S->generation.load();
  if(S->arrived.fetch_add(1) + 1 == (uint32_t)c->world) { S->arrived.store(0); S->generation.fetch_add(1); return JFGPU_OK; }
  const auto t0 = std::chrono::steady_clock::now();
  for(uint64_t spins = 0; S->generation.load() == gen; ++spins) {
    if(S->failed.load()) return fail(JFGPU_E_HIP, "ipc transport: another rank failed");
    if(spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50)); else std::this_thread::yield();
    if((spins & 0xFFFF) == 0xFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::minutes(10)) return ipc_fail(c, "barrier timed out");
  }
  return JFGPU_OK;
}

// (Re-)publish this rank's send buffer of turn `cur` if it was re-allocated since.
int ipc_publish_send(jfgpu_comm* c, int cur) {
  jfgpu_comm::Rank& R = c->ranks[0];
  IpcShared::Pub& P = c->shm->pub[c->rank];
  // (pointer AND capacity: comm_reserve frees and allocates, and the allocator likes to hand the same address back for the
  //  larger buffer -- peers would go on copying through their mapping of the freed one.  Round-3 advisor finding.)
  if(R.send[cur] && (c->send_exported[cur] != (void*)R.send[cur] || c->send_exported_cap[cur] != R.send_cap[cur])) {
    if(hipIpcGetMemHandle(&P.send[cur], R.send[cur]) != hipSuccess) return ipc_fail(c, "hipIpcGetMemHandle");
    c->send_exported[cur] = R.send[cur]; c->send_exported_cap[cur] = R.send_cap[cur];
    P.send_epoch[cur] = ++c->send_epoch[cur];
  }
  return JFGPU_OK;
}

// Peer p's send buffer of turn `cur`, mapped into this process.
int ipc_peer_send(jfgpu_comm* c, int p, int cur, uint8_t** out) {
  jfgpu_comm::PeerMap& M = c->peer_send[cur][p];
  const IpcShared::Pub& P = c->shm->pub[p];
  if(M.epoch != P.send_epoch[cur]) {
    if(M.ptr) { if(c->tun.comm_ipc_keep) c->stale_maps.push_back(M.ptr); else hipIpcCloseMemHandle(M.ptr); M.ptr = nullptr; }
    if(hipIpcOpenMemHandle(&M.ptr, P.send[cur], hipIpcMemLazyEnablePeerAccess) != hipSuccess) return ipc_fail(c, "hipIpcOpenMemHandle");
    M.epoch = P.send_epoch[cur];
  }
  *out = reinterpret_cast<uint8_t*>(M.ptr);
  return JFGPU_OK;
}

// Map every peer's send buffer of turn `cur` (only those re-allocated since they were last mapped), ONE RANK AT A TIME.
// Two processes importing each other's allocations at the same moment stalled in hipIpcOpenMemHandle for good with
// buffers of 2.6 GB and more (1.8 GB: fine; one process importing while the other idles: fine at any size,
// tools/probes/r03_ipc_size_probe.hip) -- so the imports take turns, with nothing of this exchange in flight yet.
int ipc_open_peers(jfgpu_comm* c, int cur) {
  for(int turn = 0; turn < c->world; ++turn) {
    int rc = JFGPU_OK;
    if(turn == c->rank)
      for(int p = 0; p < c->world && rc == JFGPU_OK; ++p) { uint8_t* unused; if(p != c->rank) rc = ipc_peer_send(c, p, cur, &unused); }
    if(rc) return rc;
    rc = ipc_barrier(c); if(rc) return rc;
  }
  return JFGPU_OK;
}

// Key path: every rank publishes its per-owner counts and offsets, then PULLS what is meant for it out of the peers'
// send buffers into its own receive buffer.
int comm_exchange_ipc(jfgpu_comm* c) {
  jfgpu_comm::Rank& R = c->ranks[0];
  const int cur = R.turn, W = c->world;
  IPC_TRACE(c, "keys: waiting for the routing kernels (turn %d)", cur);
  HIP_TRY(hipStreamSynchronize(R.t->stream));                // the routed keys are in send[cur]
  int rc = ipc_publish_send(c, cur); if(rc) return rc;
  IPC_TRACE(c, "keys: published, barrier 1");
  IpcShared::Pub& mine = c->shm->pub[c->rank];
  for(int p = 0; p < W; ++p) { mine.scount[p] = R.scount[cur][p]; mine.soff[p] = R.soff[cur][p]; }
  rc = ipc_barrier(c); if(rc) return rc;
  rc = ipc_open_peers(c, cur); if(rc) return rc;
  uint64_t total = 0;
  for(int p = 0; p < W; ++p) { R.rcount[cur][p] = c->shm->pub[p].scount[c->rank]; R.roff[cur][p] = total; total += R.rcount[cur][p]; }
  R.roff[cur][W] = total;
  if(R.used[cur]) HIP_TRY(hipEventSynchronize(R.consumed[cur]));
  rc = comm_reserve(R.recv[cur], R.recv_cap[cur], total, R.t->stream, c->xstream); if(rc) return rc;
  for(int p = 0; p < W; ++p) {
    if(!R.rcount[cur][p]) continue;
    uint8_t* src = reinterpret_cast<uint8_t*>(R.send[cur]);
    if(p != c->rank) { rc = ipc_peer_send(c, p, cur, &src); if(rc) return rc; }
    HIP_TRY(hipMemcpyAsync(R.recv[cur] + R.roff[cur][p], src + c->shm->pub[p].soff[c->rank] * 8, R.rcount[cur][p] * 8, hipMemcpyDeviceToDevice, c->xstream));
  }
  HIP_TRY(hipEventRecord(R.exchanged[cur], c->xstream));
  IPC_TRACE(c, "keys: copies enqueued, waiting for them");
  HIP_TRY(hipStreamSynchronize(c->xstream));
  IPC_TRACE(c, "keys: pulled, barrier 2");
  rc = ipc_barrier(c); if(rc) return rc;                     // everybody has pulled: the send buffers may be written again
  IPC_TRACE(c, "keys: step exchanged");
  comm_free_retired(c, cur);                                 // (every peer has re-imported send[cur]: nobody maps the old ones)
  R.used[cur] = true;
  return JFGPU_OK;
}

// Item path: fixed layout, nothing to publish but the buffers.
int comm_exchange_items_ipc(jfgpu_comm* c) {
  jfgpu_comm::Rank& R = c->ranks[0];
  const int cur = R.turn, W = c->world, me = c->rank;
  const ItemLayout L = item_layout(c, R.t, R.icap[cur]);
  IPC_TRACE(c, "items: waiting for the routing kernels (turn %d, cap %u)", cur, R.icap[cur]);
  HIP_TRY(hipStreamSynchronize(R.t->stream));
  int rc = ipc_publish_send(c, cur); if(rc) return rc;
  IPC_TRACE(c, "items: published; waiting for recv[%d] to be consumed", cur);
  if(R.used[cur]) HIP_TRY(hipEventSynchronize(R.consumed[cur]));
  rc = comm_reserve(R.recv[cur], R.recv_cap[cur], (L.recv_bytes + 7) / 8, R.t->stream, c->xstream); if(rc) return rc;
  IPC_TRACE(c, "items: recv buffer of %zu bytes ready, barrier 1", L.recv_bytes);
  rc = ipc_barrier(c); if(rc) return rc;
  rc = ipc_open_peers(c, cur); if(rc) return rc;
  IPC_TRACE(c, "items: peers' send buffers mapped");
  uint8_t* rb = reinterpret_cast<uint8_t*>(R.recv[cur]);
  const size_t blk = (size_t)L.nbc * L.cap;
  for(int p = 0; p < W; ++p) {                               // sender p's share for me
    uint8_t* sb = reinterpret_cast<uint8_t*>(R.send[cur]);
    if(p != me) { rc = ipc_peer_send(c, p, cur, &sb); if(rc) return rc; }
    IPC_TRACE(c, "items: rank %d's send buffer is at %p here", p, (void*)sb);
    HIP_TRY(hipMemcpyAsync(rb + (size_t)p * blk * 4, sb + (size_t)me * blk * 4, blk * 4, hipMemcpyDeviceToDevice, c->xstream));
    IPC_TRACE(c, "items: regions copy enqueued (%zu bytes)", blk * 4);
    HIP_TRY(hipMemcpyAsync(rb + L.r_offs_at + (size_t)p * 2 * L.nbc * 8, sb + L.offs_at + (size_t)me * 2 * L.nbc * 8, (size_t)2 * L.nbc * 8, hipMemcpyDeviceToDevice, c->xstream));
    IPC_TRACE(c, "items: offsets copy enqueued");
    HIP_TRY(hipMemcpyAsync(rb + L.r_claims_at + (size_t)p * 8, sb + L.claims_at + (size_t)me * 8, 8, hipMemcpyDeviceToDevice, c->xstream));
    IPC_TRACE(c, "items: claims copy enqueued");
    HIP_TRY(hipMemcpyAsync(rb + L.r_strag_at + (size_t)p * (1 + L.S) * 8, sb + L.strag_at, (size_t)(1 + L.S) * 8, hipMemcpyDeviceToDevice, c->xstream));
    IPC_TRACE(c, "items: straggler list copy enqueued");
  }
  HIP_TRY(hipEventRecord(R.exchanged[cur], c->xstream));
  IPC_TRACE(c, "items: copies enqueued, waiting for them");
  HIP_TRY(hipStreamSynchronize(c->xstream));
  IPC_TRACE(c, "items: pulled, barrier 2");
  rc = ipc_barrier(c); if(rc) return rc;
  IPC_TRACE(c, "items: step exchanged");
  comm_free_retired(c, cur);
  R.used[cur] = true;
  return JFGPU_OK;
}

int ipc_collective(jfgpu_comm* c, uint64_t* values, int n, int op /* 0 sum, 1 max, 2 gather of values[0] into values[0..world) */) {
  IpcShared* S = c->shm;
  for(int i = 0; i < (op == 2 ? 1 : n); ++i) S->coll[c->rank][i] = values[i];
  int rc = ipc_barrier(c); if(rc) return rc;
  if(op == 2) { for(int p = 0; p < c->world; ++p) values[p] = S->coll[p][0]; }
  else for(int i = 0; i < n; ++i) {
    uint64_t v = op == 0 ? 0 : S->coll[0][i];
    for(int p = 0; p < c->world; ++p) v = op == 0 ? v + S->coll[p][i] : std::max(v, S->coll[p][i]);
    values[i] = v;
  }
  return ipc_barrier(c);                                     // (the slots are free again)
}

int ipc_attach(jfgpu_comm* c, const uint8_t* id128) {
  if(c->world > kIpcMaxWorld) return fail(JFGPU_E_INVALID, "ipc transport: at most 16 ranks");
  c->shm_name.assign(reinterpret_cast<const char*>(id128) + 8);
  const int fd = shm_open(c->shm_name.c_str(), O_CREAT | O_RDWR, 0600);       // created zero-filled by whoever comes first
  if(fd < 0) return fail(JFGPU_E_HIP, "ipc transport: shm_open failed");
  if(ftruncate(fd, sizeof(IpcShared)) != 0) { close(fd); return fail(JFGPU_E_HIP, "ipc transport: ftruncate failed"); }
  void* m = mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if(m == MAP_FAILED) return fail(JFGPU_E_HIP, "ipc transport: mmap failed");
  c->shm = reinterpret_cast<IpcShared*>(m);
  c->ipc = true;
  for(int i = 0; i < 2; ++i) c->peer_send[i].assign(c->world, jfgpu_comm::PeerMap());
  c->shm->attached.fetch_add(1);
  return ipc_barrier(c);                                     // everybody is here
}

void ipc_detach(jfgpu_comm* c) {
  if(!c->shm) return;
  for(int i = 0; i < 2; ++i) for(auto& M : c->peer_send[i]) if(M.ptr) hipIpcCloseMemHandle(M.ptr);
  for(void* p : c->stale_maps) hipIpcCloseMemHandle(p);
  c->stale_maps.clear();
  const bool last = c->shm->attached.fetch_sub(1) == 1;
  munmap(c->shm, sizeof(IpcShared));
  if(last) shm_unlink(c->shm_name.c_str());
  c->shm = nullptr;
}
#else
int comm_exchange_ipc(jfgpu_comm*) { return fail(JFGPU_E_UNSUPPORTED, "no inter-process transport in the emulated build"); }
int comm_exchange_items_ipc(jfgpu_comm*) { return fail(JFGPU_E_UNSUPPORTED, "no inter-process transport in the emulated build"); }
#endif

// ---- the shards of a table grow together -------------------------------------------------------------------------------
// hash_counter::double_size (hash_counter.hpp:200-238) for a table spread over ranks.  The doubled table has a new matrix
// (the next one of the reference's random() stream, the same on every rank), so an entry's owner changes: every rank
// walks its old shard, puts what stays with it into its new shard and sends the rest -- (key, count) pairs, grouped by
// new owner -- through the key path's exchange, twice (keys, then counts, same grouping).  One-word keys.
//   pass 0: how many pairs for every owner;  pass 1: the pairs, at the cursors (= offsets), and the local inserts
__global__ __launch_bounds__(kBlock) void reshard_kernel(DevTable old, DevTable neu, int have_ovf, int pass, unsigned long long* __restrict__ cursors,
                                                         uint64_t* __restrict__ keys_out, uint64_t* __restrict__ cnts_out) {
  const uint64_t n = 1ull << old.g.lsize_l;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t w = slot_ld(old, i);
    if(!w) continue;
    const uint64_t key = slot_key(old.g, old.inv_tbl, w, i & ~old.g.tile_mask);
    const uint32_t owner = (uint32_t)(hash_tables(neu.fwd_tbl, key, neu.g.nbytes) >> neu.g.lsize_l);
    if(owner == neu.g.shard_id) { if(pass) table_add_val(neu, neu.fwd_tbl, key, full_count(old, w, i, have_ovf)); }
    else {
      const unsigned long long at = atomicAdd(&cursors[owner], 1ull);
      if(pass) { keys_out[at] = key; cnts_out[at] = full_count(old, w, i, have_ovf); }
    }
  }
}
__global__ __launch_bounds__(kBlock) void add_pairs_kernel(DevTable T, const uint64_t* __restrict__ keys, const uint64_t* __restrict__ cnts, uint64_t n) {
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    table_add_val(T, T.fwd_tbl, keys[i], cnts[i]);
}

// The same for two-word keys: a pair is two key words (low, high) and a count.
__global__ __launch_bounds__(kBlock) void reshard_wide_kernel(WideTable old, WideTable neu, int have_ovf, int pass, unsigned long long* __restrict__ cursors,
                                                              uint64_t* __restrict__ keys_out, uint64_t* __restrict__ cnts_out) {
  const TableGeom& g = old.W.g;
  const DevTable od = ovf_view(old);
  const uint64_t n = 1ull << g.lsize_l;
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t hi = old.slots[2 * i + 1];
    if(!hi) continue;
    const uint64_t lo = old.slots[2 * i];
    if(!lo) continue;                                    // hi claimed, never completed: holds no key
    const u128 key = wide_slot_key(old, old.inv_tbl, lo, hi, i & ~g.tile_mask);
    uint64_t cnt = slot_count(g, hi);
    if(have_ovf) cnt += ovf_get(od, i) << g.cnt_bits;
    const uint32_t owner = slot_addr(neu.W.g, hash_tables_wide(neu.fwd_tbl, key, neu.W.g.nbytes)).shard;
    if(owner == neu.W.g.shard_id) { if(pass) wide_add_val(neu, neu.fwd_tbl, key, cnt); }
    else {
      const unsigned long long at = atomicAdd(&cursors[owner], 1ull);
      if(pass) { keys_out[2 * at] = (uint64_t)key; keys_out[2 * at + 1] = (uint64_t)(key >> 64); cnts_out[at] = cnt; }
    }
  }
}
__global__ __launch_bounds__(kBlock) void add_pairs_wide_kernel(WideTable T, const uint64_t* __restrict__ keys, const uint64_t* __restrict__ cnts, uint64_t n) {
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    wide_add_val(T, T.fwd_tbl, ((u128)keys[2 * i + 1] << 64) | keys[2 * i], cnts[i]);
}

int comm_exchange_rccl(jfgpu_comm* c); int comm_exchange_local(jfgpu_comm* c); int comm_exchange_ipc(jfgpu_comm* c);

// Collective (every rank of the communicator; the local transport: all its ranks here).
int comm_grow(jfgpu_comm* c) {
  const int W = c->world;
  std::vector<GrowNew> N(c->ranks.size());
  std::vector<char> prepared(c->ranks.size(), 0);
  bool cannot = false;
  int hard = 0; std::string hard_msg;                       // a rank's own failure (rc > 0): agreed on like "cannot", then returned by everybody
  // every rank: nothing in flight, the new shard allocated, the pairs that leave grouped by their new owner in send[0] / send[1]
  for(size_t q = 0; q < c->ranks.size() && !hard; ++q) {
    jfgpu_comm::Rank& R = c->ranks[q];
    jfgpu_table* t = R.t;
    if(t->nword) return fail(JFGPU_E_UNSUPPORTED, "sharded tables of keys longer than two words are not built");      // (the same on every rank)
    int rc = comm_insert_prev(c, R);
    if(rc == 0) rc = grow_prepare(t, N[q]);
    if(rc > 0) { hard = rc; hard_msg = g_err; break; }
    prepared[q] = rc == 0;                                  // < 0: no memory for the doubled shard (nothing allocated)
    cannot = cannot || rc < 0;
  }
  // Growing is collective: a rank that cannot allocate its doubled shard must not leave alone while the others go on into
  // the exchange (round-4 advisor finding: they hung there).  One more agreement; if anybody cannot, nobody grows -- what
  // was prepared is freed and the shards carry on as they are, growth off, like a single table short of memory
  // (ensure_capacity): a shard that then really fills up reports "Hash full" through its tiles' probe bound.  A rank whose
  // preparation FAILED (a HIP error, a deferred "Hash full") takes part in the agreement too and everybody returns an error
  // (round-5 advisor finding: it used to return before the all-reduce, the peers blocked in it).
  auto free_prepared = [&]() {
    for(size_t q = 0; q < c->ranks.size(); ++q)
      if(prepared[q]) { DevTable& nd = N[q].nd; hipFree(nd.slots); hipFree(nd.ovf_key); hipFree(nd.ovf_cnt); hipFree(nd.dirty); hipFree(N[q].nf); hipFree(N[q].ni); prepared[q] = 0; }
  };
  uint64_t agreed = hard ? 2 : cannot ? 1 : 0;
  if(!c->local) { int rc = jfgpu_comm_allreduce_u64(c, &agreed, 1, 1); if(rc) { free_prepared(); return rc; } }
  if(agreed >= 2) {
    free_prepared();
    return hard ? fail(hard, hard_msg) : fail(JFGPU_E_HIP, "another rank failed while the shards were growing");
  }
  cannot = agreed != 0;
  if(cannot) {
    free_prepared();
    for(size_t q = 0; q < c->ranks.size(); ++q) {
      c->ranks[q].t->grow_on = false;
    }
    return JFGPU_OK;
  }
  for(size_t q = 0; q < c->ranks.size(); ++q) {
    jfgpu_comm::Rank& R = c->ranks[q];
    jfgpu_table* t = R.t;
    const uint64_t kw = t->wide ? 2 : 1;                   // 64-bit words per key
    int rc = 0;
    HIP_TRY(hipStreamSynchronize(c->xstream));
    const int have_ovf = (int)(N[q].ctr[CTR_OVF_USED] != 0);
    const dim3 grid(grid_for(t, (1ull << t->g.lsize_l) / kBlock + 1)), block(kBlock);
    HIP_TRY(hipMemsetAsync(R.d_cnt, 0, sizeof(unsigned long long) * W, t->stream));
    if(t->wide) hipLaunchKernelGGL(reshard_wide_kernel, grid, block, 0, t->stream, t->wt, N[q].nw, have_ovf, 0, R.d_cnt, (uint64_t*)nullptr, (uint64_t*)nullptr);
    else hipLaunchKernelGGL(reshard_kernel, grid, block, 0, t->stream, t->dt, N[q].nd, have_ovf, 0, R.d_cnt, (uint64_t*)nullptr, (uint64_t*)nullptr);
    std::vector<unsigned long long> h(W);
    HIP_TRY(hipMemcpyAsync(h.data(), R.d_cnt, sizeof(unsigned long long) * W, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    if(c->tun.comm_trace) {
      uint64_t ms = 0; for(uint64_t col : N[q].m2.columns) ms = ms * 0x9E3779B97F4A7C15ull + col;
      std::string hs; for(int p = 0; p < W; ++p) hs += " " + std::to_string(h[p]);
      IPC_TRACE(c, "grow: shard %u, new r %u (xs %d), matrix mix %016llx, pairs leaving per owner:%s", t->g.shard_id, N[q].m2.r, (int)N[q].g2.hash_xs, (unsigned long long)ms, hs.c_str());
    }
    uint64_t total = 0;
    // turn 0 carries the keys (kw words each), turn 1 the counts: the same groups, in the same order
    for(int p = 0; p < W; ++p) { R.scount[1][p] = h[p]; R.soff[1][p] = total; R.scount[0][p] = h[p] * kw; R.soff[0][p] = total * kw; total += h[p]; h[p] = R.soff[1][p]; }
    R.soff[1][W] = total; R.soff[0][W] = total * kw;
    rc = comm_reserve_send(c, R, 0, std::max<uint64_t>(total * kw, 1), t->stream); if(rc) return rc;
    rc = comm_reserve_send(c, R, 1, std::max<uint64_t>(total, 1), t->stream); if(rc) return rc;
    HIP_TRY(hipMemcpyAsync(R.d_cnt, h.data(), sizeof(unsigned long long) * W, hipMemcpyHostToDevice, t->stream));   // cursors = offsets
    if(t->wide) hipLaunchKernelGGL(reshard_wide_kernel, grid, block, 0, t->stream, t->wt, N[q].nw, have_ovf, 1, R.d_cnt, R.send[0], R.send[1]);
    else hipLaunchKernelGGL(reshard_kernel, grid, block, 0, t->stream, t->dt, N[q].nd, have_ovf, 1, R.d_cnt, R.send[0], R.send[1]);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));
    R.used[0] = R.used[1] = false;                            // (everything before is complete: the exchanges below start from a clean slate)
    R.icap[0] = R.icap[1] = 0;
  }
  // keys (turn 0), then counts (turn 1): the key path's exchange, as it is
  for(int turn = 0; turn < 2; ++turn) {
    for(auto& R : c->ranks) R.turn = turn;
    int rc = c->local ? comm_exchange_local(c) : c->ipc ? comm_exchange_ipc(c) : comm_exchange_rccl(c);
    if(rc) return rc;
  }
  for(size_t q = 0; q < c->ranks.size(); ++q) {
    jfgpu_comm::Rank& R = c->ranks[q];
    jfgpu_table* t = R.t;
    HIP_TRY(hipStreamWaitEvent(t->stream, R.exchanged[0], 0));
    HIP_TRY(hipStreamWaitEvent(t->stream, R.exchanged[1], 0));
    const uint64_t n = R.roff[1][W];                        // pairs that arrived (turn 1's offsets count them; turn 0's count key words)
    if(c->tun.comm_trace && !t->wide) {                     // where do the arrivals belong under the new matrix?  (host check, trace runs only)
      HIP_TRY(hipStreamSynchronize(c->xstream));
      std::vector<uint64_t> hk(n);
      if(n) HIP_TRY(hipMemcpy(hk.data(), R.recv[0], n * 8, hipMemcpyDeviceToHost));
      std::string hs;
      for(int p = 0; p < W; ++p) {
        uint64_t bad = 0;
        for(uint64_t i = R.roff[0][p]; i < R.roff[0][p + 1]; ++i) {
          uint64_t pos = 0;
          for(uint32_t j = 0; j < N[q].m2.c; ++j) if((hk[i] >> j) & 1) pos ^= N[q].m2.columns[N[q].m2.c - 1 - j];
          if((pos >> N[q].g2.lsize_l) != t->g.shard_id) ++bad;
        }
        hs += " " + std::to_string(R.roff[0][p + 1] - R.roff[0][p]) + "/" + std::to_string(bad);
      }
      IPC_TRACE(c, "grow: shard %u, arrived/not mine per sender:%s", t->g.shard_id, hs.c_str());
    }
    if(n && t->wide) hipLaunchKernelGGL(add_pairs_wide_kernel, dim3(grid_for(t, n / kBlock + 1)), dim3(kBlock), 0, t->stream, N[q].nw, (const uint64_t*)R.recv[0], (const uint64_t*)R.recv[1], n);
    else if(n) hipLaunchKernelGGL(add_pairs_kernel, dim3(grid_for(t, n / kBlock + 1)), dim3(kBlock), 0, t->stream, N[q].nd, (const uint64_t*)R.recv[0], (const uint64_t*)R.recv[1], n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(R.consumed[0], t->stream));
    HIP_TRY(hipEventRecord(R.consumed[1], t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    int rc = grow_swap(t, N[q]); if(rc) return rc;
    R.turn = 0; R.inflight = false;
    rc = measure_occupancy(t); if(rc) return rc;
  }
  return JFGPU_OK;
}

extern "C" int jfgpu_comm_allreduce_u64(jfgpu_comm* c, uint64_t* values, int n, int op);

// ---- `jellyfish bc` over several GPUs: the ranks' counters merged ----------------------------------------------------------
// Every rank inserts ITS part of the input into its own Bloom counter (same size, same matrices).  A cell of the counter
// of the whole input is min(2, sum of the ranks' cells): the increments commute and saturate at 2 (bloom_counter2.hpp:56-107),
// so the merged array is byte for byte the array one counter fed with everything holds -- what `bc` writes, what
// `count --bc` asks.  The merge is two rounds of the key path's exchange (comm_exchange_*: an all-to-all of 64-bit words,
// whatever the transport): the array is cut into W ranges of words; round one sends range p of every rank to rank p, which
// adds them digit by digit (reduce-scatter); round two sends the merged range to everybody (all-gather).  Every rank ends
// up with the whole merged counter -- the filtered count over shards wants exactly that (every rank asks before it routes).
__device__ __forceinline__ uint32_t bloom_byte_add(uint32_t a, uint32_t b) {     // five base-3 digits a byte, digit-wise min(2, a + b)
  uint32_t r = 0, w = 1;
#pragma unroll
  for(int d = 0; d < 5; ++d) { const uint32_t s = a % 3u + b % 3u; r += (s > 2u ? 2u : s) * w; w *= 3u; a /= 3u; b /= 3u; }
  return r;
}
__global__ __launch_bounds__(kBlock) void bloom_merge_kernel(uint64_t* __restrict__ dst, const uint64_t* __restrict__ src, uint64_t n_words) {
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t a = dst[i], b = src[i];
    if(b == 0) continue;
    uint64_t r = 0;
#pragma unroll
    for(int k = 0; k < 8; ++k) r |= (uint64_t)bloom_byte_add((uint32_t)(a >> (8 * k)) & 0xFFu, (uint32_t)(b >> (8 * k)) & 0xFFu) << (8 * k);
    dst[i] = r;
  }
}

int comm_exchange_rccl(jfgpu_comm* c); int comm_exchange_local(jfgpu_comm* c); int comm_exchange_ipc(jfgpu_comm* c);

// blooms[q]: the counter of local rank q (RCCL / ipc transport: one; local transport: all W).  Collective.
int comm_bc_merge(jfgpu_comm* c, jfgpu_bloom** blooms) {
  const int W = c->world;
  const size_t nr = c->ranks.size();
  jfgpu_bloom* b0 = blooms[0];
  const uint64_t words = (b0->data_bytes + 7) / 8;
  const uint64_t per = (words + (uint64_t)W - 1) / (uint64_t)W;
  auto r_lo = [&](int p) { return std::min<uint64_t>(words, (uint64_t)p * per); };
  auto r_len = [&](int p) { return std::min<uint64_t>(words, (uint64_t)(p + 1) * per) - r_lo(p); };
  // the exchange keeps its order on the rank's table stream: a stand-in that carries the counter's stream
  std::vector<std::unique_ptr<jfgpu_table>> shim(nr);
  std::vector<jfgpu_table*> saved(nr);
  int rc = JFGPU_OK;
  for(size_t q = 0; q < nr; ++q) { saved[q] = c->ranks[q].t; }
  for(size_t q = 0; q < nr && !rc; ++q) {
    jfgpu_bloom* b = blooms[q];
    if(!b || b->kind != 0) { rc = fail(JFGPU_E_INVALID, "bc merge: a Bloom counter per rank"); break; }
    if(b->data_bytes != b0->data_bytes || b->nh != b0->nh || b->m != b0->m) { rc = fail(JFGPU_E_INVALID, "bc merge: the ranks' counters differ in size"); break; }
    if(b->alloc_bytes < words * 8) { rc = fail(JFGPU_E_INVALID, "bc merge: counter allocation shorter than its last word"); break; }
    if(c->ranks[q].inflight) { rc = fail(JFGPU_E_INVALID, "bc merge: a count step is still in flight on this communicator"); break; }
    rc = bloom_flush(b); if(rc) break;
    if(hipStreamSynchronize(b->stream) != hipSuccess) { rc = fail(JFGPU_E_HIP, "bc merge: sync"); break; }
    shim[q].reset(new jfgpu_table); shim[q]->stream = b->stream; shim[q]->device = b->device;
    c->ranks[q].t = shim[q].get();
  }
  auto restore = [&]() { for(size_t q = 0; q < nr; ++q) c->ranks[q].t = saved[q]; };
  auto exchange = [&]() { return c->local ? comm_exchange_local(c) : c->ipc ? comm_exchange_ipc(c) : comm_exchange_rccl(c); };
  auto rank_of = [&](size_t q) { return c->local ? (int)q : c->rank; };
  // Before every exchange the ranks agree that all of them got this far (round-5 advisor finding: a rank with a local error
  // skipped the exchange and left the others waiting in it): a failure anywhere ends the merge on every rank.
  auto all_fine = [&]() -> int {
    std::string msg = g_err;
    uint64_t bad = rc ? 1 : 0;
    if(!c->local) { const int rc2 = jfgpu_comm_allreduce_u64(c, &bad, 1, 1); if(rc2) return rc2; }
    if(!bad) return JFGPU_OK;
    return rc ? fail(rc, msg) : fail(JFGPU_E_HIP, "bc merge: another rank failed");
  };
  // round one: range p of my array goes to rank p
  for(size_t q = 0; q < nr && !rc; ++q) {
    jfgpu_comm::Rank& R = c->ranks[q]; jfgpu_bloom* b = blooms[q];
    R.turn = 0;
    rc = comm_reserve_send(c, R, 0, std::max<uint64_t>(words, 1), b->stream); if(rc) break;
    if(hipMemcpyAsync(R.send[0], b->d_data, words * 8, hipMemcpyDeviceToDevice, b->stream) != hipSuccess) { rc = fail(JFGPU_E_HIP, "bc merge: copy"); break; }
    for(int p = 0; p < W; ++p) { R.scount[0][p] = r_len(p); R.soff[0][p] = r_lo(p); }
    R.soff[0][W] = words;
  }
  rc = all_fine(); if(rc) { restore(); return rc; }
  rc = exchange();
  for(size_t q = 0; q < nr && !rc; ++q) {
    jfgpu_comm::Rank& R = c->ranks[q]; jfgpu_bloom* b = blooms[q];
    const int me = rank_of(q);
    const uint64_t n = r_len(me);
    uint64_t* mine = reinterpret_cast<uint64_t*>(b->d_data) + r_lo(me);
    if(hipStreamWaitEvent(b->stream, R.exchanged[0], 0) != hipSuccess) { rc = fail(JFGPU_E_HIP, "bc merge: event"); break; }
    if(n) {
      (void)hipMemsetAsync(mine, 0, n * 8, b->stream);
      const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t)b->n_cu * 8));
      for(int p = 0; p < W; ++p) hipLaunchKernelGGL(bloom_merge_kernel, dim3(grid), dim3(kBlock), 0, b->stream, mine, (const uint64_t*)R.recv[0] + R.roff[0][p], n);
    }
    (void)hipEventRecord(R.consumed[0], b->stream);
    // round two: my merged range to everybody
    R.turn = 1;
    rc = comm_reserve_send(c, R, 1, std::max<uint64_t>(n, 1), b->stream); if(rc) break;
    if(n && hipMemcpyAsync(R.send[1], mine, n * 8, hipMemcpyDeviceToDevice, b->stream) != hipSuccess) { rc = fail(JFGPU_E_HIP, "bc merge: copy"); break; }
    for(int p = 0; p < W; ++p) { R.scount[1][p] = n; R.soff[1][p] = 0; }
    R.soff[1][W] = n;
  }
  rc = all_fine(); if(rc) { restore(); return rc; }
  rc = exchange();
  unsigned long long mers_sum = 0;
  for(size_t q = 0; q < nr && !rc; ++q) {
    jfgpu_comm::Rank& R = c->ranks[q]; jfgpu_bloom* b = blooms[q];
    if(hipStreamWaitEvent(b->stream, R.exchanged[1], 0) != hipSuccess) { rc = fail(JFGPU_E_HIP, "bc merge: event"); break; }
    for(int p = 0; p < W; ++p)
      if(R.rcount[1][p]) (void)hipMemcpyAsync(reinterpret_cast<uint64_t*>(b->d_data) + r_lo(p), R.recv[1] + R.roff[1][p], R.rcount[1][p] * 8, hipMemcpyDeviceToDevice, b->stream);
    (void)hipEventRecord(R.consumed[1], b->stream);
    unsigned long long m = 0;
    if(hipMemcpyAsync(&m, b->d_mers, sizeof m, hipMemcpyDeviceToHost, b->stream) != hipSuccess || hipStreamSynchronize(b->stream) != hipSuccess) { rc = fail(JFGPU_E_HIP, "bc merge: sync"); break; }
    mers_sum += m;
    R.turn = 0;
  }
  restore();
  if(rc) return rc;
  // the k-mer tally of the whole input (jfgpu_bc_sync reports it)
  if(!c->local) { uint64_t v = mers_sum; rc = jfgpu_comm_allreduce_u64(c, &v, 1, 0); if(rc) return rc; mers_sum = v; }
  for(size_t q = 0; q < nr; ++q) HIP_TRY(hipMemcpy(blooms[q]->d_mers, &mers_sum, sizeof mers_sum, hipMemcpyHostToDevice));
  return JFGPU_OK;
}

void comm_free_rank(jfgpu_comm::Rank& R) {
  for(int i = 0; i < 2; ++i) {
    if(R.send[i]) hipFree(R.send[i]);
    if(R.recv[i]) hipFree(R.recv[i]);
    if(R.routed[i]) hipEventDestroy(R.routed[i]);
    if(R.exchanged[i]) hipEventDestroy(R.exchanged[i]);
    if(R.consumed[i]) hipEventDestroy(R.consumed[i]);
    if(R.x_begin[i]) hipEventDestroy(R.x_begin[i]);
    if(R.x_end[i]) hipEventDestroy(R.x_end[i]);
  }
  if(R.d_cnt) hipFree(R.d_cnt);
  if(R.d_xc) hipFree(R.d_xc);
  if(R.d_gcur) hipFree(R.d_gcur);
  if(R.d_claimed) hipFree(R.d_claimed);
  if(R.d_arrived) hipFree(R.d_arrived);
  if(R.d_route) hipFree(R.d_route);
  if(R.h_route) hipHostFree(R.h_route);
  if(R.route_done) hipEventDestroy(R.route_done);
}

extern "C" int jfgpu_comm_allreduce_u64(jfgpu_comm* c, uint64_t* values, int n, int op);

// The size question of a step.  The size given at creation is a hint (doc/Readme.md:67-72) for a sharded table too: a
// shard keeps an upper bound of its occupancy (what it measured last + the largest piece any rank fed in every exchange
// since: k-mers <= bytes, and a hash prefix gets its even share of them), input is fed in pieces that fit the head-room the
// ranks have between them (the smallest), and when a rank's head-room runs out every rank measures its shard; if one is
// more than half full, all of them double together (comm_grow).
bool comm_growing(const jfgpu_table* t) { return t->grow_on && !t->nword && t->g.lsize_g < t->g.key_bits; }
uint64_t comm_headroom(const jfgpu_table* t) {
  const uint64_t limit = capacity_limit(t), used = t->occ_known + t->fed_since;
  return limit > used ? limit - used : 0;
}
// What a shard is charged for an exchange in which every rank fed at most `piece` bytes: a hash prefix gets its even share
// of the W pieces only on average -- a sixteenth and four standard deviations on top, so that a shard that receives a
// little more than its share cannot pass its load limit between two measurements (round-4 advisor finding).
uint64_t comm_charge(uint64_t piece) { return piece + piece / 16 + 4 * (uint64_t)std::sqrt((double)piece); }
// head-room too small for a useful piece of `left` bytes: time to measure
bool comm_bound_out(const jfgpu_table* t, uint64_t left) { return comm_headroom(t) < std::min<uint64_t>(left, std::max<uint64_t>(capacity_limit(t) / 8, 1)); }

// One piece of this rank's step through routing, exchange and the insert of the previous piece (cap: the agreed region
// capacity of the item path, 0: keys).
int comm_piece_rccl(jfgpu_comm* c, jfgpu_comm::Rank& R, const char* d_bases, size_t n, uint32_t cap) {
  jfgpu_table* t = R.t;
  int rc = ensure_ovf(t, (uint64_t)n * (uint64_t)c->world, 0); if(rc) return rc;      // (what may arrive for this shard: about a world's worth of one rank's input)
  uint64_t routed = 0;
  // the routing of this piece is enqueued, then the insert of what arrived for the previous one (it only waits for that
  // piece's exchange, on the device): the host's look at the routing pass and the agreement below happen beside device work
  if(cap) { rc = comm_route_items_enqueue(c, R, d_bases, n, cap); if(rc) return rc; }
  rc = comm_insert_prev(c, R); if(rc) return rc;
  if(cap) {
    bool overflow = false;
    rc = comm_route_items_complete(c, R, cap, &overflow, &routed); if(rc) return rc;
    IPC_TRACE(c, "step: routed %llu items%s", (unsigned long long)routed, overflow ? " (straggler list overflow)" : "");
    uint64_t o = overflow ? 1 : 0;
    rc = jfgpu_comm_allreduce_u64(c, &o, 1, 1); if(rc) return rc;
    if(o) cap = 0;                                           // somebody has more stragglers than the list holds: this piece goes as keys
  }
  R.icap[R.turn] = cap;
  if(cap) { R.sent += routed; rc = c->ipc ? comm_exchange_items_ipc(c) : comm_exchange_items_rccl(c); if(rc) return rc; }
  else {
#if !defined(JFGPU_EMU)
    // (the one-box test transport: importing a peer's send buffer of 10 GB never returned -- bench.py --gpus 2 --steps 4 on one
    //  device, a 1.26 GB step as keys; 5 GB buffers are fine -- so such a step fails here, loudly and on every rank, instead
    //  of standing in a barrier for ten minutes)
    if(c->ipc && (uint64_t)n * (t->wide ? 16 : 8) > ((uint64_t)6 << 30))
      return ipc_fail(c, "a step of this size travels as keys in send buffers of more than 6 GiB, which this runtime does not import: feed smaller steps");
#endif
    rc = comm_route(c, R, d_bases, n); if(rc) return rc;
    rc = c->ipc ? comm_exchange_ipc(c) : comm_exchange_rccl(c); if(rc) return rc;
  }
  R.inflight = true; R.turn ^= 1;
  return JFGPU_OK;
}

}  // namespace

extern "C" {

int jfgpu_comm_unique_id(uint8_t* id128) {
  if(!id128) return fail(JFGPU_E_INVALID, "null id");
#if defined(JFGPU_EMU)
  memset(id128, 0, 128);
  return JFGPU_OK;
#else
  if(Tuning::from_env().comm_ipc) {
    // "JFGPUIPC" + the name of the shared block the ranks meet in
    memset(id128, 0, 128);
    memcpy(id128, "JFGPUIPC", 8);
    snprintf(reinterpret_cast<char*>(id128) + 8, 100, "/jfgpu_ipc_%ld_%llx", (long)getpid(),
             (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return JFGPU_OK;
  }
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128, &id, 128);
  return JFGPU_OK;
#endif
}

int jfgpu_comm_create(int world, int rank, const uint8_t* id128, int device, jfgpu_comm** out) {
  if(!out || !id128) return fail(JFGPU_E_INVALID, "null argument");
  *out = nullptr;
  if(world < 1 || (world & (world - 1)) || world > 256 || rank < 0 || rank >= world) return fail(JFGPU_E_INVALID, "world size must be a power of two, rank inside it");
#if defined(JFGPU_EMU)
  (void)device;
  return fail(JFGPU_E_UNSUPPORTED, "no RCCL in the emulated build: use jfgpu_comm_create_local");
#else
  if(device < 0) HIP_TRY(hipGetDevice(&device));
  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<jfgpu_comm> c(new jfgpu_comm);
  c->world = world; c->rank = rank; c->device = device; c->local = false;
  if(!memcmp(id128, "JFGPUIPC", 8)) { int rc = ipc_attach(c.get(), id128); if(rc) return rc; }
  else {
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    NCCL_TRY(ncclCommInitRank(&c->nccl, world, id, rank));
  }
  HIP_TRY(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
  c->ranks.resize(1);
  int rc = comm_init_rank(c.get(), c->ranks[0]); if(rc) return rc;
  comm_apply_tuning(c.get());
  if(c->tun.comm_self_rccl >= 0) c->self_rccl = c->tun.comm_self_rccl != 0;
  *out = c.release();
  return JFGPU_OK;
#endif
}

int jfgpu_comm_create_local(int world, int device, jfgpu_comm** out) {
  if(!out) return fail(JFGPU_E_INVALID, "null argument");
  *out = nullptr;
  if(world < 1 || (world & (world - 1)) || world > 256) return fail(JFGPU_E_INVALID, "world size must be a power of two");
  if(device < 0) HIP_TRY(hipGetDevice(&device));
  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<jfgpu_comm> c(new jfgpu_comm);
  c->world = world; c->rank = -1; c->device = device; c->local = true;
  HIP_TRY(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
  c->ranks.resize(world);
  for(auto& R : c->ranks) { int rc = comm_init_rank(c.get(), R); if(rc) return rc; }
  comm_apply_tuning(c.get());
  *out = c.release();
  return JFGPU_OK;
}

void jfgpu_comm_destroy(jfgpu_comm* c) {
  if(!c) return;
  hipSetDevice(c->device);
  if(c->xstream) hipStreamSynchronize(c->xstream);
  for(auto& R : c->ranks) { if(R.t && R.t->stream) hipStreamSynchronize(R.t->stream); comm_free_rank(R); }
  if(c->d_coll) hipFree(c->d_coll);
#if !defined(JFGPU_EMU)
  if(c->nccl) ncclCommDestroy(c->nccl);
  ipc_detach(c);
#endif
  comm_free_retired(c, 0, true); comm_free_retired(c, 1, true);
  if(c->xstream) hipStreamDestroy(c->xstream);
  delete c;
}

// One step of this rank (RCCL transport; collective: every rank calls it the same number of times, n may be 0).
int jfgpu_comm_count_ascii_dev(jfgpu_comm* c, jfgpu_table* t, const char* d_bases, size_t n) {
  if(!c || c->local) return fail(JFGPU_E_INVALID, "not an RCCL communicator");
  int rc = use(t); if(rc) return rc;
  if((int)t->g.shard_id != c->rank) return fail(JFGPU_E_INVALID, "table shard_id is not this communicator's rank");
  jfgpu_comm::Rank& R = c->ranks[0];
  R.t = t;
  const uint64_t k = t->g.k;
  size_t off = 0;
  for(bool first = true;; first = false) {
    const size_t left = n - off;
    const bool growing = comm_growing(t);
    // item path or keys?  every rank says what region capacity it wants (0: keys); one "keys" decides for all (a rank that
    // has run out of input has no preference: it must not push the others onto the key path).  The same agreement carries
    // what is left anywhere, whether somebody's head-room has run out, and the smallest head-room.
    const uint64_t head = growing ? comm_headroom(t) : ~0ull >> 1;
    const size_t mine = (size_t)std::min<uint64_t>(left, head);
    const uint32_t want = items_cap_wanted(c, R, mine);
    uint64_t v[5] = {(!items_geometry_ok(c, t) || (want == 0 && mine >= k)) ? 1ull : 0ull, want, (uint64_t)left,
                     growing && comm_bound_out(t, left) ? 1ull : 0ull, ~0ull - head};
    IPC_TRACE(c, "step: %zu bytes left, head-room %llu, wants cap %u", left, (unsigned long long)head, want);
    rc = jfgpu_comm_allreduce_u64(c, v, 5, 1); if(rc) return rc;
    if(!first && v[2] == 0) break;                           // every rank has fed its whole buffer
    if(growing && v[3]) {
      rc = comm_insert_prev(c, R); if(rc) return rc;
      rc = measure_occupancy(t); if(rc) return rc;
      uint64_t full = t->occ_known > (1ull << t->g.lsize_l) / 2 ? 1 : 0;
      rc = jfgpu_comm_allreduce_u64(c, &full, 1, 1); if(rc) return rc;
      IPC_TRACE(c, "step: shard holds %llu of %llu slots%s", (unsigned long long)t->occ_known, (unsigned long long)(1ull << t->g.lsize_l), full ? ": the shards double" : "");
      if(full) { rc = comm_grow(c); if(rc) return rc; }
      first = true;                                          // (the agreement is taken again: new head-room, maybe a new geometry)
      continue;
    }
    const uint64_t agreed = ~0ull - v[4];                    // the smallest head-room
    size_t piece = (size_t)std::min<uint64_t>(left, agreed);
    if(piece < left && piece < 2 * k) piece = (size_t)std::min<uint64_t>(left, 2 * k);      // (a piece holds a window and moves forward)
    const uint32_t cap = v[0] ? 0u : (uint32_t)v[1];
    IPC_TRACE(c, "step: piece of %zu bytes, agreed cap %u (0: keys)", piece, cap);
    rc = comm_piece_rccl(c, R, d_bases + off, piece, cap); if(rc) return rc;
    if(growing) t->fed_since += comm_charge(std::min<uint64_t>(v[2], std::max<uint64_t>(agreed, 2 * k)));
    if(off + piece >= n) off = n; else off += piece - (size_t)(k - 1);      // the next piece re-reads the last k-1 characters: every window exactly once
    if(!growing) break;                                      // (no pieces without growth: the step is one exchange, as the caller counts them)
  }
  IPC_TRACE(c, "step: done");
  return JFGPU_OK;
}

// The same for every rank of a local communicator at once: tables[r], d_bases[r], n[r] are rank r's.
int jfgpu_comm_local_step(jfgpu_comm* c, jfgpu_table** tables, const char* const* d_bases, const size_t* n) {
  if(!c || !c->local) return fail(JFGPU_E_INVALID, "not a local communicator");
  if(!tables || !d_bases || !n) return fail(JFGPU_E_INVALID, "null argument");
  const int W = c->world;
  bool growing = true;
  for(int r = 0; r < W; ++r) {
    int rc = use(tables[r]); if(rc) return rc;
    if((int)tables[r]->g.shard_id != r) return fail(JFGPU_E_INVALID, "tables must be given in shard order");
    c->ranks[r].t = tables[r];
    growing = growing && comm_growing(tables[r]);
  }
  const uint64_t k = tables[0]->g.k;
  std::vector<size_t> off(W, 0), piece(W, 0);
  std::vector<const char*> at(W);
  for(bool first = true;; first = false) {
    // the agreement of jfgpu_comm_count_ascii_dev, with every rank in this process
    uint64_t max_left = 0, head = ~0ull >> 1; bool bound_out = false;
    for(int r = 0; r < W; ++r) {
      const uint64_t left = n[r] - off[r];
      max_left = std::max(max_left, left);
      if(growing) { head = std::min(head, comm_headroom(tables[r])); bound_out = bound_out || comm_bound_out(tables[r], left); }
    }
    if(!first && max_left == 0) break;
    if(growing && bound_out) {
      bool full = false;
      for(int r = 0; r < W; ++r) {
        int rc = comm_insert_prev(c, c->ranks[r]); if(rc) return rc;
        rc = measure_occupancy(tables[r]); if(rc) return rc;
        full = full || tables[r]->occ_known > (1ull << tables[r]->g.lsize_l) / 2;
      }
      if(full) {
        int rc = comm_grow(c); if(rc) return rc;
        growing = true;                                        // (a grow that had to be abandoned turns growth off on every shard)
        for(int r = 0; r < W; ++r) growing = growing && comm_growing(tables[r]);
      }
      first = true;
      continue;
    }
    uint32_t cap = 0xFFFFFFFFu, want_max = 0;
    for(int r = 0; r < W; ++r) {
      const uint64_t left = n[r] - off[r];
      piece[r] = (size_t)std::min<uint64_t>(left, head);
      if(piece[r] < left && piece[r] < 2 * k) piece[r] = (size_t)std::min<uint64_t>(left, 2 * k);
      at[r] = d_bases[r] + off[r];
      const uint32_t w = items_cap_wanted(c, c->ranks[r], piece[r]);
      if(!items_geometry_ok(c, tables[r]) || (!w && piece[r] >= k)) cap = 0;      // (a rank without input has no preference)
      want_max = std::max(want_max, w);
    }
    if(cap) cap = want_max;                                // (0 when nobody has input: the key path with nothing to send)
    for(int r = 0; r < W; ++r) { uint64_t all = 0; for(int q = 0; q < W; ++q) all += piece[q]; int rc = ensure_ovf(tables[r], all, 0); if(rc) return rc; }
    // (same order as the RCCL piece: routing enqueued, the previous piece's insert enqueued, then the host's look at the routing)
    if(cap) for(int r = 0; r < W; ++r) { int rc = comm_route_items_enqueue(c, c->ranks[r], at[r], piece[r], cap); if(rc) return rc; }
    for(int r = 0; r < W; ++r) { int rc = comm_insert_prev(c, c->ranks[r]); if(rc) return rc; }
    if(cap) {
      bool any_overflow = false;
      std::vector<uint64_t> routed(W, 0);
      for(int r = 0; r < W; ++r) {
        bool overflow = false;
        int rc = comm_route_items_complete(c, c->ranks[r], cap, &overflow, &routed[r]); if(rc) return rc;
        any_overflow = any_overflow || overflow;
      }
      if(any_overflow) cap = 0;
      else for(int r = 0; r < W; ++r) c->ranks[r].sent += routed[r];
    }
    for(int r = 0; r < W; ++r) c->ranks[r].icap[c->ranks[r].turn] = cap;
    int rc = JFGPU_OK;
    if(cap) rc = comm_exchange_items_local(c);
    else {
      for(int r = 0; r < W; ++r) { rc = comm_route(c, c->ranks[r], at[r], piece[r]); if(rc) return rc; }
      rc = comm_exchange_local(c);
    }
    if(rc) return rc;
    uint64_t max_piece = 0;
    for(int r = 0; r < W; ++r) {
      c->ranks[r].inflight = true; c->ranks[r].turn ^= 1;
      max_piece = std::max<uint64_t>(max_piece, piece[r]);
      if(off[r] + piece[r] >= n[r]) off[r] = n[r]; else off[r] += piece[r] - (size_t)(k - 1);
    }
    if(growing) for(int r = 0; r < W; ++r) tables[r]->fed_since += comm_charge(max_piece);
    if(!growing) break;
  }
  return JFGPU_OK;
}

// Small host-level collectives of the RCCL transport, for what a launcher has to agree on around the steps: whether any
// rank still has input (so that every rank makes the same number of steps), and how many records every shard will write
// (a rank's offset in the common output file).  values: n words, replaced by the sum (op 0) or the maximum (op 1) over
// the ranks.  all: [world] words, all[r] = rank r's `mine`.  Synchronous; collective.
int jfgpu_comm_allreduce_u64(jfgpu_comm* c, uint64_t* values, int n, int op) {
  if(!c || c->local) return fail(JFGPU_E_INVALID, "not an RCCL communicator");
  if(!values || n < 1 || n > 64 || (op != 0 && op != 1)) return fail(JFGPU_E_INVALID, "allreduce: 1..64 words, op 0 (sum) or 1 (max)");
#if defined(JFGPU_EMU)
  return fail(JFGPU_E_UNSUPPORTED, "no RCCL in the emulated build");
#else
  HIP_TRY(hipSetDevice(c->device));
  if(c->ipc) return ipc_collective(c, values, n, op);
  if(c->world == 1 && !c->self_rccl) return JFGPU_OK;
  if(!c->d_coll) HIP_TRY(hipMalloc((void**)&c->d_coll, sizeof(uint64_t) * 512));
  HIP_TRY(hipMemcpyAsync(c->d_coll, values, sizeof(uint64_t) * n, hipMemcpyHostToDevice, c->xstream));
  NCCL_TRY(ncclAllReduce(c->d_coll, c->d_coll, n, ncclUint64, op == 0 ? ncclSum : ncclMax, c->nccl, c->xstream));
  HIP_TRY(hipMemcpyAsync(values, c->d_coll, sizeof(uint64_t) * n, hipMemcpyDeviceToHost, c->xstream));
  HIP_TRY(hipStreamSynchronize(c->xstream));
  return JFGPU_OK;
#endif
}

int jfgpu_comm_allgather_u64(jfgpu_comm* c, uint64_t mine, uint64_t* all) {
  if(!c || c->local) return fail(JFGPU_E_INVALID, "not an RCCL communicator");
  if(!all) return fail(JFGPU_E_INVALID, "null argument");
#if defined(JFGPU_EMU)
  (void)mine;
  return fail(JFGPU_E_UNSUPPORTED, "no RCCL in the emulated build");
#else
  HIP_TRY(hipSetDevice(c->device));
  if(c->ipc) { all[0] = mine; return ipc_collective(c, all, 1, 2); }
  if(c->world == 1 && !c->self_rccl) { all[0] = mine; return JFGPU_OK; }
  if(c->world > 256) return fail(JFGPU_E_INVALID, "allgather: world too large");
  if(!c->d_coll) HIP_TRY(hipMalloc((void**)&c->d_coll, sizeof(uint64_t) * 512));
  HIP_TRY(hipMemcpyAsync(c->d_coll + 256, &mine, sizeof(uint64_t), hipMemcpyHostToDevice, c->xstream));
  NCCL_TRY(ncclAllGather(c->d_coll + 256, c->d_coll, 1, ncclUint64, c->nccl, c->xstream));
  HIP_TRY(hipMemcpyAsync(all, c->d_coll, sizeof(uint64_t) * c->world, hipMemcpyDeviceToHost, c->xstream));
  HIP_TRY(hipStreamSynchronize(c->xstream));
  return JFGPU_OK;
#endif
}

// `jellyfish bc` over the ranks of a communicator (bc_main.cc:84-161 with the input split between the GPUs): collective,
// after every rank has inserted its part; on return every rank's counter holds the counter of the whole input (comm_bc_merge).
int jfgpu_comm_bc_merge(jfgpu_comm* c, jfgpu_bloom* b) {
  if(!c || c->local) return fail(JFGPU_E_INVALID, "not an RCCL communicator");
  if(!b) return fail(JFGPU_E_INVALID, "null bloom counter");
  HIP_TRY(hipSetDevice(c->device));
  if(b->device != c->device) return fail(JFGPU_E_INVALID, "Bloom counter lives on another device");
  return comm_bc_merge(c, &b);
}
int jfgpu_comm_bc_merge_local(jfgpu_comm* c, jfgpu_bloom** blooms) {
  if(!c || !c->local) return fail(JFGPU_E_INVALID, "not a local communicator");
  if(!blooms) return fail(JFGPU_E_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  for(int r = 0; r < c->world; ++r) if(!blooms[r]) return fail(JFGPU_E_INVALID, "null bloom counter");
  return comm_bc_merge(c, blooms);
}

int jfgpu_comm_world(const jfgpu_comm* c, int* world, int* rank) {
  if(!c) return fail(JFGPU_E_INVALID, "null communicator");
  if(world) *world = c->world;
  if(rank) *rank = c->rank;
  return JFGPU_OK;
}

// Complete the last step's exchange and insert (call before jfgpu_sync / reading the tables).  sent / received: this
// rank's (local: all ranks') totals since creation, for the conservation check sum(sent) == sum(received).  Item-path
// steps: `received` counts what the receive side found in the regions and straggler lists that arrived (not what the
// senders announced; the two are compared here, and a difference is an error).
int jfgpu_comm_finish(jfgpu_comm* c, uint64_t* sent, uint64_t* received) {
  if(!c) return fail(JFGPU_E_INVALID, "null communicator");
  HIP_TRY(hipSetDevice(c->device));
  uint64_t s = 0, r = 0;
  for(auto& R : c->ranks) {
    if(R.t) { int rc = comm_insert_prev(c, R); if(rc) return rc; HIP_TRY(hipStreamSynchronize(R.t->stream)); }
    unsigned long long claimed = 0, arrived = 0;             // item-path steps: what the senders said they sent here / what this rank found
    HIP_TRY(hipMemcpy(&claimed, R.d_claimed, 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&arrived, R.d_arrived, 8, hipMemcpyDeviceToHost));
    if(arrived != claimed)
      return fail(JFGPU_E_HIP, "exchange: the senders announced " + std::to_string(claimed) + " k-mers for rank " + std::to_string(R.t ? (int)R.t->g.shard_id : c->rank) +
                               ", " + std::to_string(arrived) + " were found in what arrived");
    s += R.sent; r += R.received + arrived;
  }
  HIP_TRY(hipStreamSynchronize(c->xstream));
  if(sent) *sent = s;
  if(received) *received = r;
  return JFGPU_OK;
}

// hash_counter::add(key, val) for a batch with one value per key (large_hash_array.hpp:741-752 add_val with an arbitrary
// increment): what loads a binary/sorted file back into a table -- `query -s` answers from the device that way
// (sub_commands/query_main.cc:44-51).  Host arrays, staged; the growth / spill rules of jfgpu_add_keys.
int jfgpu_add_key_vals(jfgpu_table* t, const uint64_t* keys, const uint64_t* vals, size_t n) {
  int rc = use(t); if(rc) return rc;
  if(!n) return JFGPU_OK;
  if(!keys || !vals) return fail(JFGPU_E_INVALID, "null argument");
  if(t->nword) return fail(JFGPU_E_UNSUPPORTED, "add_key_vals: keys longer than two words are not built");
  if(t->g.shard_bits) return fail(JFGPU_E_UNSUPPORTED, "add_key_vals: not for a shard (route the keys first)");
  const size_t kw = t->key_words;
  uint64_t *d_k = nullptr, *d_v = nullptr;
  HIP_TRY(hipMalloc((void**)&d_k, n * kw * sizeof(uint64_t)));
  if(hipMalloc((void**)&d_v, n * sizeof(uint64_t)) != hipSuccess) { hipFree(d_k); return fail(JFGPU_E_ALLOC, "hipMalloc values"); }
  auto done = [&](int r) { hipStreamSynchronize(t->stream); hipFree(d_k); hipFree(d_v); return r; };
  if(hipMemcpyAsync(d_k, keys, n * kw * sizeof(uint64_t), hipMemcpyHostToDevice, t->stream) != hipSuccess ||
     hipMemcpyAsync(d_v, vals, n * sizeof(uint64_t), hipMemcpyHostToDevice, t->stream) != hipSuccess) return done(fail(JFGPU_E_HIP, "add_key_vals: copy"));
  size_t off = 0;
  while(off < n) {
    uint64_t take = n - off;
    if(capacity_managed(t)) { rc = ensure_capacity(t, n - off, &take); if(rc) return done(rc); }
    uint64_t small = 0, big = 0;                              // what the overflow side table has to be ready for (ensure_ovf's two bounds)
    for(size_t i = off; i < off + take; ++i) { if(t->g.cnt_bits < 64 && (vals[i] >> t->g.cnt_bits)) ++big; else small += vals[i]; }
    rc = ensure_ovf(t, small, big); if(rc) return done(rc);
    rc = part_flush(t); if(rc) return done(rc);
    t->pristine = false;
    const dim3 grid(grid_for(t, take / kBlock + 1)), block(kBlock);
    if(t->wide) hipLaunchKernelGGL(add_pairs_wide_kernel, grid, block, 0, t->stream, t->wt, (const uint64_t*)(d_k + off * kw), (const uint64_t*)(d_v + off), (uint64_t)take);
    else hipLaunchKernelGGL(add_pairs_kernel, grid, block, 0, t->stream, t->dt, (const uint64_t*)(d_k + off), (const uint64_t*)(d_v + off), (uint64_t)take);
    if(hipGetLastError() != hipSuccess) return done(fail(JFGPU_E_HIP, "add_key_vals: launch"));
    off += (size_t)take;
  }
  rc = done(JFGPU_OK);
  return rc ? rc : check_deferred(t);
}

int jfgpu_comm_exchange_times(jfgpu_comm* c, double* ms, uint64_t* wire_bytes, size_t cap, size_t* n) {
  if(!c || !n) return fail(JFGPU_E_INVALID, "null argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->xstream));
  for(auto& R : c->ranks) { comm_x_harvest(c, R, 0); comm_x_harvest(c, R, 1); }
  *n = c->x_log.size();
  for(size_t i = 0; i < c->x_log.size() && i < cap; ++i) { if(ms) ms[i] = c->x_log[i].first; if(wire_bytes) wire_bytes[i] = c->x_log[i].second; }
  c->x_log.clear(); c->x_next = 0;
  return JFGPU_OK;
}

}  // extern "C"
