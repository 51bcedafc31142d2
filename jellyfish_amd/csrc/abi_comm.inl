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
#if !defined(JFGPU_EMU)
#include <rccl/rccl.h>
#endif

struct jfgpu_comm {
  int world = 1, rank = 0, device = 0;
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
  };
  std::vector<Rank> ranks;                       // RCCL transport: one; local transport: `world`
};

namespace {

int comm_init_rank(jfgpu_comm* c, jfgpu_comm::Rank& R) {
  for(int i = 0; i < 2; ++i) {
    HIP_TRY(hipEventCreateWithFlags(&R.routed[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&R.exchanged[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&R.consumed[i], hipEventDisableTiming));
    R.scount[i].assign(c->world, 0); R.soff[i].assign(c->world + 1, 0); R.rcount[i].assign(c->world, 0); R.roff[i].assign(c->world + 1, 0);
  }
  HIP_TRY(hipMalloc((void**)&R.d_cnt, sizeof(unsigned long long) * c->world));
  HIP_TRY(hipMalloc((void**)&R.d_xc, sizeof(uint64_t) * 2 * c->world));
  return JFGPU_OK;
}

int comm_reserve(uint64_t*& buf, size_t& cap, size_t need, hipStream_t s1, hipStream_t s2) {
  if(need <= cap) return JFGPU_OK;
  HIP_TRY(hipStreamSynchronize(s1)); HIP_TRY(hipStreamSynchronize(s2));
  if(buf) hipFree(buf);
  buf = nullptr; cap = 0;
  const size_t want = need + need / 16 + 1024;
  HIP_TRY(hipMalloc((void**)&buf, want * sizeof(uint64_t)));
  cap = want;
  return JFGPU_OK;
}

// Route one contract buffer of rank R into send[cur], grouped by owner; fills scount / soff.  The host waits for the
// per-owner counts (one small copy) -- they place the groups and size the messages.
int comm_route(jfgpu_comm* c, jfgpu_comm::Rank& R, const char* d_bases, size_t n) {
  jfgpu_table* t = R.t;
  const int cur = R.turn, W = c->world;
  if((int)(1u << t->g.shard_bits) != W) return fail(JFGPU_E_INVALID, "table shard_bits does not match the communicator's world size");
  if(t->wide || t->nword) return fail(JFGPU_E_UNSUPPORTED, "sharded tables with mer length > 32 are not built yet");
  if(R.used[cur]) HIP_TRY(hipEventSynchronize(R.exchanged[cur]));       // send[cur] has left (step - 2)
  std::fill(R.scount[cur].begin(), R.scount[cur].end(), 0);
  std::fill(R.soff[cur].begin(), R.soff[cur].end(), 0);
  if(n < t->g.k) return JFGPU_OK;
  int rc = comm_reserve(R.send[cur], R.send_cap[cur], n, t->stream, c->xstream); if(rc) return rc;
  const uint8_t* base; int64_t lo, hi;
  align_buffer(d_bases, n, base, lo, hi);
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  const int grid = grid_for(t, (uint64_t)n_tiles);
  HIP_TRY(hipMemsetAsync(R.d_cnt, 0, sizeof(unsigned long long) * W, t->stream));
  {
    ProfScope ps(t, 2, n);
    hipLaunchKernelGGL(partition_count_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, base, lo, hi, R.d_cnt);
  }
  std::vector<unsigned long long> h(W);
  HIP_TRY(hipMemcpyAsync(h.data(), R.d_cnt, sizeof(unsigned long long) * W, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  uint64_t total = 0;
  for(int p = 0; p < W; ++p) { R.scount[cur][p] = h[p]; R.soff[cur][p] = total; total += h[p]; h[p] = R.soff[cur][p]; }
  R.soff[cur][W] = total;
  HIP_TRY(hipMemcpyAsync(R.d_cnt, h.data(), sizeof(unsigned long long) * W, hipMemcpyHostToDevice, t->stream));   // cursors = offsets
  {
    ProfScope ps(t, 2, 0);
    hipLaunchKernelGGL(partition_scatter_kernel, dim3(grid), dim3(kBlock), 0, t->stream, t->dt, base, lo, hi, R.d_cnt, R.send[cur]);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(t->stream));       // the host vector h is read by the copy above
  R.sent += total;
  return JFGPU_OK;
}

// What arrived for the previous step goes into the table (P1 from keys: pending batch applied at the next flush).
int comm_insert_prev(jfgpu_comm* c, jfgpu_comm::Rank& R) {
  if(!R.inflight) return JFGPU_OK;
  const int prev = R.turn ^ 1;
  jfgpu_table* t = R.t;
  HIP_TRY(hipStreamWaitEvent(t->stream, R.exchanged[prev], 0));
  const uint64_t n = R.roff[prev][c->world];
  int rc = JFGPU_OK;
  if(n) rc = add_keys_piece(t, R.recv[prev], (size_t)n, 1, nullptr);
  HIP_TRY(hipEventRecord(R.consumed[prev], t->stream));
  R.received += n;
  R.inflight = false;
  return rc;
}

#if !defined(JFGPU_EMU)
#define NCCL_TRY(expr) do { ncclResult_t r_ = (expr); if(r_ != ncclSuccess) return fail(JFGPU_E_HIP, std::string(#expr) + ": " + ncclGetErrorString(r_)); } while(0)
#endif

// The exchange of the current step for the RCCL transport: counts (host-visible), then the keys in rounds.
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
  if(R.used[cur]) HIP_TRY(hipEventSynchronize(R.consumed[cur]));          // recv[cur] was read by the insert of step - 2
  int rc = comm_reserve(R.recv[cur], R.recv_cap[cur], total, R.t->stream, c->xstream); if(rc) return rc;
  HIP_TRY(hipEventRecord(R.routed[cur], R.t->stream));
  HIP_TRY(hipStreamWaitEvent(c->xstream, R.routed[cur], 0));
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

void comm_free_rank(jfgpu_comm::Rank& R) {
  for(int i = 0; i < 2; ++i) {
    if(R.send[i]) hipFree(R.send[i]);
    if(R.recv[i]) hipFree(R.recv[i]);
    if(R.routed[i]) hipEventDestroy(R.routed[i]);
    if(R.exchanged[i]) hipEventDestroy(R.exchanged[i]);
    if(R.consumed[i]) hipEventDestroy(R.consumed[i]);
  }
  if(R.d_cnt) hipFree(R.d_cnt);
  if(R.d_xc) hipFree(R.d_xc);
}

}  // namespace

extern "C" {

int jfgpu_comm_unique_id(uint8_t* id128) {
  if(!id128) return fail(JFGPU_E_INVALID, "null id");
#if defined(JFGPU_EMU)
  memset(id128, 0, 128);
  return JFGPU_OK;
#else
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
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  NCCL_TRY(ncclCommInitRank(&c->nccl, world, id, rank));
  HIP_TRY(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
  c->ranks.resize(1);
  int rc = comm_init_rank(c.get(), c->ranks[0]); if(rc) return rc;
  if(const char* e = getenv("JFGPU_COMM_MAX_MSG")) c->max_msg_keys = std::max<uint64_t>(1, strtoull(e, 0, 10));
  if(const char* e = getenv("JFGPU_COMM_SELF_RCCL")) c->self_rccl = atoi(e) != 0;
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
  if(const char* e = getenv("JFGPU_COMM_MAX_MSG")) c->max_msg_keys = std::max<uint64_t>(1, strtoull(e, 0, 10));
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
#endif
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
  rc = comm_route(c, R, d_bases, n); if(rc) return rc;
  rc = comm_exchange_rccl(c); if(rc) return rc;
  rc = comm_insert_prev(c, R); if(rc) return rc;            // overlaps with the exchange just enqueued
  R.inflight = true; R.turn ^= 1;
  return JFGPU_OK;
}

// The same for every rank of a local communicator at once: tables[r], d_bases[r], n[r] are rank r's.
int jfgpu_comm_local_step(jfgpu_comm* c, jfgpu_table** tables, const char* const* d_bases, const size_t* n) {
  if(!c || !c->local) return fail(JFGPU_E_INVALID, "not a local communicator");
  if(!tables || !d_bases || !n) return fail(JFGPU_E_INVALID, "null argument");
  for(int r = 0; r < c->world; ++r) {
    int rc = use(tables[r]); if(rc) return rc;
    if((int)tables[r]->g.shard_id != r) return fail(JFGPU_E_INVALID, "tables must be given in shard order");
    c->ranks[r].t = tables[r];
    rc = comm_route(c, c->ranks[r], d_bases[r], n[r]); if(rc) return rc;
  }
  int rc = comm_exchange_local(c); if(rc) return rc;
  for(int r = 0; r < c->world; ++r) {
    rc = comm_insert_prev(c, c->ranks[r]); if(rc) return rc;
    c->ranks[r].inflight = true; c->ranks[r].turn ^= 1;
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

int jfgpu_comm_world(const jfgpu_comm* c, int* world, int* rank) {
  if(!c) return fail(JFGPU_E_INVALID, "null communicator");
  if(world) *world = c->world;
  if(rank) *rank = c->rank;
  return JFGPU_OK;
}

// Complete the last step's exchange and insert (call before jfgpu_sync / reading the tables).  sent / received: this
// rank's (local: all ranks') totals since creation, for the conservation check sum(sent) == sum(received).
int jfgpu_comm_finish(jfgpu_comm* c, uint64_t* sent, uint64_t* received) {
  if(!c) return fail(JFGPU_E_INVALID, "null communicator");
  HIP_TRY(hipSetDevice(c->device));
  uint64_t s = 0, r = 0;
  for(auto& R : c->ranks) {
    if(R.t) { int rc = comm_insert_prev(c, R); if(rc) return rc; HIP_TRY(hipStreamSynchronize(R.t->stream)); }
    s += R.sent; r += R.received;
  }
  HIP_TRY(hipStreamSynchronize(c->xstream));
  if(sent) *sent = s;
  if(received) *received = r;
  return JFGPU_OK;
}

}  // extern "C"
