// jellyfish_amd/csrc/abi_bloom.inl -- C ABI of the Bloom counter (jfgpu_bc_*), included by jfgpu.hip.
// ---- Bloom counter (jellyfish bc / count --bc, BASELINE config 3) ---------------------------------
struct jfgpu_bloom {
  int device = 0, n_cu = 256;
  Tuning tun;                    // the JFGPU_* switches as they were at creation (tuning.hpp)
  hipStream_t stream = nullptr;
  TableGeom g{};                 // only k / key_mask / canonical / nbytes are used (encode side)
  bool wide = false; WideGeom wg{};   // 33 <= k <= 64: two-word keys
  uint64_t m = 0; uint32_t nh = 0;
  Gf2Matrix m1, m2;
  uint64_t *d_t1 = nullptr, *d_t2 = nullptr;
  uint32_t* d_data = nullptr; size_t data_bytes = 0, alloc_bytes = 0;
  unsigned long long* d_mers = nullptr;
  uint8_t* d_stage = nullptr;
  // partitioned insert (kernels_bloom_part.hip.hpp, bloom_partition.inl)
  BloomPart bp{}; bool part_ok = false;
  uint32_t kind = 0;               // 0: Bloom counter (bc / count --bc); 1: one-pass Bloom filter of bits (count --bf-size)
  int mode = 0;                    // 0 auto, 1 direct (global CAS), 2 partitioned (JFGPU_BLOOM_MODE / jfgpu_bc_set_mode)
  double slack = 0.03;             // head-room of a bucket region over the mean
  int g1 = 0;
  uint8_t* ws = nullptr; size_t ws_cap = 0, ws_used = 0;
  struct Pending { uint32_t* items; uint64_t* off; unsigned long long* tot; uint32_t cap; };
  std::vector<Pending> pending;
  uint32_t* d_M2 = nullptr;
  uint64_t* d_strag2 = nullptr; uint32_t* d_strag2_n = nullptr; uint32_t strag2_lists = 0;     // p2_ring_kernel's straggler lists
  uint64_t* d_strag1 = nullptr; uint32_t* d_strag1_n = nullptr;                                // p1_bloom_ring_kernel's (one list per workgroup)
  int p1_ring_per = 0;                                                                         // cells per lane and round of the ring P1b (0: the sort-based kernels)
  bool prof_on = false;
  std::vector<ProfSpan> spans;
  double prof_ms[5] = {}; uint64_t prof_launches[5] = {}, prof_units[5] = {};      // [BS_COUNT] (bloom_partition.inl)
  DevBloom view() const { DevBloom b; b.data = d_data; b.m = m; b.recip = bloom_recip(m); b.nh = nh; b.kind = kind; b.pad_ = 0; b.nbytes = g.nbytes; b.tbl1 = d_t1; b.tbl2 = d_t2; b.cache = nullptr; b.cache_mask = 0; return b; }
};

namespace {
int use_b(const jfgpu_bloom* b) {
  if(!b) return fail(JFGPU_E_INVALID, "null bloom counter");
  HIP_TRY(hipSetDevice(b->device));
  return JFGPU_OK;
}
Gf2Matrix random_full_matrix(uint32_t c, uint64_t& seed_state) {   // hash_pair<mer_dna>: 64 x 2k, any matrix
  Gf2Matrix m; m.r = 64; m.c = c; m.columns.assign(c, 0);
  for(uint32_t i = 0; i < c; ++i) m.columns[i] = splitmix64(seed_state);
  return m;
}
void plain_tables(const Gf2Matrix& m, std::vector<uint64_t>& tbl) {
  std::vector<uint64_t> img(m.c);
  for(uint32_t j = 0; j < m.c; ++j) img[j] = m.col_for_bit(j);
  gf2_byte_tables(img, (m.c + 7) / 8, tbl);
}
typedef jfgpu_bloom::Pending BloomPending;
#include "bloom_partition.inl"
}  // namespace

extern "C" {

uint64_t jfgpu_bc_opt_m(double fp, uint64_t n) { return n * (uint64_t)lrint(-log(fp) / 0.4804530139182014); }   // bloom_common.hpp:61-63
uint32_t jfgpu_bc_opt_k(double fp) { return (uint32_t)lrint(-log(fp) / 0.6931471805599453); }                     // :64-66

static int bloom_create(const jfgpu_bloom_params* p, jfgpu_bloom** out, uint32_t kind);
int jfgpu_bc_create(const jfgpu_bloom_params* p, jfgpu_bloom** out) { return bloom_create(p, out, 0); }
int jfgpu_bf_create(const jfgpu_bloom_params* p, jfgpu_bloom** out) { return bloom_create(p, out, 1); }

static int bloom_create(const jfgpu_bloom_params* p, jfgpu_bloom** out, uint32_t kind) {
  if(!p || !out) return fail(JFGPU_E_INVALID, "null argument");
  *out = nullptr;
  if(p->k < 1) return fail(JFGPU_E_INVALID, "mer length must be >= 1");
  if(p->k > 64) return fail(JFGPU_E_UNSUPPORTED, "mer length > 64 (more than two key words) is not built yet");
  if(p->m < 1 || p->nb_hashes < 1 || p->nb_hashes > 64) return fail(JFGPU_E_INVALID, "bad Bloom counter size / number of hashes");
  int ndev = 0;
  if(hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(JFGPU_E_NO_DEVICE, "no HIP device: the engine has no CPU fallback");
  int dev = p->device;
  if(dev < 0) HIP_TRY(hipGetDevice(&dev));
  if(dev >= ndev) return fail(JFGPU_E_NO_DEVICE, "device ordinal out of range");
  HIP_TRY(hipSetDevice(dev));
  std::unique_ptr<jfgpu_bloom> b(new jfgpu_bloom);
  b->device = dev; b->m = p->m; b->nh = p->nb_hashes; b->kind = kind;
  if(p->k > 32) {
    b->wide = true;
    if(!wide_geom_init(b->wg, p->k, std::max<uint32_t>(kMaxTileBits, wide_min_lsize(p->k)), p->canonical ? 1 : 0)) return fail(JFGPU_E_INVALID, "bad mer length");
    b->g = b->wg.g;
  } else {
    const uint32_t ls = std::max<uint32_t>(std::min<uint32_t>(2 * p->k, 13), geom_min_lsize(p->k, 0));
    if(!geom_init(b->g, p->k, ls, 0, 0, p->canonical ? 1 : 0)) return fail(JFGPU_E_INVALID, "bad mer length");
  }
  uint64_t st = p->seed ? p->seed : 0x626C6F6F6D636E74ull;
  if(p->matrix1 && p->matrix2) {
    b->m1.r = b->m2.r = 64; b->m1.c = b->m2.c = 2 * p->k;
    b->m1.columns.assign(p->matrix1, p->matrix1 + 2 * p->k);
    b->m2.columns.assign(p->matrix2, p->matrix2 + 2 * p->k);
  } else if(p->seed) {
    b->m1 = random_full_matrix(2 * p->k, st);
    b->m2 = random_full_matrix(2 * p->k, st);
  } else {
    // the reference's default pair (mer_dna_bloom_counter.hpp:21-26): m1 then m2, 64 x 2k, straight from random_bits()
    GlibcRandom rng;
    for(Gf2Matrix* m : {&b->m1, &b->m2}) {
      m->r = 64; m->c = 2 * p->k; m->columns.assign(2 * p->k, 0);
      for(uint32_t i = 0; i < 2 * p->k; ++i) m->columns[i] = rng.bits64();
    }
  }
  std::vector<uint64_t> t1, t2;
  plain_tables(b->m1, t1); plain_tables(b->m2, t2);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  b->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  b->data_bytes = kind == 1 ? b->m / 8 + (b->m % 8 != 0)         // bloom_filter.hpp:25-27
                            : b->m / 5 + (b->m % 5 != 0);        // bloom_counter2.hpp:40-42
  b->alloc_bytes = (b->data_bytes + 3) / 4 * 4 + 4;
  if(kind == 0) bloom_part_init(b.get());
  if(b->part_ok) b->alloc_bytes = ((size_t)b->bp.n_seg << kBloomSegBits) + 4;      // whole segments are loaded and stored
  b->tun = Tuning::from_env();
  if(b->tun.bloom_mode == 1) b->mode = 1;
  else if(b->tun.bloom_mode == 2 && b->part_ok) b->mode = 2;
  {
    const int pl = (int)((size_t)kBloomChunk * 6 + (size_t)2 * 8 * 2048);
    HIP_TRY(hipFuncSetAttribute((const void*)p1_bloom_granule_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, pl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_bloom_granule_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, pl));
    HIP_TRY(hipFuncSetAttribute((const void*)p1_bloom_granule_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, pl));
    {
      const int pl2 = (int)((size_t)kPBlock * 5 * 6 + (size_t)8 * 512);
      HIP_TRY(hipFuncSetAttribute((const void*)p1_bloom_granule2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, pl2));
      HIP_TRY(hipFuncSetAttribute((const void*)p1_bloom_granule2_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, pl2));
      HIP_TRY(hipFuncSetAttribute((const void*)p1_bloom_granule2_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, pl2));
    }
    {
      const int plr = (int)((size_t)512 * kBloomRingBytes + kBloomRingBytes + (size_t)8 * 512);
      HIP_TRY(hipFuncSetAttribute((const void*)p1_bloom_ring_kernel<0, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, plr));
      HIP_TRY(hipFuncSetAttribute((const void*)p1_bloom_ring_kernel<6, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, plr));
      HIP_TRY(hipFuncSetAttribute((const void*)p1_bloom_ring_kernel<8, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, plr));
      HIP_TRY(hipFuncSetAttribute((const void*)p1_bloom_ring_kernel<0, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, plr));
      HIP_TRY(hipFuncSetAttribute((const void*)p1_bloom_ring_kernel<8, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, plr));
    }
    HIP_TRY(hipFuncSetAttribute((const void*)bloom_segment_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 1 << kBloomSegBits));
    HIP_TRY(hipFuncSetAttribute((const void*)p2_scatter_sorted_kernel<uint32_t, kP2PairPer>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * kP2PairPer * 4));
    HIP_TRY(hipFuncSetAttribute((const void*)p2_granule_kernel<uint32_t, BloomDirect, kP2PairPer>, hipFuncAttributeMaxDynamicSharedMemorySize, kPBlock * kP2PairPer * 4));
    HIP_TRY(hipFuncSetAttribute((const void*)p2_ring_kernel<BloomRingDirect>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kGranMaxB * 128 + 128)));
    HIP_TRY(hipFuncSetAttribute((const void*)p2_ring_roles_kernel<uint32_t, 2, BloomRingDirect, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kGranMaxB * 128 + 128)));
  }
  HIP_TRY(hipMalloc((void**)&b->d_data, b->alloc_bytes));
  HIP_TRY(hipMemsetAsync(b->d_data, 0, b->alloc_bytes, b->stream));
  HIP_TRY(hipMalloc((void**)&b->d_t1, t1.size() * 8)); HIP_TRY(hipMalloc((void**)&b->d_t2, t2.size() * 8));
  HIP_TRY(hipMemcpy(b->d_t1, t1.data(), t1.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b->d_t2, t2.data(), t2.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void**)&b->d_mers, sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(b->d_mers, 0, sizeof(unsigned long long), b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  *out = b.release();
  return JFGPU_OK;
}

void jfgpu_bc_destroy(jfgpu_bloom* b) {
  if(!b) return;
  hipSetDevice(b->device);
  if(b->stream) hipStreamSynchronize(b->stream);
  bloom_prof_collect(b);
  hipFree(b->d_data); hipFree(b->d_t1); hipFree(b->d_t2); hipFree(b->d_mers);
  if(b->d_stage) hipFree(b->d_stage);
  if(b->ws) hipFree(b->ws);
  if(b->d_M2) hipFree(b->d_M2);
  if(b->d_strag2) { hipFree(b->d_strag2); hipFree(b->d_strag2_n); }
  if(b->d_strag1) { hipFree(b->d_strag1); hipFree(b->d_strag1_n); }
  if(b->stream) hipStreamDestroy(b->stream);
  delete b;
}

int jfgpu_bc_insert_ascii_dev(jfgpu_bloom* b, const char* d_bases, size_t n) {
  int rc = use_b(b); if(rc) return rc;
  if(b->kind != 0) return fail(JFGPU_E_INVALID, "a one-pass Bloom filter is fed by the count it is attached to");
  if(n < b->g.k) return JFGPU_OK;
  if(!d_bases) return fail(JFGPU_E_INVALID, "null buffer");
  const uint8_t* base; int64_t lo, hi;
  align_buffer(d_bases, n, base, lo, hi);
  // the increments saturate and commute, so batches may be applied in any order: large ones are buffered as routed cell
  // updates and applied segment by segment at the next flush (sync / read / attach), small ones go straight to the array
  if(bloom_use_partitioned(b, n)) return bloom_ingest(b, base, lo, hi);
  return bloom_launch_direct(b, base, lo, hi);
}

int jfgpu_bc_insert_ascii(jfgpu_bloom* b, const char* bases, size_t n) {
  int rc = use_b(b); if(rc) return rc;
  if(n < b->g.k) return JFGPU_OK;
  if(!bases) return fail(JFGPU_E_INVALID, "null buffer");
  if(!b->d_stage) HIP_TRY(hipMalloc((void**)&b->d_stage, kStageBytes));
  const size_t step = kStageBytes - (b->g.k - 1);
  for(size_t o = 0; o < n; o += step) {
    const size_t len = std::min(kStageBytes, n - o);
    HIP_TRY(hipStreamSynchronize(b->stream));
    HIP_TRY(hipMemcpyAsync(b->d_stage, bases + o, len, hipMemcpyHostToDevice, b->stream));
    rc = jfgpu_bc_insert_ascii_dev(b, (const char*)b->d_stage, len); if(rc) return rc;
    if(o + len >= n) break;
  }
  return JFGPU_OK;
}

int jfgpu_bc_clear(jfgpu_bloom* b) {
  int rc = use_b(b); if(rc) return rc;
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->pending.clear(); b->ws_used = 0;                       // pending cell updates are dropped with the rest
  HIP_TRY(hipMemsetAsync(b->d_data, 0, b->alloc_bytes, b->stream));
  HIP_TRY(hipMemsetAsync(b->d_mers, 0, sizeof(unsigned long long), b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return JFGPU_OK;
}

int jfgpu_bc_sync(jfgpu_bloom* b, uint64_t* mers_fed) {
  int rc = use_b(b); if(rc) return rc;
  rc = bloom_flush(b); if(rc) return rc;
  unsigned long long m = 0;
  HIP_TRY(hipMemcpyAsync(&m, b->d_mers, sizeof m, hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  if(mers_fed) *mers_fed = m;
  return JFGPU_OK;
}

int jfgpu_bc_get_info(const jfgpu_bloom* b, uint64_t* m, uint32_t* nb_hashes, uint64_t* nb_bytes, uint64_t* matrix1, uint64_t* matrix2) {
  if(!b) return fail(JFGPU_E_INVALID, "null bloom counter");
  if(m) *m = b->m;
  if(nb_hashes) *nb_hashes = b->nh;
  if(nb_bytes) *nb_bytes = b->data_bytes;
  if(matrix1) memcpy(matrix1, b->m1.columns.data(), 8 * b->m1.c);
  if(matrix2) memcpy(matrix2, b->m2.columns.data(), 8 * b->m2.c);
  return JFGPU_OK;
}

int jfgpu_bc_read(jfgpu_bloom* b, uint8_t* out) {          // bloom_base::write_bits (bloom_common.hpp)
  int rc = use_b(b); if(rc) return rc;
  rc = bloom_flush(b); if(rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, b->d_data, b->data_bytes, hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return JFGPU_OK;
}

int jfgpu_bc_load(jfgpu_bloom* b, const uint8_t* data) {   // bloom_counter2(m, k, istream&, fns)
  int rc = use_b(b); if(rc) return rc;
  rc = bloom_flush(b); if(rc) return rc;
  HIP_TRY(hipMemcpyAsync(b->d_data, data, b->data_bytes, hipMemcpyHostToDevice, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return JFGPU_OK;
}

int jfgpu_bc_keys(jfgpu_bloom* b, const uint64_t* keys, size_t n, uint8_t* out, int do_insert) {
  int rc = use_b(b); if(rc) return rc;
  if(b->kind != 0) return fail(JFGPU_E_UNSUPPORTED, "check / insert on encoded k-mers is built for Bloom counters only");
  rc = bloom_flush(b); if(rc) return rc;
  if(!n) return JFGPU_OK;
  uint64_t* d_k = nullptr; uint8_t* d_o = nullptr;
  const size_t kw = b->wide ? 2 : 1;                       // words per key (little-endian words, like jfgpu_add_keys)
  HIP_TRY(hipMalloc((void**)&d_k, n * 8 * kw));
  if(hipMalloc((void**)&d_o, n) != hipSuccess) { hipFree(d_k); return fail(JFGPU_E_ALLOC, "hipMalloc"); }
  hipError_t e = hipMemcpyAsync(d_k, keys, n * 8 * kw, hipMemcpyHostToDevice, b->stream);
  if(e == hipSuccess) {
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((n + kBlock - 1) / kBlock, (size_t)b->n_cu * 8));
    if(b->wide) hipLaunchKernelGGL(bloom_keys_wide_kernel, dim3(grid), dim3(kBlock), 0, b->stream, b->view(), b->wg.key_mask, (const uint64_t*)d_k, (uint64_t)n, d_o, do_insert);
    else
    hipLaunchKernelGGL(bloom_keys_kernel, dim3(grid), dim3(kBlock), 0, b->stream, b->view(), (const uint64_t*)d_k, (uint64_t)n, d_o, do_insert);
    e = hipGetLastError();
  }
  if(e == hipSuccess && out) e = hipMemcpyAsync(out, d_o, n, hipMemcpyDeviceToHost, b->stream);
  hipStreamSynchronize(b->stream);
  hipFree(d_k); hipFree(d_o);
  if(e != hipSuccess) return fail(JFGPU_E_HIP, hipGetErrorString(e));
  return JFGPU_OK;
}

int jfgpu_bc_set_mode(jfgpu_bloom* b, int mode) {
  int rc = use_b(b); if(rc) return rc;
  if(mode < 0 || mode > 2) return fail(JFGPU_E_INVALID, "mode must be 0 (auto), 1 (direct) or 2 (partitioned)");
  if(mode == 2 && !b->part_ok) return fail(JFGPU_E_UNSUPPORTED, "this Bloom counter has no partitioned insert path");
  rc = bloom_flush(b); if(rc) return rc;
  b->mode = mode;
  return JFGPU_OK;
}

int jfgpu_bc_reserve(jfgpu_bloom* b, uint64_t workspace_bytes) {
  int rc = use_b(b); if(rc) return rc;
  if(!b->part_ok || b->mode == 1) return JFGPU_OK;
  rc = bloom_flush(b); if(rc) return rc;
  rc = bloom_ws_ensure(b, std::max<uint64_t>(workspace_bytes, (uint64_t)64 << 20));
  if(rc < 0) return fail(JFGPU_E_ALLOC, "not enough device memory for the Bloom partition workspace");
  if(rc) return rc;
  HIP_TRY(hipStreamSynchronize(b->stream));
  return JFGPU_OK;
}

int jfgpu_bc_profile_enable(jfgpu_bloom* b, int on) {
  int rc = use_b(b); if(rc) return rc;
  b->prof_on = on != 0;
  return JFGPU_OK;
}

int jfgpu_bc_profile_get(jfgpu_bloom* b, int which, double* ms, uint64_t* launches, uint64_t* units) {
  int rc = use_b(b); if(rc) return rc;
  if(which < 0 || which >= BS_COUNT) return fail(JFGPU_E_INVALID, "bad profile slot");
  HIP_TRY(hipStreamSynchronize(b->stream));
  bloom_prof_collect(b);
  if(ms) *ms = b->prof_ms[which];
  if(launches) *launches = b->prof_launches[which];
  if(units) *units = b->prof_units[which];
  return JFGPU_OK;
}

int jfgpu_bc_profile_reset(jfgpu_bloom* b) {
  int rc = use_b(b); if(rc) return rc;
  HIP_TRY(hipStreamSynchronize(b->stream));
  bloom_prof_collect(b);
  for(int i = 0; i < BS_COUNT; ++i) { b->prof_ms[i] = 0; b->prof_launches[i] = 0; b->prof_units[i] = 0; }
  return JFGPU_OK;
}

int jfgpu_attach_bloom(jfgpu_table* t, jfgpu_bloom* b) {   // count --bc (count_main.cc:191-206,313-316)
  int rc = use(t); if(rc) return rc;
  rc = part_flush(t); if(rc) return rc;
  // (the new counter is looked at BEFORE anything of the old attachment is given up: a refused counter leaves the table as
  // it was -- round-5 advisor finding: the cache used to be freed first, and a refusal left the old view pointing at it)
  if(b) {
    if(b->device != t->device) return fail(JFGPU_E_INVALID, "Bloom counter lives on another device");
    if(b->g.k != t->g.k) return fail(JFGPU_E_INVALID, "Invalid mer length in bloom filter");
    // a shard: the counter is asked on the sending side of the exchange (abi_comm.inl: comm_filter_ok), never on arrival
    if(t->g.shard_bits && b->kind != 0) return fail(JFGPU_E_UNSUPPORTED, "sharded tables take a Bloom counter (count --bc), not a one-pass filter (--bf-size)");
    if(t->nword) return fail(JFGPU_E_UNSUPPORTED, "Bloom filters for mer length > 64 are not built");
  }
  // the cache of admitted k-mers belongs to one attachment: its answers are this counter's
  if(t->d_bcache) {
    HIP_TRY(hipStreamSynchronize(t->stream));
    t->dt.bloom.cache = nullptr; t->dt.bloom.cache_mask = 0;
    hipFree(t->d_bcache); t->d_bcache = nullptr;
  }
  t->bcache_state = 0;
  if(!b) { memset(&t->dt.bloom, 0, sizeof t->dt.bloom); memset(&t->wt.bloom, 0, sizeof t->wt.bloom); return JFGPU_OK; }
  rc = bloom_flush(b); if(rc) return rc;
  HIP_TRY(hipStreamSynchronize(b->stream));
  if(t->wide) t->wt.bloom = b->view(); else t->dt.bloom = b->view();
  { uint64_t c[CTR_COUNT]; rc = read_counters(t, c); if(rc) return rc; t->mers_seen = c[CTR_MERS]; }
  if(b->kind != 0 || t->wide || t->tun.bloom_cache == 0) t->bcache_state = -1;      // (a one-pass filter changes as it is asked: nothing to remember)
  else if(t->tun.bloom_cache == 1) { rc = bloom_cache_enable(t); if(rc) return rc; }
  return JFGPU_OK;
}

}  // extern "C"
