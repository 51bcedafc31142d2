// jellyfish_amd/csrc/abi_bloom.inl -- C ABI of the Bloom counter (jfgpu_bc_*), included by jfgpu.hip.
// ---- Bloom counter (jellyfish bc / count --bc, BASELINE config 3) ---------------------------------
struct jfgpu_bloom {
  int device = 0, n_cu = 256;
  hipStream_t stream = nullptr;
  TableGeom g{};                 // only k / key_mask / canonical / nbytes are used (encode side)
  bool wide = false; WideGeom wg{};   // 33 <= k <= 64: two-word keys
  uint64_t m = 0; uint32_t nh = 0;
  Gf2Matrix m1, m2;
  uint64_t *d_t1 = nullptr, *d_t2 = nullptr;
  uint32_t* d_data = nullptr; size_t data_bytes = 0, alloc_bytes = 0;
  unsigned long long* d_mers = nullptr;
  uint8_t* d_stage = nullptr;
  DevBloom view() const { DevBloom b; b.data = d_data; b.m = m; b.nh = nh; b.nbytes = g.nbytes; b.tbl1 = d_t1; b.tbl2 = d_t2; return b; }
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
}  // namespace

extern "C" {

uint64_t jfgpu_bc_opt_m(double fp, uint64_t n) { return n * (uint64_t)lrint(-log(fp) / 0.4804530139182014); }   // bloom_common.hpp:61-63
uint32_t jfgpu_bc_opt_k(double fp) { return (uint32_t)lrint(-log(fp) / 0.6931471805599453); }                     // :64-66

int jfgpu_bc_create(const jfgpu_bloom_params* p, jfgpu_bloom** out) {
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
  b->device = dev; b->m = p->m; b->nh = p->nb_hashes;
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
  b->data_bytes = b->m / 5 + (b->m % 5 != 0);                    // bloom_counter2.hpp:40-42
  b->alloc_bytes = (b->data_bytes + 3) / 4 * 4 + 4;
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
  hipFree(b->d_data); hipFree(b->d_t1); hipFree(b->d_t2); hipFree(b->d_mers);
  if(b->d_stage) hipFree(b->d_stage);
  if(b->stream) hipStreamDestroy(b->stream);
  delete b;
}

int jfgpu_bc_insert_ascii_dev(jfgpu_bloom* b, const char* d_bases, size_t n) {
  int rc = use_b(b); if(rc) return rc;
  if(n < b->g.k) return JFGPU_OK;
  if(!d_bases) return fail(JFGPU_E_INVALID, "null buffer");
  const uint8_t* base; int64_t lo, hi;
  align_buffer(d_bases, n, base, lo, hi);
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n_tiles, (int64_t)b->n_cu * 8));
  if(b->wide) hipLaunchKernelGGL(bloom_insert_ascii_wide_kernel, dim3(grid), dim3(kBlock), 0, b->stream, b->view(), b->wg, base, lo, hi, b->d_mers);
  else hipLaunchKernelGGL(bloom_insert_ascii_kernel, dim3(grid), dim3(kBlock), 0, b->stream, b->view(), b->g, base, lo, hi, b->d_mers);
  HIP_TRY(hipGetLastError());
  return JFGPU_OK;
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

int jfgpu_bc_sync(jfgpu_bloom* b, uint64_t* mers_fed) {
  int rc = use_b(b); if(rc) return rc;
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
  HIP_TRY(hipMemcpyAsync(out, b->d_data, b->data_bytes, hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return JFGPU_OK;
}

int jfgpu_bc_load(jfgpu_bloom* b, const uint8_t* data) {   // bloom_counter2(m, k, istream&, fns)
  int rc = use_b(b); if(rc) return rc;
  HIP_TRY(hipMemcpyAsync(b->d_data, data, b->data_bytes, hipMemcpyHostToDevice, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return JFGPU_OK;
}

int jfgpu_bc_keys(jfgpu_bloom* b, const uint64_t* keys, size_t n, uint8_t* out, int do_insert) {
  int rc = use_b(b); if(rc) return rc;
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

int jfgpu_attach_bloom(jfgpu_table* t, jfgpu_bloom* b) {   // count --bc (count_main.cc:191-206,313-316)
  int rc = use(t); if(rc) return rc;
  rc = part_flush(t); if(rc) return rc;
  if(!b) { memset(&t->dt.bloom, 0, sizeof t->dt.bloom); memset(&t->wt.bloom, 0, sizeof t->wt.bloom); return JFGPU_OK; }
  if(b->device != t->device) return fail(JFGPU_E_INVALID, "Bloom counter lives on another device");
  if(b->g.k != t->g.k) return fail(JFGPU_E_INVALID, "Invalid mer length in bloom filter");
  if(t->g.shard_bits) return fail(JFGPU_E_UNSUPPORTED, "count --bc on a sharded table is not built yet");
  HIP_TRY(hipStreamSynchronize(b->stream));
  if(t->wide) t->wt.bloom = b->view(); else t->dt.bloom = b->view();
  return JFGPU_OK;
}

}  // extern "C"
