// jellyfish_amd/csrc/host_partition.inl -- host orchestration of the partitioned insert path (kernels_part.hip.hpp):
// workspace arena, P1 ingestion of a batch, the flush (P2 + tile insert).  Included by jfgpu.hip inside its
// anonymous namespace.
// ---- partitioned insert path: host orchestration (kernels_part.hip.hpp) -------------------
constexpr uint64_t kPartMinBytes = 1u << 20;   // AUTO: smaller device batches take the direct kernel

void part_geom_init(jfgpu_table* t) {
  const uint32_t bits = t->g.lsize_l - t->g.tile_bits;   // tile-index bits to resolve
  t->part_ok = false; t->item32 = false; t->item128 = false;
  if(t->g.tile_bits < kMaxTileBits) return;               // tiny table: one partial tile
  uint32_t b1, b2;
  const uint32_t one_level = t->wide ? 10 : 11;           // two-word keys only have the single-pass P1 (<= 1024 buckets)
  if(bits <= one_level) { b1 = bits; b2 = 0; }
  else { b2 = std::min<uint32_t>(11, (bits + 1) / 2); b1 = bits - b2; }
  if(b1 > one_level) return;
  t->pg.b1 = b1; t->pg.b2 = b2;
  t->pg.rest_shift = t->g.lsize_l - b1;
  t->pg.item_bits = t->pg.rest_shift + t->g.rem_bits;
  if(t->wide) { t->item128 = true; t->part_ok = t->pg.item_bits <= 128; return; }
  if(t->pg.item_bits > 64) return;
  t->item32 = t->pg.item_bits <= 32;
  t->part_ok = true;
}

size_t item_size(const jfgpu_table* t) { return t->item128 ? 16 : t->item32 ? 4 : 8; }

// The table's descriptor in device memory, for kernels that leave their hot loop through a real call (item_direct_call):
// refreshed in stream order before every launch that uses it.
int refresh_d_dt(jfgpu_table* t) {
  if(!t->d_dt) HIP_TRY(hipMalloc((void**)&t->d_dt, sizeof(DevTable)));
  HIP_TRY(hipMemcpyAsync(t->d_dt, &t->dt, sizeof(DevTable), hipMemcpyHostToDevice, t->stream));
  return JFGPU_OK;
}



size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Grow the arena to at least `need` bytes.  Only legal while it holds nothing (ws_used == 0).
int ws_grow(jfgpu_table* t, size_t need) {
  if(need <= t->ws_cap) return JFGPU_OK;
  size_t want = std::max(need + need / 8, t->ws_cap * 2);
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipStreamSynchronize(t->stream));
  if(t->ws) { hipFree(t->ws); t->ws = nullptr; t->ws_cap = 0; }
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const size_t keep = (size_t)2 << 30;                      // leave room for the caller's buffers
  if(want + keep > free_b) want = need;
  if(want + keep / 2 > free_b) return -1;
  HIP_TRY(hipMalloc((void**)&t->ws, want));
  t->ws_cap = want;
  return JFGPU_OK;
}

void* ws_alloc(jfgpu_table* t, size_t bytes) {
  const size_t at = align_up(t->ws_used, 256);
  if(at + bytes > t->ws_cap) return nullptr;
  t->ws_used = at + bytes;
  return t->ws + at;
}

template <typename ITEM>
void launch_p1(jfgpu_table* t, bool scatter, bool from_keys, const uint8_t* base, int64_t lo, int64_t hi,
               const uint64_t* d_off, void* d_items) {
  const dim3 grid(t->g1), block(kPBlock);
  ITEM* out = (ITEM*)d_items;
#define P1(SC, FK, RT, BL) hipLaunchKernelGGL((p1_kernel<ITEM, SC, FK, RT, BL>), grid, block, 0, t->stream, t->dt, t->pg, base, lo, hi, t->d_M1, d_off, out)
  const bool rt = t->returning, bl = t->dt.bloom.data != nullptr && !from_keys;   // the filter applies to the sequence feed only
  // the common key widths get kernels with the byte count of the hash compiled in (no per-k-mer switch)
#define P1N(FK, N) hipLaunchKernelGGL((p1_kernel<ITEM, false, FK, false, false, N>), grid, block, 0, t->stream, t->dt, t->pg, base, lo, hi, t->d_M1, d_off, out)
  if(!scatter && !bl && t->g.nbytes >= 6) {
    if(from_keys) { if(t->g.nbytes == 6) P1N(true, 6); else if(t->g.nbytes == 7) P1N(true, 7); else P1N(true, 8); }
    else { if(t->g.nbytes == 6) P1N(false, 6); else if(t->g.nbytes == 7) P1N(false, 7); else P1N(false, 8); }
    return;
  }
#undef P1N
  if(!scatter) { if(from_keys) P1(false, true, false, false); else if(bl) P1(false, false, false, true); else P1(false, false, false, false); }
  else if(from_keys) { if(rt) P1(true, true, true, false); else P1(true, true, false, false); }
  else if(bl) { if(rt) P1(true, false, true, true); else P1(true, false, false, true); }
  else { if(rt) P1(true, false, true, false); else P1(true, false, false, false); }
#undef P1
}

int part_flush(jfgpu_table* t);

// per-workgroup straggler lists of the ring P1 kernels (kernels_p1ring.hip.hpp), allocated at first use
// JFGPU_FLUSH_TRACE: how full the blocks' straggler lists of a p2_ring_kernel launch ran (waits for the stream)
int trace_strag_lists(hipStream_t stream, const uint32_t* d_n, uint32_t n_lists, uint32_t list_cap, uint32_t per_bucket, uint32_t bucket0) {
  std::vector<uint32_t> hn(n_lists);
  HIP_TRY(hipStreamSynchronize(stream));
  HIP_TRY(hipMemcpy(hn.data(), d_n, n_lists * sizeof(uint32_t), hipMemcpyDeviceToHost));
  uint64_t sum = 0; uint32_t mx = 0, at_cap = 0, over256 = 0;
  for(uint32_t v : hn) { sum += v; mx = std::max(mx, v); at_cap += v >= list_cap; over256 += v > 256; }
  fprintf(stderr, "[jfgpu flush] P2 ring stragglers: %llu in %u lists, max %u, %u lists > 256, %u lists full\n", (unsigned long long)sum, n_lists, mx, over256, at_cap);
  for(uint32_t l = 0, shown = 0; l < n_lists && shown < 8; ++l)
    if(hn[l] >= list_cap) { fprintf(stderr, "[jfgpu flush]   full list %u (bucket %u, block %u)\n", l, bucket0 + l / per_bucket, l % per_bucket); ++shown; }
  return JFGPU_OK;
}

// Single-pass P2 of one bucket group of 4-byte items through per-destination rings (kernels_p1ring.hip.hpp): one workgroup
// per bucket with loader and storer waves when that fills the chip (p2_ring_roles_kernel), else several workgroups per
// bucket sharing the regions through reservations (p2_ring_kernel); then what did not go through a ring is appended to its
// region by the straggler kernel, before the regions' bounds are taken.  Rings hold 32 items: the kernels are for buckets
// of 1024 destinations (512 with the loader / storer kernel's short rounds) -- p2_rings_fit says so, else the sort.
constexpr uint32_t kG2Blocks = 4;      // workgroups per bucket of the single-pass P2 kernels that share regions
bool p2_rings_roles(const jfgpu_table* t, uint32_t b2e, uint32_t nbk) { return t->tun.p2_ring != 3 && nbk >= 2 * (uint32_t)t->n_cu && (b2e == 10 || b2e == 9); }
bool p2_rings_fit(const jfgpu_table* t, uint32_t b2e, uint32_t nbk) { return t->tun.p2_ring && (b2e == 10 || p2_rings_roles(t, b2e, nbk)); }
int launch_p2_rings(jfgpu_table* t, uint32_t b2e, uint32_t tag_bits, const SegList& S1, uint32_t cap2, unsigned int* d_gcur2, uint32_t n_dest, uint32_t* out_v,
                    uint32_t b0, uint32_t nbk, bool rt) {
  const bool roles = p2_rings_roles(t, b2e, nbk);
  const uint32_t n_lists = roles ? nbk : kG2Blocks * nbk;
  if(!t->d_strag2 || t->strag2_lists < n_lists) {
    if(t->d_strag2) { hipFree(t->d_strag2); hipFree(t->d_strag2_n); t->d_strag2 = nullptr; t->d_strag2_n = nullptr; }
    HIP_TRY(hipMalloc((void**)&t->d_strag2, (size_t)n_lists * kP2StragPerBlock * sizeof(uint64_t)));
    HIP_TRY(hipMalloc((void**)&t->d_strag2_n, (size_t)n_lists * sizeof(uint32_t)));
    t->strag2_lists = n_lists;
  }
  const P2RingDirect pd{t->d_dt, t->pg.b2, b2e, (int)rt};
  unsigned long long* ctr = (unsigned long long*)&t->dt.counters[CTR_DIRECT];
  const size_t lds = ((size_t)1 << b2e) * 128 + 128;
  if(roles) ++t->n_p2_roles; else ++t->n_p2_ring;
#define P2R(NV, PD) hipLaunchKernelGGL((p2_ring_roles_kernel<uint32_t, NV, P2RingDirect, PD>), dim3(nbk), dim3(kPBlock), lds, t->stream, pd, b2e, tag_bits, S1, cap2, d_gcur2, out_v, b0, t->d_strag2, t->d_strag2_n, ctr)
  if(roles && b2e == 10) { if(t->tun.p2_depth == 1) P2R(2, 1); else if(t->tun.p2_depth == 2) P2R(2, 2); else P2R(2, 3); }
  else if(roles) { if(t->tun.p2_depth == 1) P2R(1, 1); else if(t->tun.p2_depth == 2) P2R(1, 2); else P2R(1, 3); }
#undef P2R
  else
    hipLaunchKernelGGL((p2_ring_kernel<P2RingDirect>), dim3(kG2Blocks, nbk), dim3(kPBlock), lds, t->stream, pd, b2e, tag_bits, S1, cap2, d_gcur2, d_gcur2 + n_dest,
                       out_v, b0, (unsigned long long*)nullptr, t->d_strag2, t->d_strag2_n, ctr);
  hipLaunchKernelGGL((p1_stragglers_kernel<uint32_t, P2RingDirect>), dim3(t->n_cu), dim3(256), 0, t->stream, pd, ctr, (const uint64_t*)t->d_strag2, (const uint32_t*)t->d_strag2_n,
                     n_lists, cap2, d_gcur2, (unsigned long long*)nullptr, out_v, kP2StragPerBlock);
  if(t->tun.flush_trace) return trace_strag_lists(t->stream, t->d_strag2_n, n_lists, kP2StragPerBlock, roles ? 1 : kG2Blocks, b0);
  return JFGPU_OK;
}

int ensure_strag(jfgpu_table* t) {
  if(t->d_strag) return JFGPU_OK;
  const size_t words = Ring<uint32_t>::kWords;
  HIP_TRY(hipMalloc((void**)&t->d_strag, (size_t)t->n_cu * kStragPerBlock * words * sizeof(uint64_t)));
  HIP_TRY(hipMalloc((void**)&t->d_strag_n, (size_t)t->n_cu * sizeof(uint32_t)));
  return JFGPU_OK;
}

// Single-pass P1 (p1_ring_kernel, p1_keys_granule_kernel, ...): items per bucket region, 0 when the batch takes the exact
// two-pass scheme.  Every block may strand part of one reservation per bucket, so small batches would be
// mostly holes: auto mode wants the mean bucket load to be at least 4x that.
uint32_t granule_cap(const jfgpu_table* t, bool from_keys, uint64_t max_items) {
  if(t->item128) { if(t->pg.b1 > 10 || !t->g1) return 0; }              // the only P1 two-word keys have
  else if(t->pg.b2 == 0 || t->pg.b1 > 10 || t->tun.p1_single == 0 || !t->g1) return 0;
  else if(!t->item32 && from_keys) return 0;                             // 64-bit items: single-pass from sequence only
  const uint64_t nb = 1ull << t->pg.b1, strand = (uint64_t)t->g1 * kGran;
  // Sequence input: one item per byte is the upper bound, what earlier flushes saw per byte (+10 %) the estimate -- reads
  // of length L give (L - k + 1) / (L + 1) items per byte, 0.58 at k = 63.  An underestimate is safe: what does not fit
  // a region is inserted directly (granule_emit), it only costs speed.
  if(!from_keys && t->items_per_byte > 0) max_items = std::min<uint64_t>(max_items, (uint64_t)((double)max_items * (t->items_per_byte * 1.10 + 0.005)) + 4096);
  const uint64_t mean = (max_items + nb - 1) / nb;
  // a filtered pass (count --bc) stores few items, but the exact two-pass scheme would ask the filter twice per k-mer
  // (random reads: what the pass costs); stranded reservations are at most g1 * nb * kGran items per batch
  const bool filtered = !from_keys && (t->wide ? t->wt.bloom.data : t->dt.bloom.data) != nullptr;
  if(t->tun.p1_single < 0 && mean < 4 * strand && !filtered && !(t->item128 && t->mode == MODE_PARTITIONED)) return 0;
  const double want = (double)mean * (1.0 + t->tun.p1_slack) + (double)strand;
  uint64_t cap = want < (double)kGran ? kGran : (uint64_t)want;
  cap = (cap + kGran - 1) / kGran * kGran;
  if(cap > 0xFFFF0000ull) return 0;
  return (uint32_t)cap;
}

// count --bc: the cache of admitted k-mers (kernels_bloom.hip.hpp: bloom_cache_hit / _insert), 2^log2 two-way sets of
// 8-byte words, as much as the device has room for.  No room: the pass goes on without (state -1).
int bloom_cache_enable(jfgpu_table* t) {
  t->bcache_state = -1;
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  uint32_t lg = t->tun.bloom_cache_log2;
  while(lg > 16 && ((size_t)16 << lg) + ((size_t)6 << 30) > free_b) --lg;
  if(((size_t)16 << lg) + ((size_t)1 << 30) > free_b && lg > 16) return JFGPU_OK;
  if(hipMalloc((void**)&t->d_bcache, (size_t)16 << lg) != hipSuccess) { (void)hipGetLastError(); t->d_bcache = nullptr; return JFGPU_OK; }
  HIP_TRY(hipMemsetAsync(t->d_bcache, 0, (size_t)16 << lg, t->stream));
  t->dt.bloom.cache = t->d_bcache; t->dt.bloom.cache_mask = ((uint64_t)1 << lg) - 1;
  t->bcache_state = 1;
  if(t->tun.flush_trace) fprintf(stderr, "[jfgpu] count --bc: cache of admitted k-mers on, 2^%u sets (%.1f GB)\n", lg, (double)((size_t)16 << lg) / 1e9);
  return JFGPU_OK;
}
// Undecided and batches of a filtered pass are pending: how many of their windows did the counter admit?  One wait for
// the P1 launches already enqueued, once per attachment.  Uniform reads admit next to nothing (a cache would add a read
// per window: off); high-coverage reads admit most windows, each true k-mer dozens of times (on).
int bloom_cache_decide(jfgpu_table* t) {
  const uint32_t nb1 = 1u << t->pg.b1;
  std::vector<uint64_t> tot(nb1);
  uint64_t ctr[CTR_COUNT], admitted = 0;
  bool all_known = true;
  for(const PendingBatch& pb : t->pending) {
    if(!pb.gran_cap) { all_known = false; continue; }
    HIP_TRY(hipMemcpyAsync(tot.data(), pb.tot, nb1 * sizeof(uint64_t), hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    for(uint64_t v : tot) admitted += v;
  }
  HIP_TRY(hipMemcpyAsync(ctr, t->dt.counters, sizeof(ctr), hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  const uint64_t windows = ctr[CTR_MERS] >= t->mers_seen ? ctr[CTR_MERS] - t->mers_seen : 0;
  if(windows < (1u << 20) || !all_known) return JFGPU_OK;             // too little to tell: ask again at the next batch
  if(admitted * 100 < windows * 15) { t->bcache_state = -1; return JFGPU_OK; }
  return bloom_cache_enable(t);
}

// One batch (contract buffer or key array, on the device) through P1 into a pending batch.
int part_ingest(jfgpu_table* t, const uint8_t* base, int64_t lo, int64_t hi, bool from_keys, uint64_t max_items) {
  if(!max_items) return JFGPU_OK;
  const uint32_t nb = 1u << t->pg.b1;
  if(!t->d_M1) {
    t->g1 = 2 * t->n_cu;   // two 1024-thread blocks per CU: one stages while the other computes
    HIP_TRY(hipMalloc((void**)&t->d_M1, (size_t)t->g1 * kMaxBuckets * sizeof(uint32_t)));
  }
  if(t->pending.size() >= kMaxSeg) { int rc = part_flush(t); if(rc) return rc; }
  // 16-byte items: the arena is what limits the k-mers per flush, and every flush streams the whole table.  Before the
  // first flush the bucket regions are sized for one item per input byte; reads of length L give (L - k + 1) / (L + 1)
  // (0.58 at k = 63), so the first batch's exact count is read back once (one early wait for its P1) to size the others.
  if(t->item128 && !from_keys && t->items_per_byte <= 0 && !t->pending.empty()) {
    const PendingBatch& pb = t->pending.front();
    if(pb.gran_cap && pb.input_bytes >= (1u << 20)) {
      std::vector<uint64_t> tot(nb);
      HIP_TRY(hipMemcpyAsync(tot.data(), pb.tot, nb * sizeof(uint64_t), hipMemcpyDeviceToHost, t->stream));
      HIP_TRY(hipStreamSynchronize(t->stream));
      uint64_t n = 0; for(uint64_t v : tot) n += v;
      t->items_per_byte = std::max(0.01, (double)n / (double)pb.input_bytes);
    }
  }
  if(!from_keys && !t->wide && t->bcache_state == 0 && t->dt.bloom.data && !t->pending.empty()) { int rc = bloom_cache_decide(t); if(rc) return rc; }
  const uint32_t gcap = granule_cap(t, from_keys, max_items);
  if(t->item128 && (!gcap || from_keys)) return -1;       // two-word keys: single-pass P1 from sequence or the direct kernel
  // a one-pass Bloom filter (count --bf-size) changes as it is asked: the two-pass P1 would ask it twice per k-mer
  if(!gcap && !from_keys && (t->wide ? t->wt.bloom : t->dt.bloom).data && (t->wide ? t->wt.bloom : t->dt.bloom).kind == 1) return -1;
  const size_t bytes = gcap ? (size_t)nb * gcap * item_size(t) : max_items * item_size(t);
  const size_t need = align_up(bytes, 256) + align_up((2 * nb + 1) * sizeof(uint64_t), 256) + (gcap ? align_up(nb * 16, 256) : 0) + 1024;
  // 16-byte items: the arena may be smaller than what the whole input needs, so keep half of it for the flush's P2 output
  // (the flush's P2 output: an eighth of the arena is enough, P2 and the tile insert then go group by group: part_flush_t)
  const size_t ws_limit = t->item128 && t->pg.b2 ? t->ws_cap - t->ws_cap / 8 : t->ws_cap;
  if(t->item128 && !t->pending.empty() && t->ws_used + need > ws_limit) { int rc = part_flush(t); if(rc) return rc; }
  // ... and every flush streams the whole table, so when the announced input (jfgpu_reserve) needs n of them, make them
  // n equal ones: 6 + 3 + 1 batches cost a third more table traffic than 5 + 5
  if(t->item128 && !from_keys && t->reserved_input && !t->pending.empty() && need > 0) {
    const double per_flush = (double)ws_limit / (double)need * (double)(hi - lo);       // input bytes one flush can hold
    if(per_flush > 0 && (double)t->reserved_input > per_flush) {
      const double n = std::ceil((double)t->reserved_input / per_flush);
      uint64_t pend_in = 0;
      for(const PendingBatch& pb : t->pending) pend_in += pb.input_bytes;
      if((double)pend_in + (double)(hi - lo) > (double)t->reserved_input / n * 1.02) { int rc = part_flush(t); if(rc) return rc; }
    }
  }
  if(t->ws_used + need > t->ws_cap) {
    if(!t->pending.empty()) { int rc = part_flush(t); if(rc) return rc; }   // apply what is pending, arena is empty again
    if(need > t->ws_cap) { int rc = ws_grow(t, need); if(rc) return rc; }     // rc < 0: no memory -> caller goes direct
  }
  ++t->n_p1_other;                                             // (the ring kernel's branch below moves its launch to n_p1_ring)
  PendingBatch b{nullptr, nullptr, max_items};
  b.input_bytes = from_keys ? 0 : (uint64_t)(hi - lo);
  b.bound = t->cur_bound;
  b.items = ws_alloc(t, bytes);
  b.off = (uint64_t*)ws_alloc(t, (2 * nb + 1) * sizeof(uint64_t));
  if(!b.items || !b.off) return fail(JFGPU_E_ALLOC, "partition workspace exhausted");
  if(gcap) {
    // one pass: reservations of kGran items inside fixed bucket regions
    unsigned int* gcur = (unsigned int*)ws_alloc(t, nb * 16);           // gcur[2 nb] (u32) then tot[nb] (u64)
    if(!gcur) return fail(JFGPU_E_ALLOC, "partition workspace exhausted");
    b.gran_cap = gcap; b.tot = (unsigned long long*)(gcur + 2 * nb);
    HIP_TRY(hipMemsetAsync(gcur, 0, nb * 16, t->stream));
    ProfScope ps(t, 4, (uint64_t)(from_keys ? hi : hi - lo));
    const size_t lds = (size_t)kPTilePos * 6;
    const bool bl = (t->wide ? t->wt.bloom.data : t->dt.bloom.data) != nullptr && !from_keys;
    if(t->item128) {
      const size_t wlds = (size_t)kWideChunk * 18 + (size_t)t->g.nbytes * 2048;
#define PW(RT, BL) hipLaunchKernelGGL((p1_wide_granule_kernel<RT, BL>), dim3(t->g1), dim3(kPBlock), wlds, t->stream, t->wt, t->pg, base, lo, hi, gcap, gcur, b.tot, (u128*)b.items)
#define PWX(RT) hipLaunchKernelGGL((p1_wide_granule_kernel<RT, false, true>), dim3(t->g1), dim3(kPBlock), wlds, t->stream, t->wt, t->pg, base, lo, hi, gcap, gcur, b.tot, (u128*)b.items)
      if(t->g.hash_xs && !bl) { if(t->returning) PWX(true); else PWX(false); }
      else if(t->returning) { if(bl) PW(true, true); else PW(true, false); } else { if(bl) PW(false, true); else PW(false, false); }
#undef PWX
#undef PW
    } else if(!t->item32) {
      const size_t glds = (size_t)kG64Chunk * 10;
#define P64(RT, BL, N) hipLaunchKernelGGL((p1_granule64_kernel<RT, BL, N>), dim3(t->g1), dim3(kPBlock), glds, t->stream, t->dt, t->pg, base, lo, hi, gcap, gcur, b.tot, (uint64_t*)b.items)
      if(bl) { if(t->returning) P64(true, true, 0); else P64(false, true, 0); }
      else if(t->g.hash_xs) { if(t->returning) P64(true, false, kHashXS); else P64(false, false, kHashXS); }
      else if(t->returning) P64(true, false, 0);
      else if(t->g.nbytes == 8) P64(false, false, 8);
      else if(t->g.nbytes == 7) P64(false, false, 7);
      else P64(false, false, 0);
#undef P64
    } else
#define PK(RT, N) hipLaunchKernelGGL((p1_keys_granule_kernel<RT, N>), dim3(t->g1), dim3(kPBlock), lds, t->stream, t->dt, t->pg, (const uint64_t*)base, hi, gcap, gcur, b.tot, (uint32_t*)b.items)
    if(from_keys) {
      if(t->returning) PK(true, 0);
      else if(t->g.nbytes == 6) PK(false, 6);
      else if(t->g.nbytes == 7) PK(false, 7);
      else if(t->g.nbytes == 8) PK(false, 8);
      else PK(false, 0);
    } else
#define PG(IT, BL, N, CN) hipLaunchKernelGGL((p1_ring_kernel<IT, BL, N, CN>), dim3(t->n_cu), dim3(kPBlock), (size_t)nb * 128 + 128, t->stream, t->dt, od, t->pg, base, lo, hi, gcap, gcur, b.tot, (IT*)b.items, t->d_strag, t->d_strag_n)
    {
      // what cannot be stored in a region (p1_stragglers_kernel) reads the table's descriptor from device memory
      { int rc = refresh_d_dt(t); if(rc) return rc; }
      { int rc = ensure_strag(t); if(rc) return rc; }
      ++t->n_p1_ring; --t->n_p1_other;
      const OneWordDirect od{t->d_dt, t->pg.b2, (int)t->returning};
      unsigned long long* ctr = (unsigned long long*)&t->dt.counters[CTR_DIRECT];
      // (32-bit items exist for keys of at most 42 bits: six key bytes is the only width worth a compiled-in hash)
      // the xor-shift matrix is evaluated in registers (kmer_core.hpp: xs_hash); on the key's two dwords when the position has
      // more than 32 bits, which is the metric's geometry
      const bool xs = t->g.hash_xs && !bl, xs_hi = xs && t->g.lsize_g > 32 && t->g.lsize_g < 64 && t->g.lsize_l - t->pg.b1 < 32;
      if(xs_hi) { if(t->g.canonical) PG(uint32_t, false, kHashXS, 1); else PG(uint32_t, false, kHashXS, 0); }
      else if(xs) PG(uint32_t, false, kHashXSLow, 2);
      else if(t->g.nbytes == 6 && !bl) { if(t->g.canonical) PG(uint32_t, false, 6, 1); else PG(uint32_t, false, 6, 0); }
      else if(bl) PG(uint32_t, true, 0, 2);
      else PG(uint32_t, false, 0, 2);
      hipLaunchKernelGGL((p1_stragglers_kernel<uint32_t, OneWordDirect>), dim3(t->n_cu), dim3(256), 0, t->stream, od, ctr, (const uint64_t*)t->d_strag, (const uint32_t*)t->d_strag_n,
                         (uint32_t)t->n_cu, gcap, gcur, b.tot, (uint32_t*)b.items);
    }
#undef PG
#undef PK
    hipLaunchKernelGGL(granule_finish_kernel, dim3((nb + 255) / 256), dim3(256), 0, t->stream, gcur, gcap, nb, b.off);
  } else {
    ProfScope ps(t, 4, (uint64_t)(from_keys ? hi : hi - lo));
    if(t->item32) launch_p1<uint32_t>(t, false, from_keys, base, lo, hi, b.off, b.items);
    else          launch_p1<uint64_t>(t, false, from_keys, base, lo, hi, b.off, b.items);
    hipLaunchKernelGGL(scan_matrix_kernel, dim3(1), dim3(1024), 0, t->stream, t->d_M1, (uint32_t)t->g1, nb, (const uint64_t*)nullptr, b.off, 0u);
    if(t->item32 && !from_keys) {     // write-combining scatter (whole runs per bucket)
      const size_t lds = (size_t)kPTilePos * 6;
      const bool bl = t->dt.bloom.data != nullptr;
#define PS(RT, BL) hipLaunchKernelGGL((p1_scatter_sorted_kernel<RT, BL>), dim3(t->g1), dim3(kPBlock), lds, t->stream, t->dt, t->pg, base, lo, hi, (const uint32_t*)t->d_M1, (const uint64_t*)b.off, (uint32_t*)b.items)
#define PSN(N) hipLaunchKernelGGL((p1_scatter_sorted_kernel<false, false, N>), dim3(t->g1), dim3(kPBlock), lds, t->stream, t->dt, t->pg, base, lo, hi, (const uint32_t*)t->d_M1, (const uint64_t*)b.off, (uint32_t*)b.items)
      if(!t->returning && !bl && t->g.nbytes >= 6) { if(t->g.nbytes == 6) PSN(6); else if(t->g.nbytes == 7) PSN(7); else PSN(8); }
      else if(t->returning) { if(bl) PS(true, true); else PS(true, false); } else { if(bl) PS(false, true); else PS(false, false); }
#undef PSN
#undef PS
    }
    else if(t->item32) {              // encoded keys, 32-bit items: same write-combining scatter
#define PK(N) hipLaunchKernelGGL(p1_keys_scatter_sorted_kernel<N>, dim3(t->g1), dim3(kPBlock), (size_t)kPTilePos * 6, t->stream, t->dt, t->pg, \
                                 (const uint64_t*)base, hi, (const uint32_t*)t->d_M1, (const uint64_t*)b.off, (uint32_t*)b.items)
      if(t->g.nbytes == 6) PK(6); else if(t->g.nbytes == 7) PK(7); else if(t->g.nbytes == 8) PK(8); else PK(0);
#undef PK
    }
    else launch_p1<uint64_t>(t, true, from_keys, base, lo, hi, b.off, b.items);
  }
  hipError_t e = hipGetLastError();
  t->pending.push_back(b);
  t->pending_bytes += bytes;
  if(e != hipSuccess) return fail(JFGPU_E_HIP, hipGetErrorString(e));
  return JFGPU_OK;
}

// ---- a flush, first part: what is pending, bucket by bucket ------------------------------------------------------------
// offs: for every pending batch the running sum of its nb1 bucket sizes (nb1 + 1 entries); bucket_tot: items per P1 bucket
// over all batches.  One small copy per batch, which also drains the stream.  Side effects: the capacity accounting is
// corrected from "one k-mer per input byte" to the exact counts, and items_per_byte (what sizes the next batches' regions)
// is refreshed.
struct FlushSizes { std::vector<uint64_t> offs, bucket_tot; uint64_t total = 0; };
int flush_sizes(jfgpu_table* t, FlushSizes& fs) {
  const uint32_t nb1 = 1u << t->pg.b1;
  const size_t nbatch = t->pending.size();
  // bucket sizes of every pending batch (one small D2H; also drains the stream)
  // (granule batches: the exact per-bucket counts, stored as a running sum so both kinds read alike)
  std::vector<uint64_t>& offs = fs.offs;
  offs.assign(nbatch * (nb1 + 1), 0);
  for(size_t s = 0; s < nbatch; ++s) {
    const PendingBatch& pb = t->pending[s];
    if(pb.gran_cap) HIP_TRY(hipMemcpyAsync(&offs[s * (nb1 + 1) + 1], pb.tot, nb1 * sizeof(uint64_t), hipMemcpyDeviceToHost, t->stream));
    else HIP_TRY(hipMemcpyAsync(&offs[s * (nb1 + 1)], pb.off, (nb1 + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, t->stream));
  }
  uint64_t ctr[CTR_COUNT];
  HIP_TRY(hipMemcpyAsync(ctr, t->dt.counters, sizeof(ctr), hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  for(size_t s = 0; s < nbatch; ++s)
    if(t->pending[s].gran_cap) {
      uint64_t* o = &offs[s * (nb1 + 1)];
      o[0] = 0;
      for(uint32_t j = 0; j < nb1; ++j) o[j + 1] += o[j];
    }
  std::vector<uint64_t>& bucket_tot = fs.bucket_tot;
  bucket_tot.assign(nb1, 0);
  uint64_t total = 0, max_bucket = 0;
  for(size_t s = 0; s < nbatch; ++s)
    for(uint32_t j = 0; j < nb1; ++j) bucket_tot[j] += offs[s * (nb1 + 1) + j + 1] - offs[s * (nb1 + 1) + j];
  for(uint32_t j = 0; j < nb1; ++j) { total += bucket_tot[j]; max_bucket = std::max(max_bucket, bucket_tot[j]); }
  if(max_bucket > 0xF0000000ull) return fail(JFGPU_E_UNSUPPORTED, "more than 2^32 pending k-mers in one partition bucket: sync more often");
  {   // capacity accounting (ensure_capacity): these batches were charged one k-mer per input byte, now their exact item
      // counts are known -- plus whatever went straight to the table since the last look
    uint64_t charged = 0, exact = 0;
    for(size_t s = 0; s < nbatch; ++s) if(t->pending[s].bound) { charged += t->pending[s].bound; exact += offs[s * (nb1 + 1) + nb1]; }
    const uint64_t dd = ctr[CTR_DIRECT] >= t->direct_seen ? ctr[CTR_DIRECT] - t->direct_seen : 0;
    t->direct_seen = ctr[CTR_DIRECT];
    if(charged && exact + dd < charged) t->fed_since -= std::min(t->fed_since, charged - (exact + dd));
  }
  {   // items per input byte of what is being flushed (sequence batches only): sizes the next batches' bucket regions
    uint64_t in_bytes = 0, in_items = 0;
    for(size_t s = 0; s < nbatch; ++s)
      if(t->pending[s].input_bytes) { in_bytes += t->pending[s].input_bytes; in_items += offs[s * (nb1 + 1) + nb1]; }
    if(in_bytes >= (1u << 20)) t->items_per_byte = (double)in_items / (double)in_bytes;
  }
  fs.total = total;
  if(t->bcache_state == 0) {       // count --bc, cache of admitted k-mers still undecided: this flush's batches tell (bloom_cache_decide's rule)
    const uint64_t windows = ctr[CTR_MERS] >= t->mers_seen ? ctr[CTR_MERS] - t->mers_seen : 0;
    t->mers_seen = ctr[CTR_MERS];
    if(!t->wide && t->dt.bloom.data && windows >= (1u << 20)) {
      if(total * 100 < windows * 15) t->bcache_state = -1;
      else { const int rc = bloom_cache_enable(t); if(rc) return rc; }
    }
  }
  return JFGPU_OK;
}

// ---- the tile insert of one-word keys (kernels_tile.hip.hpp) --------------------------------------------------------
// One instantiation of tile_rank_insert_kernel over tiles [tile0, tile0 + TPB * ntile): HV the HEAVY variant, SM the plain
// one with its two sample counters on.
template <typename ITEM, typename SLOT, int TPB, bool HV, bool SM>
void launch_tile_rank_variant(jfgpu_table* t, const SegList& S, uint64_t tile0, uint32_t ntile, hipStream_t ts) {
  const size_t lds = tile_rank_lds(sizeof(SLOT), t->g.tile_bits, TPB);
  const dim3 grid((unsigned)std::min<uint64_t>(ntile, (uint64_t)t->n_cu * 16)), block(kTileBlock);
  if(t->returning) hipLaunchKernelGGL((tile_rank_insert_kernel<ITEM, true, SLOT, TPB, kTileBlock, HV, SM>), grid, block, lds, ts, t->dt, S, tile0, ntile);
  else             hipLaunchKernelGGL((tile_rank_insert_kernel<ITEM, false, SLOT, TPB, kTileBlock, HV, SM>), grid, block, lds, ts, t->dt, S, tile0, ntile);
}
// Which instantiation (plain, or HEAVY for high-coverage input) is decided from the flush itself: the first 64th of a
// large launch's units goes through the plain kernel with its counters on, the host reads them (one wait inside the
// flush) and the rest follows in the instantiation they call for.  JFGPU_TILE_ADAPT=0: always plain; 2: always HEAVY (tests).
template <typename ITEM, typename SLOT, int TPB>
void launch_tile_rank(jfgpu_table* t, const SegList& S, uint64_t tile0, uint32_t ntile, hipStream_t ts) {
  const int adapt = t->tun.tile_adapt;
  bool heavy = adapt == 2;
  uint32_t done = 0;
  if(adapt == 1 && ntile >= 8192 && S.n == 1) {
    const uint32_t ns = ntile / 64;
    (void)hipMemsetAsync(&t->dt.counters[CTR_T_ITEMS], 0, 2 * sizeof(uint64_t), ts);
    launch_tile_rank_variant<ITEM, SLOT, TPB, false, true>(t, S, tile0, ns, ts);
    uint64_t smp[2] = {0, 0};                              // items placed, items past rank 3
    if(hipMemcpyAsync(smp, &t->dt.counters[CTR_T_ITEMS], sizeof smp, hipMemcpyDeviceToHost, ts) == hipSuccess && hipStreamSynchronize(ts) == hipSuccess)
      heavy = smp[1] * 4 > smp[0];
    done = ns;
  }
  SegList Sr = S; Sr.off[0] = S.off[0] + ((size_t)done << S.sh[0]);
  if(heavy) ++t->flushes_heavy; else ++t->flushes_plain;
  if(heavy) launch_tile_rank_variant<ITEM, SLOT, TPB, true, false>(t, Sr, tile0 + (uint64_t)TPB * done, ntile - done, ts);
  else      launch_tile_rank_variant<ITEM, SLOT, TPB, false, false>(t, Sr, tile0 + (uint64_t)TPB * done, ntile - done, ts);
}

template <typename ITEM>
int part_flush_t(jfgpu_table* t) {
  const uint32_t nb1 = 1u << t->pg.b1, nb2 = 1u << t->pg.b2;
  const size_t nbatch = t->pending.size();
  FlushSizes fs;
  { const int rc = flush_sizes(t, fs); if(rc) return rc; }
  const std::vector<uint64_t>& offs = fs.offs;
  const std::vector<uint64_t>& bucket_tot = fs.bucket_tot;
  const uint64_t total = fs.total;
  const uint64_t n_tiles = n_tiles_of(t);
  if constexpr(!(sizeof(ITEM) == 16)) { int rc = refresh_d_dt(t); if(rc) return rc; }
  SegList S1; memset(&S1, 0, sizeof S1);
  S1.n = (uint32_t)nbatch;
  for(size_t s = 0; s < nbatch; ++s) { S1.items[s] = t->pending[s].items; S1.off[s] = t->pending[s].off; S1.sh[s] = t->pending[s].gran_cap ? 1 : 0; }
  constexpr bool kWideItems = sizeof(ITEM) == 16;         // two-word keys: 128-bit items, 128-bit slots
  const size_t tile_lds = (size_t)(kWideItems ? 16 : t->g.slot32 ? 4 : 8) << t->g.tile_bits;
  const bool rt = t->returning;      // a tile is read only if its dirty byte is set (clean after jfgpu_clear)
  // the tile insert of one item array (or of the pending batches themselves), on stream `ts`
  auto launch_tile_kernel = [&](const SegList& S, uint64_t tile0, uint32_t ntile, hipStream_t ts, bool pair = false) {
    if constexpr(kWideItems) {
      const dim3 block(kPBlock), grid((unsigned)std::min<uint64_t>(ntile, (uint64_t)t->n_cu * 4));     // 128 KiB of LDS: one block per CU
      if(S.n == 1 && t->tun.wide_pipe) {        // one item array (a flush's P2 output): the pipelined kernel
        const dim3 gridp((unsigned)std::min<uint64_t>(ntile, (uint64_t)t->n_cu));
        if(rt) hipLaunchKernelGGL(tile_insert_wide_pipe_kernel<true>, gridp, block, tile_lds, ts, t->wt, S, tile0, ntile);
        else   hipLaunchKernelGGL(tile_insert_wide_pipe_kernel<false>, gridp, block, tile_lds, ts, t->wt, S, tile0, ntile);
      }
      else if(rt) hipLaunchKernelGGL(tile_insert_wide_kernel<true>, grid, block, tile_lds, ts, t->wt, S, tile0, ntile);
      else   hipLaunchKernelGGL(tile_insert_wide_kernel<false>, grid, block, tile_lds, ts, t->wt, S, tile0, ntile);
    } else {
      // one-word keys: placement by rank inside buckets of four (kernels_tile.hip.hpp); two workgroups per CU
      if(t->g.slot32) { if(pair) launch_tile_rank<ITEM, unsigned int, 2>(t, S, tile0, ntile, ts); else launch_tile_rank<ITEM, unsigned int, 1>(t, S, tile0, ntile, ts); }
      else launch_tile_rank<ITEM, unsigned long long, 1>(t, S, tile0, ntile, ts);
    }
  };
  auto launch_tiles = [&](const SegList& S, uint64_t tile0, uint32_t ntile, uint64_t units) {
    ProfScope ps(t, 6, units);
    launch_tile_kernel(S, tile0, ntile, t->stream);
  };
  if(total == 0) {
    // nothing to insert
  } else if(t->mode != MODE_PARTITIONED && total < n_tiles * 256) {
    // too few items to be worth streaming the tiles: global atomics straight from the items
    for(size_t s = 0; s < nbatch; ++s) {
      const uint64_t n = offs[s * (nb1 + 1) + nb1];
      if(!n) continue;
      ProfScope ps(t, 7, n);
      const uint64_t span = t->pending[s].gran_cap ? (uint64_t)nb1 * t->pending[s].gran_cap : n;
      const dim3 grid((unsigned)grid_for(t, (span + kBlock - 1) / kBlock)), block(kBlock);
      const uint64_t gc = t->pending[s].gran_cap;
      if constexpr(kWideItems) {
        if(rt) hipLaunchKernelGGL(items_direct_wide_kernel<true>, grid, block, 0, t->stream, t->wt, t->pg, (const u128*)t->pending[s].items, (const uint64_t*)t->pending[s].off, gc);
        else   hipLaunchKernelGGL(items_direct_wide_kernel<false>, grid, block, 0, t->stream, t->wt, t->pg, (const u128*)t->pending[s].items, (const uint64_t*)t->pending[s].off, gc);
      } else {
        if(rt) hipLaunchKernelGGL((items_direct_kernel<ITEM, true>), grid, block, 0, t->stream, t->dt, t->pg, (const ITEM*)t->pending[s].items, (const uint64_t*)t->pending[s].off, gc);
        else   hipLaunchKernelGGL((items_direct_kernel<ITEM, false>), grid, block, 0, t->stream, t->dt, t->pg, (const ITEM*)t->pending[s].items, (const uint64_t*)t->pending[s].off, gc);
      }
    }
  } else if(t->pg.b2 == 0) {
    // single-level table (at most 2^11 tiles): the P1 buckets are the tiles; one pass over them per pending batch
    if constexpr(kWideItems) launch_tiles(S1, 0, nb1, total);
    else for(size_t s = 0; s < nbatch; ++s) {
      const uint64_t n = offs[s * (nb1 + 1) + nb1];
      if(!n) continue;
      SegList Sb; memset(&Sb, 0, sizeof Sb);
      Sb.n = 1; Sb.items[0] = S1.items[s]; Sb.off[0] = S1.off[s]; Sb.sh[0] = S1.sh[s];
      launch_tiles(Sb, 0, nb1, n);
    }
  } else {
    // breadth-first: every P1 bucket through P2 in one launch per pass, then every tile in one launch
    const int g2 = 32;
    if(!t->d_M2) HIP_TRY(hipMalloc((void**)&t->d_M2, (size_t)nb1 * g2 * nb2 * sizeof(uint32_t)));
    ITEM* tmp = nullptr; uint64_t *d_goff = nullptr, *d_base = nullptr;
    bool tmp_owned = false;
    // 32-bit items into 32-bit slots: P2 routes to pairs of adjacent tiles (half as many destinations, and chunks of 28 Ki
    // items), so the runs it writes are ~112 bytes on average instead of 32; the tile kernel owns a pair (64 KiB) in LDS
    const bool pair = sizeof(ITEM) == 4 && t->g.slot32 && t->pg.b2 >= 1 && t->tun.tile_pair;
    // Single-pass P2 (32-bit items into pairs of tiles; 16-byte items into tiles): fixed regions of cap2 items per
    // destination, reservations of kGran items (p2_granule_kernel).  Worth it when the regions are mostly items: every
    // block may strand one reservation per destination.  When the regions of the whole table do not fit the arena beside
    // what is pending, the P1 buckets go through P2 and the tile insert in `single_groups` groups sharing one buffer.
    constexpr uint32_t kG2Single = 4;                       // blocks per P1 bucket
    constexpr bool kSingleItems = sizeof(ITEM) == 4 || sizeof(ITEM) == 8 || sizeof(ITEM) == 16;
    uint32_t cap2 = 0, single_groups = 1; unsigned int* d_gcur2 = nullptr; uint64_t* d_off2 = nullptr; ITEM* out2 = nullptr; bool own2 = false;
    const bool single_ok = kSingleItems && t->tun.p2_single && (sizeof(ITEM) >= 8 || pair) && t->tun.flush_groups <= 1;      // (4-byte items: pairs only; 8- and 16-byte items: single tiles)
    const uint64_t n_dest = pair ? n_tiles >> 1 : n_tiles;
    // The ring kernels of P2 read every all-ones item as a hole and load 16 bytes at a time: right for granule batches (fixed
    // regions, holes marked so), wrong for an exact two-pass batch, which stores every item -- the all-ones one too when
    // items are 32 bits wide -- at arbitrary offsets (round-4 advisor finding).  The loader / storer kernel takes such a
    // segment item by item (load_exact); a flush holding one keeps the sort-based kernel where the shared-ring kernel would run.
    bool all_granule = true;
    for(size_t s = 0; s < nbatch; ++s) all_granule = all_granule && t->pending[s].gran_cap != 0;
    if(single_ok) {
      // what a destination's region may lose to reservations nobody fills: one granule per block -- two with the ring kernel,
      // whose owners ask for the next reservation a round ahead -- and the holes behind the blocks' last units
      const uint64_t mean = total / n_dest, strand = (uint64_t)kG2Single * kGran * (sizeof(ITEM) == 4 && t->tun.p2_ring ? 2 : 1) + (sizeof(ITEM) == 4 && t->tun.p2_ring ? kGran : 0);
      // ... except with the loader / storer kernel: one workgroup owns a bucket's regions, nothing is reserved, a region
      // loses at most its last partial unit -- worth it from a few units per destination (a 155 Mbp sample in the
      // metric's 2^34-slot table takes the timed job's kernels: bench.py's digest check against the reference)
      const bool roles_geom = sizeof(ITEM) == 4 && pair && p2_rings_fit(t, t->pg.b2 - 1, nb1) && p2_rings_roles(t, t->pg.b2 - 1, nb1);
      if(mean >= (roles_geom ? 32 : 8 * strand) || t->tun.p2_single > 1) {
        // head-room over the mean load: a pair of tiles takes ~8 K items a flush, 1 % standard deviation on uniform reads --
        // but on high-coverage input its ~100 hot k-mers come 80 times each (10 %), and what overflows a region is
        // inserted with global atomics: with 8 % head-room P2 took 38 ms on distribution G instead of 29
        const double slack = t->tun.p2_slack >= 0 ? t->tun.p2_slack : (sizeof(ITEM) <= 8 ? 0.30 : 0.08);
        cap2 = (uint32_t)(((uint64_t)((double)mean * (1.0 + slack)) + strand + 2 * kGran - 1) / kGran * kGran);
        if(t->tun.p2_cap) cap2 = t->tun.p2_cap;
        const size_t mark = t->ws_used;
        d_gcur2 = (unsigned int*)ws_alloc(t, 2 * n_dest * sizeof(unsigned int));
        d_off2 = (uint64_t*)ws_alloc(t, 2 * n_dest * sizeof(uint64_t));
        const size_t used = align_up(t->ws_used, 256) + 4096;
        const size_t free_b = t->ws_cap > used ? t->ws_cap - used : 0;
        const uint32_t forced = t->tun.flush_share;      // (tests)
        uint32_t G = forced && forced <= nb1 / 8 ? forced : 1;
        while(!forced && G <= nb1 / 8 && (n_dest / G) * cap2 * sizeof(ITEM) > free_b) G *= 2;
        if(d_gcur2 && d_off2 && G <= std::max<uint32_t>(1, nb1 / 8)) out2 = (ITEM*)ws_alloc(t, (n_dest / G) * cap2 * sizeof(ITEM));
        if(out2) single_groups = G;
        else {                                   // no room for the regions in the arena: exact P2 (forced: one-off allocations)
          t->ws_used = mark;
          d_gcur2 = nullptr; d_off2 = nullptr; out2 = nullptr;
          single_groups = forced && forced <= nb1 / 8 ? forced : 1;
          if(t->tun.p2_single > 1 && hipMalloc((void**)&d_gcur2, 2 * n_dest * sizeof(unsigned int)) == hipSuccess &&
             hipMalloc((void**)&d_off2, 2 * n_dest * sizeof(uint64_t)) == hipSuccess &&
             hipMalloc((void**)&out2, (n_dest / single_groups) * cap2 * sizeof(ITEM)) == hipSuccess) own2 = true;
          else { if(d_gcur2) hipFree(d_gcur2); if(d_off2) hipFree(d_off2); if(out2) hipFree(out2); cap2 = 0; single_groups = 1; }
        }
        if(cap2) HIP_TRY(hipMemsetAsync(d_gcur2, 0, 2 * n_dest * sizeof(unsigned int), t->stream));
      }
    }
    if(t->tun.flush_trace)
      fprintf(stderr, "[flush] %llu items in %zu batches, %llu tiles, pairs %d, single-pass P2 regions of %u items (0: exact P2) in %u group(s)\n",
              (unsigned long long)total, nbatch, (unsigned long long)n_tiles, (int)pair, cap2, single_groups);
    // When the arena cannot hold a P2 output of the whole flush beside what is pending, the P1 buckets go through P2 and
    // the tile insert in groups that share one small output buffer (group g's tiles are inserted before group g+1 is
    // partitioned): 16-byte items at 10 Gbp then need 104 GB (P1 regions) + 12 GB instead of 104 + 94, the whole job fits
    // one flush, and the table is streamed once instead of twice.
    uint32_t share_groups = 0;
    uint64_t tmp_items = std::max<uint64_t>(total, 1);
    if(!cap2) {
      const size_t fixed = align_up((n_tiles + 1) * sizeof(uint64_t), 256) + align_up(nb1 * sizeof(uint64_t), 256) + 1024;
      const size_t used = align_up(t->ws_used, 256);
      const size_t free_b = t->ws_cap > used + fixed ? t->ws_cap - used - fixed : 0;
      const uint32_t forced = t->tun.flush_share;      // (tests)
      if((total * sizeof(ITEM) > free_b || forced) && t->tun.flush_groups <= 1) {
        for(uint32_t G = 2; G <= nb1 / 8; G *= 2) {
          uint64_t mx = 0;
          for(uint32_t g = 0; g < G; ++g) { uint64_t sum = 0; for(uint32_t j = g * (nb1 / G); j < (g + 1) * (nb1 / G); ++j) sum += bucket_tot[j]; mx = std::max(mx, sum); }
          if(forced ? G == forced : mx * sizeof(ITEM) <= free_b) { share_groups = G; tmp_items = std::max<uint64_t>(mx, 1); break; }
        }
      }
      d_goff = (uint64_t*)ws_alloc(t, (n_tiles + 1) * sizeof(uint64_t));
      d_base = (uint64_t*)ws_alloc(t, nb1 * sizeof(uint64_t));
      tmp = (ITEM*)ws_alloc(t, tmp_items * sizeof(ITEM));
      if(!d_goff || !d_base || !tmp) {     // arena too small for the flush temporaries: one-off allocation
        if(!forced) { share_groups = 0; tmp_items = std::max<uint64_t>(total, 1); }
        tmp_owned = true;
        tmp = nullptr; d_goff = nullptr; d_base = nullptr;
        HIP_TRY(hipMalloc((void**)&tmp, tmp_items * sizeof(ITEM)));
        if(hipMalloc((void**)&d_goff, (n_tiles + 1) * sizeof(uint64_t)) != hipSuccess ||
           hipMalloc((void**)&d_base, nb1 * sizeof(uint64_t)) != hipSuccess) {
          hipFree(tmp); if(d_goff) hipFree(d_goff);
          return fail(JFGPU_E_ALLOC, "hipMalloc partition offsets");
        }
      }
      std::vector<uint64_t> base(nb1);                     // where a bucket's tiles start in tmp (groups sharing tmp: from 0 again)
      { uint64_t run = 0; for(uint32_t j = 0; j < nb1; ++j) { if(share_groups && j % (nb1 / share_groups) == 0) run = 0; base[j] = run; run += bucket_tot[j]; } }
      HIP_TRY(hipMemcpyAsync(d_base, base.data(), nb1 * sizeof(uint64_t), hipMemcpyHostToDevice, t->stream));
    }
    if(t->tun.flush_trace && share_groups) fprintf(stderr, "[flush] P2 + tile insert in %u groups sharing an output buffer of %llu items\n", share_groups, (unsigned long long)tmp_items);
    // Optionally (JFGPU_FLUSH_GROUPS > 1) the P1 buckets go through P2 and the tile insert in groups, P2 on the
    // table's stream and the tile insert on a second one, so that group g's tiles are inserted while group g+1
    // is partitioned (one P2-scatter block, 88 KB LDS, and one tile block, 64 KB, fit a CU together).
    if(cap2 && single_groups > 1) share_groups = single_groups;      // same loop, same per-group timers; the shared buffer holds regions
    const uint32_t n_groups = share_groups ? share_groups : t->tun.flush_groups > 1 && nb1 >= (uint32_t)t->tun.flush_groups * 8 ? (uint32_t)t->tun.flush_groups : 1;
    const bool two_streams = n_groups > 1 && !share_groups;
    const uint32_t gsz = nb1 / n_groups;
    if(two_streams && !t->stream2) {
      HIP_TRY(hipStreamCreateWithFlags(&t->stream2, hipStreamNonBlocking));
      for(auto& ev : t->flush_ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&t->flush_done, hipEventDisableTiming));
    }
    hipEvent_t p2a = nullptr, p2b = nullptr, ta = nullptr, tb = nullptr;
    if(t->prof_on && !share_groups) { p2a = get_event(t); p2b = get_event(t); ta = get_event(t); tb = get_event(t); hipEventRecord(p2a, t->stream); }
    constexpr int per_thread = sizeof(ITEM) == 4 ? 16 : sizeof(ITEM) == 8 ? 14 : 7;     // 8- and 16-byte items: chunks of 112 KiB (longer runs per destination)
    PartGeom pg2 = t->pg;
    if(pair) pg2.b2 -= 1;
    const uint32_t nb2e = 1u << pg2.b2;
    const uint32_t p2_tag_bits = (kWideItems ? t->wt.W.tag_full : t->g.tag_bits) + (pair ? 1 : 0);      // where an item's P2 sub-bucket starts
    for(uint32_t g = 0; g < n_groups; ++g) {
      const uint32_t b0 = g * gsz, nbk = g + 1 == n_groups ? nb1 - b0 : gsz;
      const dim3 grid(g2, nbk), block(kPBlock);
      hipEvent_t ga = nullptr;                              // groups sharing tmp: every group's two stages are timed on their own
      if(t->prof_on && share_groups) { ga = get_event(t); hipEventRecord(ga, t->stream); }
      if constexpr(kSingleItems) {
        if(cap2) {
          const dim3 g1p(kG2Single, nbk);
          const uint64_t d0 = (uint64_t)b0 << pg2.b2, nd = (uint64_t)nbk << pg2.b2;
          // destination d of the whole table sits at d * cap2 of a buffer that only holds this group's: shifted base
          ITEM* out_v = out2 - (share_groups ? (int64_t)d0 * (int64_t)cap2 : 0);
          if constexpr(sizeof(ITEM) == 4) {
            const size_t lds = (size_t)kPBlock * kP2PairPer * sizeof(ITEM);
            if(p2_rings_fit(t, pg2.b2, nbk) && (all_granule || p2_rings_roles(t, pg2.b2, nbk))) {
              const int rc_ = launch_p2_rings(t, pg2.b2, p2_tag_bits, S1, cap2, d_gcur2, (uint32_t)n_dest, (uint32_t*)out_v, b0, nbk, rt);
              if(rc_) return rc_;
            } else {
              ++t->n_p2_sort;
              if(rt) hipLaunchKernelGGL((p2_granule_kernel<uint32_t, TableDirect<true>, kP2PairPer>), g1p, block, lds, t->stream, TableDirect<true>{t->d_dt, t->pg, (unsigned long long*)&t->dt.counters[CTR_DIRECT]}, pg2.b2, p2_tag_bits, S1, cap2, d_gcur2, d_gcur2 + n_dest, (uint32_t*)out_v, b0);
              else   hipLaunchKernelGGL((p2_granule_kernel<uint32_t, TableDirect<false>, kP2PairPer>), g1p, block, lds, t->stream, TableDirect<false>{t->d_dt, t->pg, (unsigned long long*)&t->dt.counters[CTR_DIRECT]}, pg2.b2, p2_tag_bits, S1, cap2, d_gcur2, d_gcur2 + n_dest, (uint32_t*)out_v, b0);
            }
          } else if constexpr(sizeof(ITEM) == 8) {     // keys of 22 to 32 bases: 8-byte items into single tiles
            const size_t lds = (size_t)kPBlock * kP2MidPer * sizeof(ITEM);        // chunks of 112 KiB
            ++t->n_p2_sort;
            if(rt) hipLaunchKernelGGL((p2_granule_kernel<uint64_t, TableDirect<true>, kP2MidPer>), g1p, block, lds, t->stream, TableDirect<true>{t->d_dt, t->pg, (unsigned long long*)&t->dt.counters[CTR_DIRECT]}, pg2.b2, p2_tag_bits, S1, cap2, d_gcur2, d_gcur2 + n_dest, (uint64_t*)out_v, b0);
            else   hipLaunchKernelGGL((p2_granule_kernel<uint64_t, TableDirect<false>, kP2MidPer>), g1p, block, lds, t->stream, TableDirect<false>{t->d_dt, t->pg, (unsigned long long*)&t->dt.counters[CTR_DIRECT]}, pg2.b2, p2_tag_bits, S1, cap2, d_gcur2, d_gcur2 + n_dest, (uint64_t*)out_v, b0);
          } else {
            const size_t lds = (size_t)kPBlock * kP2WidePer * sizeof(ITEM);
            ++t->n_p2_sort;
            if(rt) hipLaunchKernelGGL((p2_granule_kernel<u128, WideDirect<true>, kP2WidePer>), g1p, block, lds, t->stream, WideDirect<true>{t->wt, t->pg}, pg2.b2, p2_tag_bits, S1, cap2, d_gcur2, d_gcur2 + n_dest, (u128*)out_v, b0);
            else   hipLaunchKernelGGL((p2_granule_kernel<u128, WideDirect<false>, kP2WidePer>), g1p, block, lds, t->stream, WideDirect<false>{t->wt, t->pg}, pg2.b2, p2_tag_bits, S1, cap2, d_gcur2, d_gcur2 + n_dest, (u128*)out_v, b0);
          }
          // (granule_finish_kernel reads gcur[nb + j] as destination j's overflow note: the two halves of d_gcur2)
          if(n_groups == 1) hipLaunchKernelGGL(granule_finish_kernel, dim3(1024), dim3(256), 0, t->stream, d_gcur2, cap2, (uint32_t)n_dest, d_off2);
          else hipLaunchKernelGGL(granule_finish_range_kernel, dim3(256), dim3(256), 0, t->stream, d_gcur2, cap2, (uint32_t)n_dest, d_off2, (uint32_t)d0, (uint32_t)nd);
          SegList S2; memset(&S2, 0, sizeof S2);
          S2.n = 1; S2.items[0] = out_v; S2.off[0] = d_off2 + 2 * d0; S2.sh[0] = 1;     // offsets from the group's first destination, items absolute
          hipStream_t ts = t->stream;
          if(share_groups) {    // one stream, P2 and T alternate (the next group overwrites the regions): timed per group
            uint64_t gtot = 0; for(uint32_t j = b0; j < b0 + nbk; ++j) gtot += bucket_tot[j];
            if(t->prof_on) {
              hipEvent_t gb = get_event(t), gc = get_event(t), gd = get_event(t);
              hipEventRecord(gb, ts); hipEventRecord(gc, ts);
              t->prof_pending.push_back({ga, gb, 5, gtot});
              launch_tile_kernel(S2, (uint64_t)b0 << t->pg.b2, (uint32_t)nd, ts, pair);
              hipEventRecord(gd, ts);
              t->prof_pending.push_back({gc, gd, 6, gtot});
            } else launch_tile_kernel(S2, (uint64_t)b0 << t->pg.b2, (uint32_t)nd, ts, pair);
            continue;
          }
          if(t->prof_on && g + 1 == n_groups) hipEventRecord(p2b, t->stream);
          if(two_streams) {
            HIP_TRY(hipEventRecord(t->flush_ev[g & 1], t->stream));
            HIP_TRY(hipStreamWaitEvent(t->stream2, t->flush_ev[g & 1], 0));
            ts = t->stream2;
          }
          if(t->prof_on && g == 0) hipEventRecord(ta, ts);
          launch_tile_kernel(S2, (uint64_t)b0 << t->pg.b2, (uint32_t)nd, ts, pair);
          if(t->prof_on && g + 1 == n_groups) hipEventRecord(tb, ts);
          continue;
        }
      }
      ++t->n_p2_exact;
      hipLaunchKernelGGL((p2_kernel<ITEM, false>), grid, block, 0, t->stream, pg2, p2_tag_bits, S1, t->d_M2, (const uint64_t*)d_goff, tmp, b0);
      hipLaunchKernelGGL(scan_matrix_kernel, dim3(nbk), dim3(1024), 0, t->stream, t->d_M2, (uint32_t)g2, nb2e, (const uint64_t*)d_base, d_goff, b0);
      bool launched = false;
      if constexpr(sizeof(ITEM) == 4) {
        if(pair) {
          hipLaunchKernelGGL((p2_scatter_sorted_kernel<ITEM, kP2PairPer>), grid, block, (size_t)kPBlock * kP2PairPer * sizeof(ITEM), t->stream,
                             pg2, p2_tag_bits, S1, (const uint32_t*)t->d_M2, (const uint64_t*)d_goff, tmp, b0);
          launched = true;
        }
      }
      if(!launched)
        hipLaunchKernelGGL((p2_scatter_sorted_kernel<ITEM, per_thread>), grid, block, (size_t)kPBlock * per_thread * sizeof(ITEM), t->stream,
                           pg2, p2_tag_bits, S1, (const uint32_t*)t->d_M2, (const uint64_t*)d_goff, tmp, b0);
      if(t->prof_on && g + 1 == n_groups && !share_groups) hipEventRecord(p2b, t->stream);
      const uint64_t tile_start = (uint64_t)b0 << t->pg.b2;
      const uint32_t ntile = nbk << pg2.b2;                 // units of the tile kernel: tiles, or pairs of tiles
      SegList S2; memset(&S2, 0, sizeof S2);
      S2.n = 1; S2.items[0] = tmp; S2.off[0] = d_goff + ((uint64_t)b0 << pg2.b2);
      hipStream_t ts = t->stream;
      if(two_streams) {
        HIP_TRY(hipEventRecord(t->flush_ev[g & 1], t->stream));
        HIP_TRY(hipStreamWaitEvent(t->stream2, t->flush_ev[g & 1], 0));
        ts = t->stream2;
      }
      if(share_groups) {      // one stream, P2 and T alternate: time every group's two stages separately
        uint64_t gtot = 0; for(uint32_t j = b0; j < b0 + nbk; ++j) gtot += bucket_tot[j];
        if(t->prof_on) {
          hipEvent_t gb = get_event(t), gc = get_event(t), gd = get_event(t);
          hipEventRecord(gb, ts); hipEventRecord(gc, ts);
          t->prof_pending.push_back({ga, gb, 5, gtot});
          launch_tile_kernel(S2, tile_start, ntile, ts, pair);
          hipEventRecord(gd, ts);
          t->prof_pending.push_back({gc, gd, 6, gtot});
        } else launch_tile_kernel(S2, tile_start, ntile, ts, pair);
        continue;
      }
      if(t->prof_on && g == 0) hipEventRecord(ta, ts);
      launch_tile_kernel(S2, tile_start, ntile, ts, pair);
      if(t->prof_on && g + 1 == n_groups) hipEventRecord(tb, ts);
    }
    if(t->prof_on && !share_groups) {
      t->prof_pending.push_back({p2a, p2b, 5, total});
      t->prof_pending.push_back({ta, tb, 6, total});
    }
    if(two_streams) {        // the table's stream continues only after the last tiles are in
      HIP_TRY(hipEventRecord(t->flush_done, t->stream2));
      HIP_TRY(hipStreamWaitEvent(t->stream, t->flush_done, 0));
    }
    hipError_t e = hipStreamSynchronize(t->stream);
    if(tmp_owned) { hipFree(tmp); hipFree(d_goff); hipFree(d_base); }
    if(own2) { hipFree(d_gcur2); hipFree(d_off2); hipFree(out2); }
    if(e != hipSuccess) return fail(JFGPU_E_HIP, hipGetErrorString(e));
  }
  hipError_t e = hipGetLastError();
  if(e == hipSuccess) e = hipStreamSynchronize(t->stream);
  t->pending.clear(); t->pending_bytes = 0; t->ws_used = 0;
  if(total) t->pristine = false;
  if(e != hipSuccess) return fail(JFGPU_E_HIP, hipGetErrorString(e));
  return JFGPU_OK;
}

int part_flush(jfgpu_table* t) {
  if(t->pending.empty()) return JFGPU_OK;
  const int rc = t->item128 ? part_flush_t<u128>(t) : t->item32 ? part_flush_t<uint32_t>(t) : part_flush_t<uint64_t>(t);
  if(rc) {      // what was pending is lost with the failed flush: the table no longer holds what it was fed, and says so until
    if(t->stream) hipStreamSynchronize(t->stream);      // it is cleared (jfgpu_clear), instead of carrying on with k-mers missing
    t->pending.clear(); t->pending_bytes = 0; t->ws_used = 0;
    t->failed = rc; t->failed_msg = jfgpu_last_error();
  }
  return rc;
}

// Replace the charges of the pending batches (one k-mer per input byte) by their exact item counts: one wait for the P1
// kernels already enqueued, no flush.  The batches keep the corrected charge, so the flush's own correction stays neutral.
int refine_pending_charges(jfgpu_table* t) {
  if(t->pending.empty()) return JFGPU_OK;
  const uint32_t nb1 = 1u << t->pg.b1;
  std::vector<size_t> idx;
  for(size_t s = 0; s < t->pending.size(); ++s) if(t->pending[s].bound) idx.push_back(s);
  if(idx.empty()) return JFGPU_OK;
  std::vector<uint64_t> buf(idx.size() * nb1);
  uint64_t ctr[CTR_COUNT];
  for(size_t i = 0; i < idx.size(); ++i) {
    const PendingBatch& pb = t->pending[idx[i]];
    if(pb.gran_cap) HIP_TRY(hipMemcpyAsync(&buf[i * nb1], pb.tot, nb1 * sizeof(uint64_t), hipMemcpyDeviceToHost, t->stream));
    else HIP_TRY(hipMemcpyAsync(&buf[i * nb1], pb.off + nb1, sizeof(uint64_t), hipMemcpyDeviceToHost, t->stream));
  }
  HIP_TRY(hipMemcpyAsync(ctr, t->dt.counters, sizeof(ctr), hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  uint64_t charged = 0, exact = 0;
  for(size_t i = 0; i < idx.size(); ++i) {
    PendingBatch& pb = t->pending[idx[i]];
    uint64_t n = 0;
    if(pb.gran_cap) for(uint32_t j = 0; j < nb1; ++j) n += buf[i * nb1 + j]; else n = buf[i * nb1];
    charged += pb.bound; exact += std::min(n, pb.bound);
    pb.bound = std::max<uint64_t>(std::min(n, pb.bound), 1);
  }
  const uint64_t dd = ctr[CTR_DIRECT] >= t->direct_seen ? ctr[CTR_DIRECT] - t->direct_seen : 0;
  t->direct_seen = ctr[CTR_DIRECT];
  if(exact + dd < charged) t->fed_since -= std::min(t->fed_since, charged - (exact + dd));
  else t->fed_since += (exact + dd) - charged;
  return JFGPU_OK;
}

void part_discard(jfgpu_table* t) {
  if(t->stream) hipStreamSynchronize(t->stream);
  t->pending.clear(); t->pending_bytes = 0; t->ws_used = 0;
}


