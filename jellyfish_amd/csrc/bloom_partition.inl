// jellyfish_amd/csrc/bloom_partition.inl -- host orchestration of the partitioned Bloom insert
// (kernels_bloom_part.hip.hpp): geometry, workspace arena, P1b ingestion in pieces, the flush (P2 + segment kernel).
// Included by abi_bloom.inl inside its anonymous namespace.

constexpr size_t kBloomMinPiece = (size_t)1 << 20;        // bytes of sequence: smaller pieces take the direct kernel
enum BloomSlot { BS_DIRECT = 0, BS_P1 = 1, BS_P2 = 2, BS_SEG = 3, BS_P2RING = 4, BS_COUNT = 5 };      // (BS_P2RING: launches of the ring P2 only, inside BS_P2's time)

struct BloomProf {      // HIP-event pair around a group of launches on the Bloom counter's stream
  jfgpu_bloom* b; int which; uint64_t units; hipEvent_t a = nullptr, e = nullptr;
  BloomProf(jfgpu_bloom* b_, int w, uint64_t u) : b(b_), which(w), units(u) {
    if(b->prof_on) { hipEventCreate(&a); hipEventCreate(&e); hipEventRecord(a, b->stream); }
  }
  ~BloomProf() { if(b->prof_on) { hipEventRecord(e, b->stream); b->spans.push_back({a, e, which, units}); } }
};

void bloom_prof_collect(jfgpu_bloom* b) {
  for(auto& s : b->spans) {
    float ms = 0;
    hipEventSynchronize(s.b);
    if(hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { b->prof_ms[s.which] += ms; b->prof_launches[s.which] += 1; b->prof_units[s.which] += s.units; }
    hipEventDestroy(s.a); hipEventDestroy(s.b);
  }
  b->spans.clear();
}

// Segments, bucket bits.  part_ok == false: the direct kernel is the only path (tiny or > 128 GiB filters, two-word keys).
void bloom_part_init(jfgpu_bloom* b) {
  b->part_ok = false;
  const uint64_t n_seg = (b->data_bytes + ((1ull << kBloomSegBits) - 1)) >> kBloomSegBits;
  if(b->wide || n_seg == 0 || n_seg > (1ull << 21)) return;
  uint32_t sb = 0;
  while((1ull << sb) < n_seg) ++sb;
  uint32_t b1, b2;
  if(sb <= 10) { b1 = sb; b2 = 0; }
  else { b2 = std::min<uint32_t>(11, (sb + 1) / 2); b1 = sb - b2; }
  if(b1 > 10) return;
  b->bp.b1 = b1; b->bp.b2 = b2; b->bp.n_seg = (uint32_t)n_seg; b->bp.pad_ = 0;
  b->part_ok = true;
}

bool bloom_use_partitioned(const jfgpu_bloom* b, size_t nbytes) {
  if(!b->part_ok || b->mode == 1) return false;
  if(b->mode == 2) return true;
  return b->bp.n_seg >= 64 && (nbytes >= kBloomMinPiece || !b->pending.empty());
}

void* bloom_ws_alloc(jfgpu_bloom* b, size_t bytes) {
  const size_t at = align_up(b->ws_used, 256);
  if(at + bytes > b->ws_cap) return nullptr;
  b->ws_used = at + bytes;
  return b->ws + at;
}

int bloom_ws_ensure(jfgpu_bloom* b, size_t want) {
  if(b->ws && (want == 0 || b->ws_cap >= want)) return JFGPU_OK;
  if(!b->pending.empty()) return JFGPU_OK;                    // never reallocate under pending batches
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipStreamSynchronize(b->stream));
  if(b->ws) { hipFree(b->ws); b->ws = nullptr; b->ws_cap = 0; }
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const size_t keep = (size_t)4 << 30;
  size_t cap = want ? want : (size_t)96 << 30;
  if(free_b < keep + ((size_t)256 << 20)) return -1;
  cap = std::min(cap, free_b - keep);
  if(hipMalloc((void**)&b->ws, cap) != hipSuccess) { (void)hipGetLastError(); b->ws = nullptr; return -1; }
  b->ws_cap = cap; b->ws_used = 0;
  HIP_TRY(hipMemsetAsync(b->ws, 0, cap, b->stream));          // first touch now, not inside the first pass
  return JFGPU_OK;
}

int bloom_flush(jfgpu_bloom* b);

// Region capacity (items per P1 bucket) for a piece of n sequence bytes, and the arena bytes it takes.
uint32_t bloom_region_cap(const jfgpu_bloom* b, uint64_t n_bytes) {
  const uint64_t nb = 1ull << b->bp.b1, strand = (uint64_t)b->g1 * kGran;
  const uint64_t mean = (n_bytes * b->nh + nb - 1) / nb;
  uint64_t cap = (uint64_t)((double)mean * (1.0 + b->slack)) + strand + kGran;
  cap = (cap + kGran - 1) / kGran * kGran;
  return cap > 0xFFFF0000ull ? 0 : (uint32_t)cap;
}
size_t bloom_piece_bytes(const jfgpu_bloom* b, uint32_t cap) {
  const size_t nb = (size_t)1 << b->bp.b1;
  return align_up(nb * cap * sizeof(uint32_t), 256) + align_up((2 * nb + 1) * sizeof(uint64_t), 256) + align_up(nb * 16, 256) + 1024;
}

int bloom_launch_direct(jfgpu_bloom* b, const uint8_t* base, int64_t lo, int64_t hi) {
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(n_tiles, (int64_t)b->n_cu * 8));
  BloomProf ps(b, BS_DIRECT, (uint64_t)(hi - lo));
  if(b->wide) hipLaunchKernelGGL(bloom_insert_ascii_wide_kernel, dim3(grid), dim3(kBlock), 0, b->stream, b->view(), b->wg, base, lo, hi, b->d_mers);
  else hipLaunchKernelGGL(bloom_insert_ascii_kernel, dim3(grid), dim3(kBlock), 0, b->stream, b->view(), b->g, base, lo, hi, b->d_mers);
  HIP_TRY(hipGetLastError());
  return JFGPU_OK;
}

// One contract buffer [lo, hi) through P1b, in pieces that fit half of the arena (the other half is the flush's).
int bloom_ingest(jfgpu_bloom* b, const uint8_t* base, int64_t lo, int64_t hi) {
  if(!b->g1) {
    b->g1 = b->tun.bloom_p1_two ? 2 * b->n_cu : b->n_cu;      // workgroups of P1b: two per CU (78 KB of LDS each) or one (136 KB)
    // The ring P1b (p1_bloom_ring_kernel): one workgroup per CU, every owner lane may strand two reservations per bucket
    // (g1 stays the regions' head-room unit: 2 x n_cu).  It wants its rounds to put ~25 cell updates on a ring of 64:
    // ten cells a lane and round with >= 400 buckets in use, five with >= 200; smaller filters keep the sort-based kernels.
    const uint32_t used = (uint32_t)(((uint64_t)b->bp.n_seg + ((1u << b->bp.b2) - 1)) >> b->bp.b2);
    // (JFGPU_BLOOM_P1_RING=2 / 3: rounds of ten / five whatever the filter -- tests: on a small filter the rings overflow all
    //  the time, so the lists, their overflow into global compare-and-swaps and exhausted regions are all on the path)
    const int ring = b->tun.bloom_p1_ring;
    if(ring && b->bp.b1 <= 9 && b->bp.b2 > 0 && (used >= 200 || ring >= 2) && b->g.nbytes <= 8) {
      b->p1_ring_per = ring == 2 ? 10 : ring == 3 ? 5 : used >= 400 ? 10 : 5;
      b->g1 = 2 * b->n_cu;
      if(!b->d_strag1) {
        HIP_TRY(hipMalloc((void**)&b->d_strag1, (size_t)b->n_cu * kStragPerBlock * sizeof(uint64_t)));
        HIP_TRY(hipMalloc((void**)&b->d_strag1_n, (size_t)b->n_cu * sizeof(uint32_t)));
      }
    }
  }
  int rc = bloom_ws_ensure(b, 0);
  if(rc < 0) return bloom_launch_direct(b, base, lo, hi);      // no memory for an arena
  if(rc) return rc;
  const uint32_t nb = 1u << b->bp.b1, k = b->g.k;
  const size_t lds = (size_t)kBloomChunk * 6 + (size_t)2 * b->g.nbytes * 2048;
  int64_t at = lo;
  while(hi - at >= (int64_t)k) {
    // largest piece whose regions fit what is left of the arena's pending part: 7/8 of it -- the flush's P2 output takes
    // the rest, P1b bucket group by bucket group (bloom_flush_inner), so a flush holds twice the sequence it used to
    // and the 28 GB array is streamed half as often
    const size_t half = b->ws_cap - b->ws_cap / 8;
    const size_t room = half > align_up(b->ws_used, 256) ? half - align_up(b->ws_used, 256) : 0;
    uint64_t piece = (uint64_t)(hi - at);
    uint32_t cap = bloom_region_cap(b, piece);
    if(!cap || bloom_piece_bytes(b, cap) > room || b->pending.size() >= kMaxSeg) {
      const uint64_t fixed = (uint64_t)nb * ((uint64_t)b->g1 * kGran + 2 * kGran) * 4 + (uint64_t)nb * 48 + 4096;
      uint64_t fit = room > fixed ? (uint64_t)((double)(room - fixed) / (4.0 * (1.0 + b->slack) * b->nh)) : 0;
      if(b->pending.size() >= kMaxSeg) fit = 0;
      if(fit < std::min<uint64_t>(piece, kBloomMinPiece)) {
        if(!b->pending.empty()) { rc = bloom_flush(b); if(rc) return rc; continue; }   // arena empty again: retry
        if(fit < (uint64_t)k) {                                 // arena too small even when empty: global atomics
          rc = bloom_launch_direct(b, base, at, hi); if(rc) return rc;
          return JFGPU_OK;
        }
      }
      piece = std::min<uint64_t>(piece, fit);
      cap = bloom_region_cap(b, piece);
      if(!cap || bloom_piece_bytes(b, cap) > room) { piece = piece / 2 + k; cap = bloom_region_cap(b, piece); }
      if(!cap || bloom_piece_bytes(b, cap) > room) return fail(JFGPU_E_ALLOC, "Bloom partition workspace exhausted");
    }
    BloomPending p;
    p.cap = cap;
    p.items = (uint32_t*)bloom_ws_alloc(b, (size_t)nb * cap * sizeof(uint32_t));
    p.off = (uint64_t*)bloom_ws_alloc(b, (2 * (size_t)nb + 1) * sizeof(uint64_t));
    unsigned int* gcur = (unsigned int*)bloom_ws_alloc(b, (size_t)nb * 16);
    if(!p.items || !p.off || !gcur) return fail(JFGPU_E_ALLOC, "Bloom partition workspace exhausted");
    p.tot = (unsigned long long*)(gcur + 2 * nb);
    HIP_TRY(hipMemsetAsync(gcur, 0, (size_t)nb * 16, b->stream));
    const int64_t pe = at + (int64_t)piece;
    // the kernel walks tiles from its base pointer: rebase on this piece (16-byte aligned) instead of skipping [0, at)
    const uint8_t* pbase = base + (at & ~(int64_t)15);
    const int64_t plo = at & 15, phi = plo + (int64_t)piece;
    {
      BloomProf ps(b, BS_P1, piece);
#define PB(N) hipLaunchKernelGGL(p1_bloom_granule_kernel<N>, dim3(b->g1), dim3(kPBlock), lds, b->stream, b->view(), b->bp, b->g, pbase, plo, phi, cap, gcur, p.tot, p.items, b->d_mers)
#define PB2(N) hipLaunchKernelGGL(p1_bloom_granule2_kernel<N>, dim3(b->g1), dim3(kPBlock), (size_t)kPBlock * 5 * 6 + (size_t)b->g.nbytes * 512, b->stream, \
                                  b->view(), b->bp, b->g, pbase, plo, phi, cap, gcur, p.tot, p.items, b->d_mers)
#define PBR(N, PER) hipLaunchKernelGGL((p1_bloom_ring_kernel<N, PER>), dim3(b->n_cu), dim3(kPBlock), (size_t)nb * kBloomRingBytes + kBloomRingBytes + (size_t)b->g.nbytes * 512, b->stream, \
                                       b->view(), b->bp, b->g, rd, pbase, plo, phi, cap, gcur, p.tot, p.items, b->d_mers, b->d_strag1, b->d_strag1_n)
      if(b->p1_ring_per) {
        const BloomP1RingDirect rd{b->view().data, b->bp.b2};
        if(b->p1_ring_per == 10) { if(b->g.nbytes == 8) PBR(8, 10); else if(b->g.nbytes == 6) PBR(6, 10); else PBR(0, 10); }
        else { if(b->g.nbytes == 8) PBR(8, 5); else PBR(0, 5); }
        hipLaunchKernelGGL((p1_stragglers_kernel<uint32_t, BloomP1RingDirect>), dim3(b->n_cu), dim3(256), 0, b->stream, rd, (unsigned long long*)nullptr, (const uint64_t*)b->d_strag1,
                           (const uint32_t*)b->d_strag1_n, (uint32_t)b->n_cu, cap, gcur, p.tot, p.items);
      }
      else if(b->tun.bloom_p1_two) { if(b->g.nbytes == 8) PB2(8); else if(b->g.nbytes == 6) PB2(6); else PB2(0); }
      else if(b->g.nbytes == 8) PB(8); else if(b->g.nbytes == 6) PB(6); else PB(0);
#undef PBR
#undef PB2
#undef PB
      hipLaunchKernelGGL(granule_finish_kernel, dim3((nb + 255) / 256), dim3(256), 0, b->stream, gcur, cap, nb, p.off);
    }
    HIP_TRY(hipGetLastError());
    b->pending.push_back(p);
    if(pe >= hi) break;
    at = pe - (int64_t)(k - 1);                                 // next piece re-reads k-1 characters: every window once
  }
  return JFGPU_OK;
}

int bloom_flush_inner(jfgpu_bloom* b) {
  const uint32_t nb1 = 1u << b->bp.b1, nb2 = 1u << b->bp.b2;
  const size_t nbatch = b->pending.size();
  std::vector<uint64_t> tots(nbatch * nb1);
  for(size_t s = 0; s < nbatch; ++s)
    HIP_TRY(hipMemcpyAsync(&tots[s * nb1], b->pending[s].tot, nb1 * sizeof(uint64_t), hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  std::vector<uint64_t> bucket_tot(nb1, 0);
  uint64_t total = 0, max_bucket = 0;
  for(size_t s = 0; s < nbatch; ++s) for(uint32_t j = 0; j < nb1; ++j) bucket_tot[j] += tots[s * nb1 + j];
  for(uint32_t j = 0; j < nb1; ++j) { total += bucket_tot[j]; max_bucket = std::max(max_bucket, bucket_tot[j]); }
  if(max_bucket > 0xF0000000ull) return fail(JFGPU_E_UNSUPPORTED, "more than 2^32 pending cell updates in one Bloom partition bucket");
  SegList S1; memset(&S1, 0, sizeof S1);
  S1.n = (uint32_t)nbatch;
  for(size_t s = 0; s < nbatch; ++s) { S1.items[s] = b->pending[s].items; S1.off[s] = b->pending[s].off; S1.sh[s] = 1; }
  const DevBloom B = b->view();
  const size_t seg_lds = (size_t)1 << kBloomSegBits;
  auto launch_segments = [&](const SegList& S, uint32_t nseg, uint64_t units, uint32_t seg0 = 0) {
    BloomProf ps(b, BS_SEG, units);
    const dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nseg, (uint64_t)b->n_cu * 2)));
    hipLaunchKernelGGL(bloom_segment_kernel, grid, dim3(kPBlock), seg_lds, b->stream, B, S, nseg, seg0);
  };
  if(total == 0) {
    // nothing
  } else if(b->mode != 2 && total < (uint64_t)b->bp.n_seg * 1024) {
    for(size_t s = 0; s < nbatch; ++s) {
      BloomProf ps(b, BS_DIRECT, 0);
      const uint64_t span = (uint64_t)nb1 * b->pending[s].cap;
      const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>((span + kBlock - 1) / kBlock, (uint64_t)b->n_cu * 8));
      hipLaunchKernelGGL(bloom_items_direct_kernel, dim3(grid), dim3(kBlock), 0, b->stream, B, b->bp, (const uint32_t*)b->pending[s].items,
                         (const uint64_t*)b->pending[s].off, (uint64_t)b->pending[s].cap);
    }
  } else if(b->bp.b2 == 0) {
    launch_segments(S1, std::min<uint32_t>(nb1, b->bp.n_seg), total);
  } else {
    const int g2 = 32;
    if(!b->d_M2) HIP_TRY(hipMalloc((void**)&b->d_M2, (size_t)nb1 * g2 * nb2 * sizeof(uint32_t)));
    const uint64_t n_tiles = (uint64_t)nb1 * nb2;
    uint64_t* d_goff = (uint64_t*)bloom_ws_alloc(b, (n_tiles + 1) * sizeof(uint64_t));
    uint64_t* d_base = (uint64_t*)bloom_ws_alloc(b, nb1 * sizeof(uint64_t));
    // the P2 output of the whole flush, or -- when that does not fit beside what is pending -- of one group of P1b buckets
    // at a time: the groups share the buffer, each group's segments are applied before the next group is partitioned
    const size_t free_b = b->ws_cap > align_up(b->ws_used, 256) + 4096 ? b->ws_cap - align_up(b->ws_used, 256) - 4096 : 0;
    uint32_t n_groups = 1;
    uint64_t tmp_items = std::max<uint64_t>(total, 1);
    const uint32_t forced = b->tun.flush_share;      // (tests)
    if(total * sizeof(uint32_t) > free_b || forced)
      for(uint32_t G = 2; G <= nb1 / 4; G *= 2) {
        uint64_t mx = 0;
        for(uint32_t g = 0; g < G; ++g) { uint64_t sum = 0; for(uint32_t j = g * (nb1 / G); j < (g + 1) * (nb1 / G); ++j) sum += bucket_tot[j]; mx = std::max(mx, sum); }
        if(forced ? G == forced : mx * sizeof(uint32_t) <= free_b) { n_groups = G; tmp_items = std::max<uint64_t>(mx, 1); break; }
      }
    // Single-pass P2 (p2_granule_kernel, like the count path's): fixed regions of cap2 cell updates per segment instead of a
    // count pass + exact placement -- the updates are read once instead of twice.  Regions of one bucket group at a time.
    constexpr uint32_t kG2Single = 4;
    uint32_t cap2 = 0;
    unsigned int* d_gcur2 = nullptr; uint64_t* d_off2 = nullptr; uint32_t* out2 = nullptr;
    const int p2_single = b->tun.p2_single;
    if(p2_single) {
      const bool ring2 = b->tun.p2_ring && nb2 == (uint32_t)kGranMaxB;      // (its owners hold a second reservation: see part_flush_t)
      const uint64_t mean = total / std::max<uint32_t>(1, b->bp.n_seg), strand = (uint64_t)kG2Single * kGran * (ring2 ? 2 : 1) + (ring2 ? kGran : 0);   // (the array ends before the last bucket does)
      if(mean >= 8 * strand || p2_single > 1) {
        cap2 = (uint32_t)(((uint64_t)((double)mean * 1.08) + strand + 2 * kGran - 1) / kGran * kGran);
        const size_t mark = b->ws_used;
        d_gcur2 = (unsigned int*)bloom_ws_alloc(b, (2 * n_tiles + 2) * sizeof(unsigned int));
        d_off2 = (uint64_t*)bloom_ws_alloc(b, 2 * n_tiles * sizeof(uint64_t));
        const size_t used = align_up(b->ws_used, 256) + 4096;
        const size_t fr = b->ws_cap > used ? b->ws_cap - used : 0;
        uint32_t G = forced && forced <= nb1 / 4 ? forced : 1;
        while(!forced && G <= nb1 / 4 && (n_tiles / G) * cap2 * sizeof(uint32_t) > fr) G *= 2;
        if(d_gcur2 && d_off2 && G <= std::max<uint32_t>(1, nb1 / 4)) out2 = (uint32_t*)bloom_ws_alloc(b, (n_tiles / G) * cap2 * sizeof(uint32_t));
        if(out2) { n_groups = G; HIP_TRY(hipMemsetAsync(d_gcur2, 0, (2 * n_tiles + 2) * sizeof(unsigned int), b->stream)); }
        else { b->ws_used = mark; cap2 = 0; d_gcur2 = nullptr; d_off2 = nullptr; }
      }
    }
    uint32_t* tmp = cap2 ? out2 : (uint32_t*)bloom_ws_alloc(b, tmp_items * sizeof(uint32_t));
    if(!d_goff || !d_base || !tmp) return fail(JFGPU_E_ALLOC, "Bloom partition workspace too small for the flush");
    const uint32_t gsz = nb1 / n_groups;
    if(b->tun.flush_trace) fprintf(stderr, "[jfgpu flush] Bloom: %zu batches, %llu cell updates, %u bucket groups of %u, regions of %u\n", nbatch, (unsigned long long)total, n_groups, gsz, cap2);
    std::vector<uint64_t> basev(nb1);
    { uint64_t run = 0; for(uint32_t j = 0; j < nb1; ++j) { if(j % gsz == 0) run = 0; basev[j] = run; run += bucket_tot[j]; } }
    HIP_TRY(hipMemcpyAsync(d_base, basev.data(), nb1 * sizeof(uint64_t), hipMemcpyHostToDevice, b->stream));
    PartGeom P; memset(&P, 0, sizeof P);
    P.b1 = b->bp.b1; P.b2 = b->bp.b2;
    for(uint32_t g = 0; g < n_groups; ++g) {
      const uint32_t b0 = g * gsz;
      uint64_t gtot = 0; for(uint32_t j = b0; j < b0 + gsz; ++j) gtot += bucket_tot[j];
      const uint64_t seg0 = (uint64_t)b0 * nb2;
      if(cap2) {
        uint32_t* out_v = out2 - (n_groups > 1 ? (int64_t)seg0 * (int64_t)cap2 : 0);      // segment d of the whole array sits at d * cap2
        {
          BloomProf ps(b, BS_P2, gtot);
          const dim3 block(kPBlock);
          const BloomDirect D{B, b->bp, reinterpret_cast<unsigned long long*>(d_gcur2 + 2 * n_tiles)};
          // The count path's ring kernel (kernels_p1ring.hip.hpp) -- a cell update names its segment in its own bits too --
          // for the buckets whose segments all exist.  The array ends inside the last bucket: its rounds of 8 Ki updates
          // fall on a fraction of the 1024 rings and overflow them (measured: that one bucket's four blocks, all through
          // the lists and global atomics, doubled the stage's time), so it keeps the sort-based kernel, like every bucket
          // when JFGPU_P2_RING=0.
          const uint32_t whole = (uint32_t)std::min<uint64_t>(b->bp.n_seg >> b->bp.b2, (uint64_t)b0 + gsz);   // buckets below this one are whole
          const uint32_t n_ring = b->tun.p2_ring && nb2 == (uint32_t)kGranMaxB && whole > b0 ? whole - b0 : 0;
          // ... through the loader / storer kernel (one workgroup per bucket, nothing reserved: the count path's 27.8 -> 19 ms
          // step of round 4) when the group gives every CU a bucket, else through the shared-ring kernel
          const bool roles = n_ring >= (uint32_t)b->n_cu && b->tun.p2_ring != 3;
          if(n_ring) {
            const uint32_t n_lists = roles ? n_ring : kG2Single * n_ring;
            if(!b->d_strag2 || b->strag2_lists < n_lists) {
              if(b->d_strag2) { hipFree(b->d_strag2); hipFree(b->d_strag2_n); b->d_strag2 = nullptr; b->d_strag2_n = nullptr; }
              HIP_TRY(hipMalloc((void**)&b->d_strag2, (size_t)n_lists * kP2StragPerBlock * sizeof(uint64_t)));
              HIP_TRY(hipMalloc((void**)&b->d_strag2_n, (size_t)n_lists * sizeof(uint32_t)));
              b->strag2_lists = n_lists;
            }
            const BloomRingDirect RD{B.data};
            ++b->prof_launches[BS_P2RING]; b->prof_units[BS_P2RING] += gtot;
            if(roles)
              hipLaunchKernelGGL((p2_ring_roles_kernel<uint32_t, 2, BloomRingDirect, 3>), dim3(n_ring), block, (size_t)nb2 * 128 + 128, b->stream, RD, b->bp.b2, kBloomItemLow, S1, cap2, d_gcur2,
                                 out_v, b0, b->d_strag2, b->d_strag2_n, D.counter);
            else
            hipLaunchKernelGGL((p2_ring_kernel<BloomRingDirect>), dim3(kG2Single, n_ring), block, (size_t)nb2 * 128 + 128, b->stream, RD, b->bp.b2, kBloomItemLow, S1, cap2, d_gcur2, d_gcur2 + n_tiles,
                               out_v, b0, (unsigned long long*)nullptr, b->d_strag2, b->d_strag2_n, D.counter);
            hipLaunchKernelGGL((p1_stragglers_kernel<uint32_t, BloomRingDirect>), dim3(b->n_cu), dim3(256), 0, b->stream, RD, D.counter, (const uint64_t*)b->d_strag2, (const uint32_t*)b->d_strag2_n,
                               n_lists, cap2, d_gcur2, (unsigned long long*)nullptr, out_v, kP2StragPerBlock);
            if(b->tun.flush_trace) { const int rc_ = trace_strag_lists(b->stream, b->d_strag2_n, n_lists, kP2StragPerBlock, roles ? 1 : kG2Single, b0); if(rc_) return rc_; }
          }
          if(n_ring < gsz && (uint64_t)(b0 + n_ring) * nb2 < b->bp.n_seg)
            hipLaunchKernelGGL((p2_granule_kernel<uint32_t, BloomDirect, kP2PairPer>), dim3(kG2Single, n_ring ? 1 : gsz), block, (size_t)kPBlock * kP2PairPer * sizeof(uint32_t), b->stream,
                               D, b->bp.b2, kBloomItemLow, S1, cap2, d_gcur2, d_gcur2 + n_tiles, out_v, b0 + n_ring);
          hipLaunchKernelGGL(granule_finish_range_kernel, dim3(256), dim3(256), 0, b->stream, d_gcur2, cap2, (uint32_t)n_tiles, d_off2, (uint32_t)seg0, gsz * nb2);
        }
        if(seg0 >= b->bp.n_seg) break;
        const uint32_t nseg = (uint32_t)std::min<uint64_t>((uint64_t)gsz * nb2, b->bp.n_seg - seg0);
        SegList S2; memset(&S2, 0, sizeof S2);
        S2.n = 1; S2.items[0] = out_v; S2.off[0] = d_off2 + 2 * seg0; S2.sh[0] = 1;
        launch_segments(S2, nseg, gtot, (uint32_t)seg0);
        continue;
      }
      {
        BloomProf ps(b, BS_P2, gtot);
        const dim3 grid(g2, gsz), block(kPBlock);
        hipLaunchKernelGGL((p2_kernel<uint32_t, false>), grid, block, 0, b->stream, P, kBloomItemLow, S1, b->d_M2, (const uint64_t*)d_goff, tmp, b0);
        hipLaunchKernelGGL(scan_matrix_kernel, dim3(gsz), dim3(1024), 0, b->stream, b->d_M2, (uint32_t)g2, nb2, (const uint64_t*)d_base, d_goff, b0);
        // chunks of 28 Ki cell updates: the runs written per destination are what this pass costs (scatter_write_probe)
        hipLaunchKernelGGL((p2_scatter_sorted_kernel<uint32_t, kP2PairPer>), grid, block, (size_t)kPBlock * kP2PairPer * sizeof(uint32_t), b->stream,
                           P, kBloomItemLow, S1, (const uint32_t*)b->d_M2, (const uint64_t*)d_goff, tmp, b0);
      }
      // this group's segments: numbers b0 * nb2 .. ; the last group stops at the array's last segment
      if(seg0 >= b->bp.n_seg) break;
      const uint32_t nseg = (uint32_t)std::min<uint64_t>((uint64_t)gsz * nb2, b->bp.n_seg - seg0);
      SegList S2; memset(&S2, 0, sizeof S2);
      S2.n = 1; S2.items[0] = tmp; S2.off[0] = d_goff + seg0;
      launch_segments(S2, nseg, gtot, (uint32_t)seg0);
    }
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(b->stream));
  return JFGPU_OK;
}

int bloom_flush(jfgpu_bloom* b) {
  if(b->pending.empty()) return JFGPU_OK;
  const int rc = bloom_flush_inner(b);
  if(rc && b->stream) hipStreamSynchronize(b->stream);
  b->pending.clear(); b->ws_used = 0;
  return rc;
}
