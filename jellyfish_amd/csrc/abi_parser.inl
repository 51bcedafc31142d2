// jellyfish_amd/csrc/abi_parser.inl -- C ABI of the device-side sequence parser (jfgpu_parser_*), included by jfgpu.hip.
// ---- device-side FASTA / FASTQ parse (kernels_parse.hip.hpp) ----------------------------------------
struct jfgpu_parser {
  int device = 0;
  uint32_t k = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  uint8_t* d_raw = nullptr; size_t raw_cap = 0;            // host-fed chunks land here
  // pipelined feed (jfgpu_parser_upload / _parse_uploaded): chunk i+1 travels on the copy stream while chunk i is parsed
  uint8_t* d_up[2] = {nullptr, nullptr}; size_t up_cap[2] = {0, 0}; size_t up_len[2] = {0, 0};
  hipStream_t copy_stream = nullptr; hipEvent_t up_done[2] = {nullptr, nullptr};
  uint8_t* d_out[2] = {nullptr, nullptr}; size_t out_cap[2] = {0, 0};
  int cur = 0;
  ParseAgg* d_agg = nullptr; ParseStart* d_start = nullptr; size_t tiles_cap = 0;
  uint32_t* d_nlpos = nullptr; size_t nlpos_cap = 0;
  ParseResult* d_res = nullptr;
  uint8_t* d_carry = nullptr; uint32_t carry_len = 0;      // last k-1 characters of the previous chunk's output
  char* h_pin[2] = {nullptr, nullptr}; size_t pin_cap[2] = {0, 0};   // pinned host staging for callers that read files
  double last_ms = 0;
  uint32_t min_qual = 0;                                   // > 0: FASTQ bases with a lower quality character become 'N'
};

namespace {
constexpr size_t kOutPad = 256;                             // room for the seam in front of the compacted bytes
int use_p(const jfgpu_parser* p) {
  if(!p) return fail(JFGPU_E_INVALID, "null parser");
  HIP_TRY(hipSetDevice(p->device));
  return JFGPU_OK;
}
template <typename T>
int grow_buf(T*& ptr, size_t& cap, size_t need) {
  if(need <= cap) return JFGPU_OK;
  if(ptr) { HIP_TRY(hipFree(ptr)); ptr = nullptr; cap = 0; }
  const size_t want = need + need / 8;
  HIP_TRY(hipMalloc((void**)&ptr, want * sizeof(T)));
  cap = want;
  return JFGPU_OK;
}
}  // namespace

extern "C" {

int jfgpu_parser_create(int device, uint32_t k, jfgpu_parser** out) {
  if(!out) return fail(JFGPU_E_INVALID, "null argument");
  *out = nullptr;
  if(k < 1 || k > 127) return fail(JFGPU_E_INVALID, "mer length must be in [1, 127]");
  int ndev = 0;
  if(hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(JFGPU_E_NO_DEVICE, "no HIP device: the engine has no CPU fallback");
  int dev = device;
  if(dev < 0) HIP_TRY(hipGetDevice(&dev));
  if(dev >= ndev) return fail(JFGPU_E_NO_DEVICE, "device ordinal out of range");
  HIP_TRY(hipSetDevice(dev));
  std::unique_ptr<jfgpu_parser> p(new jfgpu_parser);
  p->device = dev; p->k = k;
  HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&p->ev_a)); HIP_TRY(hipEventCreate(&p->ev_b));
  HIP_TRY(hipMalloc((void**)&p->d_res, sizeof(ParseResult)));
  HIP_TRY(hipMalloc((void**)&p->d_carry, 256));
  *out = p.release();
  return JFGPU_OK;
}

void jfgpu_parser_destroy(jfgpu_parser* p) {
  if(!p) return;
  hipSetDevice(p->device);
  if(p->stream) hipStreamSynchronize(p->stream);
  hipFree(p->d_raw); hipFree(p->d_out[0]); hipFree(p->d_out[1]); hipFree(p->d_agg); hipFree(p->d_start);
  hipFree(p->d_nlpos); hipFree(p->d_res); hipFree(p->d_carry);
  if(p->copy_stream) { hipStreamSynchronize(p->copy_stream); hipStreamDestroy(p->copy_stream); }
  for(int i = 0; i < 2; ++i) { if(p->d_up[i]) hipFree(p->d_up[i]); if(p->up_done[i]) hipEventDestroy(p->up_done[i]); }
  for(int i = 0; i < 2; ++i) if(p->h_pin[i]) hipHostFree(p->h_pin[i]);
  if(p->ev_a) hipEventDestroy(p->ev_a);
  if(p->ev_b) hipEventDestroy(p->ev_b);
  if(p->stream) hipStreamDestroy(p->stream);
  delete p;
}

int jfgpu_parser_parse_dev(jfgpu_parser* p, const char* d_bytes, size_t n, unsigned flags, const char** d_out, size_t* n_out,
                           uint64_t* n_records) {
  int rc = use_p(p); if(rc) return rc;
  if(!d_out || !n_out) return fail(JFGPU_E_INVALID, "null argument");
  *d_out = nullptr; *n_out = 0;
  if(n_records) *n_records = 0;
  const unsigned fmt = flags & 3u;
  if(fmt != JFGPU_PARSE_FASTA && fmt != JFGPU_PARSE_FASTQ) return fail(JFGPU_E_INVALID, "format must be JFGPU_PARSE_FASTA or JFGPU_PARSE_FASTQ");
  if(n > ((size_t)1 << 31)) return fail(JFGPU_E_INVALID, "chunk larger than 2^31 bytes");
  if(!(flags & JFGPU_PARSE_CONTINUE) || fmt == JFGPU_PARSE_FASTQ) p->carry_len = 0;
  if(n == 0) return JFGPU_OK;
  if(!d_bytes) return fail(JFGPU_E_INVALID, "null buffer");
  const uint8_t* base; int64_t lo, hi;
  align_buffer(d_bytes, n, base, lo, hi);
  const int64_t nt = (hi + kParseTile - 1) / kParseTile;
  const int w = p->cur; p->cur ^= 1;
  rc = grow_buf(p->d_out[w], p->out_cap[w], kOutPad + n + 64); if(rc) return rc;
  if((size_t)nt > p->tiles_cap) {
    size_t c1 = p->tiles_cap, c2 = p->tiles_cap;
    rc = grow_buf(p->d_agg, c1, (size_t)nt); if(rc) return rc;
    rc = grow_buf(p->d_start, c2, (size_t)nt); if(rc) return rc;
    p->tiles_cap = std::min(c1, c2);
  }
  const uint64_t max_lines = n / 8 + 16;                     // FASTQ chunks with shorter lines go to the host parser
  if(fmt == JFGPU_PARSE_FASTQ) { rc = grow_buf(p->d_nlpos, p->nlpos_cap, (size_t)max_lines); if(rc) return rc; }
  uint8_t* out0 = p->d_out[w] + kOutPad;
  HIP_TRY(hipMemsetAsync(p->d_res, 0, sizeof(ParseResult), p->stream));
  HIP_TRY(hipEventRecord(p->ev_a, p->stream));
  if(fmt == JFGPU_PARSE_FASTA) {
    hipLaunchKernelGGL(parse_agg_kernel<PARSE_FASTA>, dim3((unsigned)nt), dim3(kParseBlock), 0, p->stream, base, lo, hi, (int64_t)0, p->d_agg);
    hipLaunchKernelGGL(parse_scan_kernel<PARSE_FASTA>, dim3(1), dim3(kScanBlock), 0, p->stream, p->d_agg, nt, p->d_start, p->d_res);
    hipLaunchKernelGGL(parse_emit_kernel<PARSE_FASTA>, dim3((unsigned)nt), dim3(kParseBlock), 0, p->stream, base, lo, hi, (int64_t)0,
                       p->d_start, out0, (uint32_t*)nullptr, (uint64_t)0, p->d_res);
  } else {
    hipLaunchKernelGGL(parse_agg_kernel<PARSE_FASTQ>, dim3((unsigned)nt), dim3(kParseBlock), 0, p->stream, base, lo, hi, (int64_t)0, p->d_agg);
    hipLaunchKernelGGL(parse_scan_kernel<PARSE_FASTQ>, dim3(1), dim3(kScanBlock), 0, p->stream, p->d_agg, nt, p->d_start, p->d_res);
    hipLaunchKernelGGL(parse_emit_kernel<PARSE_FASTQ>, dim3((unsigned)nt), dim3(kParseBlock), 0, p->stream, base, lo, hi, (int64_t)0,
                       p->d_start, out0, p->d_nlpos, (uint64_t)max_lines, p->d_res);
  }
  HIP_TRY(hipGetLastError());
  ParseResult res; uint8_t last = 0;
  HIP_TRY(hipMemcpyAsync(&res, p->d_res, sizeof(res), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipMemcpyAsync(&last, d_bytes + n - 1, 1, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  uint64_t records = res.records;
  if(fmt == JFGPU_PARSE_FASTQ) {
    // a last line without '\n' still counts; so does an EMPTY last quality line (a record with no bases at the
    // end of a file whose final newline is missing) -- the record check below compares the two lengths anyway
    const uint64_t lines = res.lines + ((last != '\n' || res.lines % 4 == 3) ? 1 : 0);
    uint64_t flags_bad = 0;
    if(res.lines > max_lines) flags_bad |= PF_TOO_MANY_LINES;
    else if(lines % 4) flags_bad |= PF_TRUNCATED;
    else {
      records = lines / 4;
      const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((records + 255) / 256, 4096));
      hipLaunchKernelGGL(fastq_check_kernel, dim3(grid), dim3(256), 0, p->stream, base, lo, hi, p->d_nlpos, res.lines, records, p->d_res);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(&res, p->d_res, sizeof(res), hipMemcpyDeviceToHost, p->stream));
      HIP_TRY(hipStreamSynchronize(p->stream));
      flags_bad = res.flags;
    }
    if(flags_bad) {
      p->cur ^= 1;                                            // nothing was produced: keep the previous buffers' turn
      std::string why;
      if(flags_bad & PF_TOO_MANY_LINES) why += " lines shorter than 8 bytes on average;";
      if(flags_bad & PF_TRUNCATED) why += " number of lines not a multiple of 4;";
      if(flags_bad & PF_BAD_AT) why += " a record does not start with '@';";
      if(flags_bad & PF_BAD_PLUS) why += " third line of a record does not start with '+';";
      if(flags_bad & PF_BAD_LEN) why += " sequence and quality lengths differ;";
      return fail(JFGPU_E_FORMAT, "not a strict 4-line FASTQ chunk:" + why);
    }
  }
  if(fmt == JFGPU_PARSE_FASTQ && p->min_qual) {
    // quality masking (mer_qual_iterator.hpp:75-84): edit the raw chunk in place, emit again
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((records + 3) / 4, 8192));
    hipLaunchKernelGGL(fastq_qual_mask_kernel, dim3(grid), dim3(256), 0, p->stream, const_cast<uint8_t*>(base), lo, hi, p->d_nlpos, res.lines, records, p->min_qual);
    HIP_TRY(hipMemsetAsync(p->d_res, 0, sizeof(ParseResult), p->stream));
    hipLaunchKernelGGL(parse_emit_kernel<PARSE_FASTQ>, dim3((unsigned)nt), dim3(kParseBlock), 0, p->stream, base, lo, hi, (int64_t)0,
                       p->d_start, out0, p->d_nlpos, (uint64_t)max_lines, p->d_res);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(p->ev_b, p->stream));
  // seam: the previous chunk's last k-1 characters in front, then remember this chunk's
  uint8_t* start = out0 - p->carry_len;
  if(p->carry_len) HIP_TRY(hipMemcpyAsync(start, p->d_carry, p->carry_len, hipMemcpyDeviceToDevice, p->stream));
  const size_t total = (size_t)res.total + p->carry_len;
  if(fmt == JFGPU_PARSE_FASTA) {
    const uint32_t keep = (uint32_t)std::min<size_t>(total, p->k - 1);
    if(keep) HIP_TRY(hipMemcpyAsync(p->d_carry, start + total - keep, keep, hipMemcpyDeviceToDevice, p->stream));
    p->carry_len = keep;
  }
  HIP_TRY(hipStreamSynchronize(p->stream));
  float ms = 0;
  if(hipEventElapsedTime(&ms, p->ev_a, p->ev_b) == hipSuccess) p->last_ms = ms;
  *d_out = (const char*)start; *n_out = total;
  if(n_records) *n_records = records;
  return JFGPU_OK;
}

int jfgpu_parser_parse(jfgpu_parser* p, const char* bytes, size_t n, unsigned flags, const char** d_out, size_t* n_out,
                       uint64_t* n_records) {
  int rc = use_p(p); if(rc) return rc;
  if(n && !bytes) return fail(JFGPU_E_INVALID, "null buffer");
  if(n > ((size_t)1 << 31)) return fail(JFGPU_E_INVALID, "chunk larger than 2^31 bytes");
  if(n) {
    rc = grow_buf(p->d_raw, p->raw_cap, n + 64); if(rc) return rc;
    HIP_TRY(hipMemcpyAsync(p->d_raw, bytes, n, hipMemcpyHostToDevice, p->stream));
  }
  return jfgpu_parser_parse_dev(p, (const char*)p->d_raw, n, flags, d_out, n_out, n_records);
}

// Pipelined form of jfgpu_parser_parse for callers that read files: upload(which) enqueues the host-to-device copy of
// a pinned buffer on the parser's copy stream and returns at once; parse_uploaded(which) waits for that copy only and
// parses.  With two host buffers and two device buffers the copy of chunk i+1 overlaps the parse (and the counting)
// of chunk i while the caller reads chunk i+2 from the file.
int jfgpu_parser_upload(jfgpu_parser* p, int which, const char* bytes, size_t n) {
  int rc = use_p(p); if(rc) return rc;
  if(which < 0 || which > 1 || (n && !bytes)) return fail(JFGPU_E_INVALID, "bad argument");
  if(n > ((size_t)1 << 31)) return fail(JFGPU_E_INVALID, "chunk larger than 2^31 bytes");
  if(!p->copy_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
    for(int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&p->up_done[i], hipEventDisableTiming));
  }
  if(n + 64 > p->up_cap[which]) {
    HIP_TRY(hipStreamSynchronize(p->copy_stream)); HIP_TRY(hipStreamSynchronize(p->stream));
    rc = grow_buf(p->d_up[which], p->up_cap[which], n + 64); if(rc) return rc;
  }
  if(n) HIP_TRY(hipMemcpyAsync(p->d_up[which], bytes, n, hipMemcpyHostToDevice, p->copy_stream));
  HIP_TRY(hipEventRecord(p->up_done[which], p->copy_stream));
  p->up_len[which] = n;
  return JFGPU_OK;
}

int jfgpu_parser_upload_wait(jfgpu_parser* p, int which) {      // the host buffer given to upload(which) may be refilled
  int rc = use_p(p); if(rc) return rc;
  if(which < 0 || which > 1 || !p->copy_stream) return fail(JFGPU_E_INVALID, "nothing was uploaded");
  HIP_TRY(hipEventSynchronize(p->up_done[which]));
  return JFGPU_OK;
}

int jfgpu_parser_parse_uploaded(jfgpu_parser* p, int which, unsigned flags, const char** d_out, size_t* n_out, uint64_t* n_records) {
  int rc = use_p(p); if(rc) return rc;
  if(which < 0 || which > 1 || !p->copy_stream) return fail(JFGPU_E_INVALID, "nothing was uploaded");
  HIP_TRY(hipStreamWaitEvent(p->stream, p->up_done[which], 0));
  return jfgpu_parser_parse_dev(p, (const char*)p->d_up[which], p->up_len[which], flags, d_out, n_out, n_records);
}

int jfgpu_parser_host_buffer(jfgpu_parser* p, int which, size_t bytes, char** out) {
  int rc = use_p(p); if(rc) return rc;
  if(!out || which < 0 || which > 1) return fail(JFGPU_E_INVALID, "bad argument");
  *out = nullptr;
  if(bytes > p->pin_cap[which]) {
    if(p->h_pin[which]) { HIP_TRY(hipHostFree(p->h_pin[which])); p->h_pin[which] = nullptr; p->pin_cap[which] = 0; }
    HIP_TRY(hipHostMalloc((void**)&p->h_pin[which], bytes, hipHostMallocDefault));
    p->pin_cap[which] = bytes;
  }
  *out = p->h_pin[which];
  return JFGPU_OK;
}

int jfgpu_parser_set_min_quality(jfgpu_parser* p, int min_qual_char) {
  int rc = use_p(p); if(rc) return rc;
  if(min_qual_char < 0 || min_qual_char > 126) return fail(JFGPU_E_INVALID, "quality character outside [0, '~']");
  p->min_qual = (uint32_t)min_qual_char;
  return JFGPU_OK;
}

int jfgpu_parser_last_ms(jfgpu_parser* p, double* ms) {
  int rc = use_p(p); if(rc) return rc;
  if(ms) *ms = p->last_ms;
  return JFGPU_OK;
}

}  // extern "C"
