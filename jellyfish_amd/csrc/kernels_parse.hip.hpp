// jellyfish_amd/csrc/kernels_parse.hip.hpp -- FASTA / FASTQ bytes -> contract buffer, on the device (gfx950).
//
// the reference turns file bytes into the "sequence only, one 'N' between
// records" buffers that mer_iterator walks on the CPU, one istream call at a time
// (include/jellyfish/mer_overlap_sequence_parser.hpp:160-215, read_sequence :260-281).  At tens of
// G k-mers/s the device count would wait on that, so the same transformation runs here as a stream
// compaction over raw file bytes already in HBM:
//
//   A  parse_agg_kernel    per 4 KiB tile: the tile's effect on the line state + how many bytes it
//                          keeps for every state it could be entered in
//   B  parse_scan_kernel   one workgroup: scan of the tile summaries -> entry state and output
//                          offset of every tile, total output length
//   C  parse_emit_kernel   replay each tile with its entry state, stage the kept bytes in LDS,
//                          write them coalesced at the tile's offset
//   D  fastq_check_kernel  (FASTQ) per record: '@' / '+' line starts and |sequence| == |quality|
//
// FASTA rules (what the reference's reader does, restated per byte):
//   * a '>' that follows a '\n' (any '\r' in between skipped: skip_newlines :283-290) or starts the
//     chunk opens a header; the header ends at the next '\n'; the '>' itself becomes the 'N'
//     separator (:173-176), the rest of the header is dropped
//   * outside headers '\n' is dropped; a '\r' is dropped when its run of '\r' touches a '\n' or the
//     chunk boundary (trailing '\r' stripped :270-273, leading ones skipped :283-290), kept otherwise
//   * every other byte is kept as is ('>' in the middle of a line included): the count kernels map
//     non-ACGT to "break the k-mer" exactly as mer_iterator does
// FASTQ: strict 4-line records ('@' header, sequence, '+' line, quality).  Anything else -- wrapped
// sequence, blank lines, a quality string of the wrong length -- is reported back so the caller can
// hand the chunk to the host parser, which follows the reference's general reader including its
// "Invalid fastq sequence" error (:292-309).
#pragma once
#include "kernels.hip.hpp"

namespace jfgpu {

constexpr int kParseBlock = 256;
constexpr int kParseLane = 16;                           // bytes per thread, one 16-byte load
constexpr int kParseTile = kParseBlock * kParseLane;     // 4096 bytes per workgroup
constexpr int kScanBlock = 1024;

enum : uint32_t { PARSE_FASTA = 1, PARSE_FASTQ = 2 };
enum : uint32_t { EV_NONE = 0, EV_HEADER = 1, EV_SEQ = 2 };          // FASTA line state
enum : uint32_t { PF_BAD_AT = 1, PF_BAD_PLUS = 2, PF_BAD_LEN = 4, PF_TOO_MANY_LINES = 8, PF_TRUNCATED = 16 };

// FASTA: ev = last state-changing event in the tile, c[0] = bytes kept before the first event if
// the tile is entered outside a header, c[1] = bytes kept after it.
// FASTQ: ev = number of '\n' in the tile, c[p] = bytes kept if the tile's first line has phase p.
struct ParseAgg { uint32_t ev; uint32_t c[4]; };
struct ParseStart { uint64_t out_off; uint64_t state; };              // FASTA: entry state; FASTQ: index of the first line
struct ParseResult { uint64_t total; uint64_t records; uint64_t lines; uint64_t flags; };

__device__ inline uint32_t byte_of(const uint4& w, int i) {
  const uint32_t x = i < 4 ? w.x : i < 8 ? w.y : i < 12 ? w.z : w.w;
  return (x >> ((i & 3) * 8)) & 0xFFu;
}

// Is p the first character of a line the way the reference's reader sees it?
__device__ inline bool after_newline(const uint8_t* __restrict__ b, int64_t lo, int64_t p) {
  int64_t q = p - 1;
  for(int s = 0; s < 64 && q >= lo && b[q] == '\r'; ++s) --q;
  return q < lo || b[q] == '\n';
}

__device__ inline bool cr_dropped(const uint8_t* __restrict__ b, int64_t lo, int64_t hi, int64_t p) {
  int64_t q = p + 1;
  for(int s = 0; s < 64 && q < hi && b[q] == '\r'; ++s) ++q;
  if(q >= hi || b[q] == '\n') return true;
  return after_newline(b, lo, p);
}

// exclusive "last non-zero" scan over the workgroup; *total = last non-zero of all
__device__ inline uint32_t block_scan_last(uint32_t v, uint32_t* s_w, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint32_t incl = v;
  for(int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(incl, o, 64);
    if((int)lane >= o && incl == 0) incl = up;
  }
  uint32_t excl = __shfl_up(incl, 1, 64);
  if(lane == 0) excl = 0;
  if(lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t prefix = 0, all = 0;
  for(uint32_t w = 0; w < nw; ++w) {
    const uint32_t x = s_w[w];
    if(x) { all = x; if(w < wave) prefix = x; }
  }
  __syncthreads();
  if(total) *total = all;
  return excl ? excl : prefix;
}

// exclusive sum over the workgroup; *total = sum of all
template <typename T>
__device__ inline T block_scan_sum(T v, T* s_w, T* total) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  T incl = v;
  for(int o = 1; o < 64; o <<= 1) {
    const T up = __shfl_up(incl, o, 64);
    if((int)lane >= o) incl += up;
  }
  if(lane == 63) s_w[wave] = incl;
  __syncthreads();
  T prefix = 0, all = 0;
  for(uint32_t w = 0; w < nw; ++w) { const T x = s_w[w]; all += x; if(w < wave) prefix += x; }
  __syncthreads();
  if(total) *total = all;
  return prefix + incl - v;
}

// What one thread knows about its 16 bytes before any cross-thread state: which are valid, which
// are '\n', which open a FASTA header.
struct LaneBytes {
  uint4 w; int64_t p0; uint32_t valid, nl, hdr;
};

template <uint32_t FMT>
__device__ inline LaneBytes lane_load(const uint8_t* __restrict__ b, int64_t lo, int64_t hi, int64_t tile) {
  LaneBytes L;
  L.p0 = tile * kParseTile + (int64_t)threadIdx.x * kParseLane;
  L.w = make_uint4(0, 0, 0, 0); L.valid = L.nl = L.hdr = 0;
  if(L.p0 < hi && L.p0 + kParseLane > lo) {
    L.w = *reinterpret_cast<const uint4*>(b + L.p0);
#pragma unroll
    for(int i = 0; i < kParseLane; ++i) {
      const int64_t p = L.p0 + i;
      if(p < lo || p >= hi) continue;
      L.valid |= 1u << i;
      const uint32_t c = byte_of(L.w, i);
      if(c == '\n') L.nl |= 1u << i;
      if(FMT == PARSE_FASTA && c == '>' && after_newline(b, lo, p)) L.hdr |= 1u << i;
    }
  }
  return L;
}

__device__ inline uint32_t lane_last_event(const LaneBytes& L) {
  if((L.nl | L.hdr) == 0) return EV_NONE;
  return L.hdr > L.nl ? EV_HEADER : EV_SEQ;     // the higher bit is the later byte
}

// FASTA: output character for byte i given the state before it (0 = dropped); updates the state.
__device__ inline uint32_t fasta_step(const LaneBytes& L, int i, uint32_t& st, const uint8_t* __restrict__ b, int64_t lo, int64_t hi) {
  const uint32_t c = byte_of(L.w, i);
  if(L.hdr >> i & 1) { st = EV_HEADER; return 'N'; }
  if(L.nl >> i & 1) { st = EV_SEQ; return 0; }
  if(st == EV_HEADER) return 0;
  if(c == '\r') return cr_dropped(b, lo, hi, L.p0 + i) ? 0u : c;
  return c;
}

// FASTQ: is byte i the first of a line?
__device__ inline bool fastq_line_start(const LaneBytes& L, int i, const uint8_t* __restrict__ b, int64_t lo) {
  const int64_t p = L.p0 + i;
  if(p == lo) return true;
  return i > 0 ? (L.nl >> (i - 1) & 1) : b[p - 1] == '\n';
}

// ---- A: tile summaries ------------------------------------------------------------------
template <uint32_t FMT>
__global__ __launch_bounds__(kParseBlock) void parse_agg_kernel(const uint8_t* __restrict__ b, int64_t lo, int64_t hi,
                                                                int64_t tile0, ParseAgg* __restrict__ agg) {
  __shared__ uint32_t s_w[kParseBlock / 64];
  __shared__ uint32_t s_c[4];
  const int64_t tile = tile0 + blockIdx.x;
  if(threadIdx.x < 4) s_c[threadIdx.x] = 0;
  const LaneBytes L = lane_load<FMT>(b, lo, hi, tile);
  uint32_t acc = 0;                                       // four 8-bit counters
  uint32_t tile_ev = 0;
  if(FMT == PARSE_FASTA) {
    uint32_t st = block_scan_last(lane_last_event(L), s_w, &tile_ev);    // also orders the s_c reset
    bool seen = st != EV_NONE;
    for(int i = 0; i < kParseLane; ++i) {
      if(!(L.valid >> i & 1)) continue;
      const bool ev = (L.hdr | L.nl) >> i & 1;
      uint32_t s2 = st == EV_NONE ? (uint32_t)EV_SEQ : st;
      const uint32_t o = fasta_step(L, i, s2, b, lo, hi);
      if(o) acc += (seen || ev) ? 0x100u : 1u;           // the header's own 'N' counts as "after"
      if(ev) { seen = true; st = s2; }
    }
  } else {
    uint32_t r = block_scan_sum<uint32_t>((uint32_t)__popc(L.nl), s_w, &tile_ev);
    for(int i = 0; i < kParseLane; ++i) {
      if(!(L.valid >> i & 1)) continue;
      const uint32_t c = byte_of(L.w, i);
      if(fastq_line_start(L, i, b, lo)) acc += 1u << (((0u - r) & 3) * 8);          // 'N' when this line has phase 0
      if(c == '\n') ++r;
      else if(c != '\r' || !cr_dropped(b, lo, hi, L.p0 + i)) acc += 1u << (((1u - r) & 3) * 8);   // kept when phase 1
    }
  }
#pragma unroll
  for(int j = 0; j < 4; ++j) {
    uint32_t v = (acc >> (8 * j)) & 0xFFu;
    for(int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
    if((threadIdx.x & 63) == 0 && v) atomicAdd(&s_c[j], v);
  }
  __syncthreads();
  if(threadIdx.x == 0) {
    ParseAgg a; a.ev = tile_ev; a.c[0] = s_c[0]; a.c[1] = s_c[1]; a.c[2] = s_c[2]; a.c[3] = s_c[3];
    agg[blockIdx.x] = a;
  }
}

// ---- B: scan of the summaries (single workgroup) ------------------------------------------
template <uint32_t FMT>
__global__ __launch_bounds__(kScanBlock) void parse_scan_kernel(const ParseAgg* __restrict__ agg, int64_t nt,
                                                                ParseStart* __restrict__ start, ParseResult* __restrict__ res) {
  __shared__ uint64_t s_w64[kScanBlock / 64];
  __shared__ uint32_t s_w32[kScanBlock / 64];
  const int64_t span = (nt + kScanBlock - 1) / kScanBlock;
  int64_t a = (int64_t)threadIdx.x * span; if(a > nt) a = nt;
  int64_t e = a + span; if(e > nt) e = nt;
  uint64_t st0;
  if(FMT == PARSE_FASTA) {
    uint32_t last = 0;
    for(int64_t i = a; i < e; ++i) if(agg[i].ev) last = agg[i].ev;
    const uint32_t in = block_scan_last(last, s_w32, nullptr);
    st0 = in ? in : (uint32_t)EV_SEQ;                     // a chunk starts outside a header
  } else {
    uint64_t nl = 0;
    for(int64_t i = a; i < e; ++i) nl += agg[i].ev;
    uint64_t tot = 0;
    st0 = block_scan_sum<uint64_t>(nl, s_w64, &tot);
    if(threadIdx.x == 0) res->lines = tot;
  }
  uint64_t kept = 0, st = st0;
  for(int64_t i = a; i < e; ++i) {
    const ParseAgg g = agg[i];
    if(FMT == PARSE_FASTA) { kept += g.c[1] + (st != EV_HEADER ? g.c[0] : 0u); if(g.ev) st = g.ev; }
    else { kept += g.c[st & 3]; st += g.ev; }
  }
  uint64_t total = 0;
  uint64_t off = block_scan_sum<uint64_t>(kept, s_w64, &total);
  st = st0;
  for(int64_t i = a; i < e; ++i) {
    const ParseAgg g = agg[i];
    start[i].out_off = off; start[i].state = st;
    if(FMT == PARSE_FASTA) { off += g.c[1] + (st != EV_HEADER ? g.c[0] : 0u); if(g.ev) st = g.ev; }
    else { off += g.c[st & 3]; st += g.ev; }
  }
  if(threadIdx.x == 0) res->total = total;
}

// ---- C: emit ------------------------------------------------------------------------------
// out_base: where the chunk's first kept byte goes.  FASTQ also records the position of every '\n'
// (relative to lo) for the record check.
template <uint32_t FMT>
__global__ __launch_bounds__(kParseBlock) void parse_emit_kernel(const uint8_t* __restrict__ b, int64_t lo, int64_t hi, int64_t tile0,
                                                                 const ParseStart* __restrict__ start, uint8_t* __restrict__ out_base,
                                                                 uint32_t* __restrict__ nlpos, uint64_t nlpos_cap,
                                                                 ParseResult* __restrict__ res) {
  __shared__ uint32_t s_w[kParseBlock / 64];
  __shared__ uint8_t s_out[kParseTile];
  const int64_t tile = tile0 + blockIdx.x;
  const ParseStart S = start[blockIdx.x];
  const LaneBytes L = lane_load<FMT>(b, lo, hi, tile);
  uint32_t recs = 0;
  uint32_t st_in = 0; uint64_t line_in = 0;
  if(FMT == PARSE_FASTA) {
    st_in = block_scan_last(lane_last_event(L), s_w, nullptr);
    if(st_in == EV_NONE) st_in = (uint32_t)S.state;
    recs = __popc(L.hdr);
  } else {
    line_in = S.state + block_scan_sum<uint32_t>((uint32_t)__popc(L.nl), s_w, nullptr);
  }
  // the lane's bytes are walked twice (count, then place) so nothing is indexed dynamically in registers
  auto walk = [&](auto&& put) {
    if(FMT == PARSE_FASTA) {
      uint32_t st = st_in;
#pragma unroll
      for(int i = 0; i < kParseLane; ++i) {
        if(!(L.valid >> i & 1)) continue;
        const uint32_t c = fasta_step(L, i, st, b, lo, hi);
        if(c) put(c);
      }
    } else {
      uint64_t line = line_in;
#pragma unroll
      for(int i = 0; i < kParseLane; ++i) {
        if(!(L.valid >> i & 1)) continue;
        const uint32_t c = byte_of(L.w, i);
        if((line & 3) == 0 && fastq_line_start(L, i, b, lo)) put('N');
        if(c == '\n') ++line;
        else if((line & 3) == 1 && (c != '\r' || !cr_dropped(b, lo, hi, L.p0 + i))) put(c);
      }
    }
  };
  uint32_t n = 0;
  walk([&](uint32_t) { ++n; });
  uint32_t tile_n = 0;
  uint32_t at = block_scan_sum<uint32_t>(n, s_w, &tile_n);
  walk([&](uint32_t c) { s_out[at++] = (uint8_t)c; });
  if(FMT == PARSE_FASTQ) {
    uint64_t line = line_in;
#pragma unroll
    for(int i = 0; i < kParseLane; ++i) {
      if(!(L.nl >> i & 1)) continue;
      if((line & 3) == 0) ++recs;                        // one record per line of phase 0 that ends
      if(line < nlpos_cap) nlpos[line] = (uint32_t)(L.p0 + i - lo);
      ++line;
    }
  }
  __syncthreads();
  uint8_t* dst = out_base + S.out_off;
  for(uint32_t i = threadIdx.x; i < tile_n; i += kParseBlock) dst[i] = s_out[i];
  for(int k = 32; k; k >>= 1) recs += __shfl_xor(recs, k, 64);
  if((threadIdx.x & 63) == 0 && recs) atomicAdd((unsigned long long*)&res->records, (unsigned long long)recs);
}

// ---- D: FASTQ record check ------------------------------------------------------------------
// n_lines complete lines (the last one may lack its '\n': then its end is hi).
__global__ void fastq_check_kernel(const uint8_t* __restrict__ b, int64_t lo, int64_t hi, const uint32_t* __restrict__ nlpos,
                                   uint64_t n_newlines, uint64_t n_records, ParseResult* __restrict__ res) {
  uint32_t bad = 0;
  for(uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n_records; r += (uint64_t)gridDim.x * blockDim.x) {
    int64_t s[4], e[4];
#pragma unroll
    for(int j = 0; j < 4; ++j) {
      const uint64_t l = 4 * r + j;
      s[j] = l == 0 ? lo : lo + (int64_t)nlpos[l - 1] + 1;
      e[j] = l < n_newlines ? lo + (int64_t)nlpos[l] : hi;
    }
    if(s[0] >= hi || b[s[0]] != '@') bad |= PF_BAD_AT;
    if(s[2] >= hi || b[s[2]] != '+') bad |= PF_BAD_PLUS;
    int64_t l1 = e[1] - s[1], l3 = e[3] - s[3];
    while(l1 > 0 && b[s[1] + l1 - 1] == '\r') --l1;
    while(l3 > 0 && b[s[3] + l3 - 1] == '\r') --l3;
    if(l1 != l3) bad |= PF_BAD_LEN;
  }
  if(bad) atomicOr((unsigned long long*)&res->flags, (unsigned long long)bad);
}

// ---- E: quality masking (-Q / --min-quality) ---------------------------------------------------
// mer_qual_iterator (/root/reference/include/jellyfish/mer_qual_iterator.hpp:75-84): a base whose quality character is
// below min_qual counts as an invalid character.  On a chunk that passed the record check (strict 4-line records, equal
// sequence and quality lengths) that is: sequence byte i of a record becomes 'N' when quality byte i < min_qual.  The
// raw chunk is edited in place, then the emit pass is run again (the layout of kept bytes does not change).
__global__ void fastq_qual_mask_kernel(uint8_t* __restrict__ b, int64_t lo, int64_t hi, const uint32_t* __restrict__ nlpos,
                                       uint64_t n_newlines, uint64_t n_records, uint32_t min_qual) {
  // one wave per record, lanes stride over its bases
  const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  for(uint64_t r = wave; r < n_records; r += n_waves) {
    const uint64_t l1 = 4 * r + 1, l3 = 4 * r + 3;
    const int64_t s1 = lo + (int64_t)nlpos[l1 - 1] + 1, e1 = lo + (int64_t)nlpos[l1];
    const int64_t s3 = lo + (int64_t)nlpos[l3 - 1] + 1;
    const int64_t e3 = l3 < n_newlines ? lo + (int64_t)nlpos[l3] : hi;
    int64_t len = e1 - s1;
    if(e3 - s3 < len) len = e3 - s3;
    for(int64_t i = lane; i < len; i += 64) {
      const uint32_t q = b[s3 + i];
      if(q != '\r' && q < min_qual) b[s1 + i] = 'N';
    }
  }
}

}  // namespace jfgpu
