// Round-3 LDS micro-benchmarks (gfx950).  Two questions, no global traffic in either:
//   T*: what does it cost to place the ~8270 items of a pair of tiles (16384 32-bit slots in LDS)?
//       T0  CAS claim + triangular probing (round 2's tile_insert_one)
//       T1  CAS claim + linear probing from the 8-aligned home bucket
//       T2  rank placement: one returning ds_add per item on the home bucket's counter gives the slot; a lane-per-bucket
//           pass merges equal tags; what does not fit the bucket goes through T1's loop
//       T3  T2 without the merge pass (price of the pass)
//   H*: what does the GF(2) hash of a 42-bit key cost?
//       H0  6 x ds_read_b64 byte tables (round 2)      H1  6 x ds_read_b32 byte tables + 2 parity rows on the VALU
//       H2  4 x ds_read_b32 11-bit tables + 2 parity rows   H3  3 x ds_read_b32 14-bit tables (48 KiB.. 64Ki entries do not fit: 2 x 16 + 10 bits)
//       H4  42 select-XORs on the VALU
// hipcc --offload-arch=gfx950 -O3 -o r03_lds_probe r03_lds_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int kThreads = 1024, kTiles = 64, kSlots = 16384, kBuckets = kSlots / 8, kNP = 9;
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// slot word: count (10 bits) | occ (bit 21) | tag (21 bits: idx0 13 bits << 8 | rem 8 bits); item: half (bit 21) | tag
constexpr uint32_t kTagBits = 21, kOcc = 1u << kTagBits, kLow = (kOcc << 1) - 1, kInc = kOcc << 1, kRem = 8;

template <int MODE>
__global__ __launch_bounds__(kThreads, 8) void tprobe(uint32_t* out, uint32_t pool) {
  extern __shared__ __align__(16) uint32_t s_tile[];
  uint32_t* s_cnt = s_tile + kSlots;           // [kBuckets]
  uint32_t* s_q = s_cnt + kBuckets;            // [1 + 2047] overflow queue
  uint32_t acc = 0;
  for(int t = 0; t < kTiles; ++t) {
    for(uint32_t i = threadIdx.x * 4; i < kSlots; i += kThreads * 4) *reinterpret_cast<uint4*>(s_tile + i) = make_uint4(0, 0, 0, 0);
    for(uint32_t i = threadIdx.x; i < kBuckets; i += kThreads) s_cnt[i] = 0;   // (T5/T6: 4096 16-bit counters in the same 2048 words)
    if(threadIdx.x == 0) s_q[0] = 0;
    lds_barrier();
    // 8270 items per pair: every lane 8, lanes < 78 a ninth
    const int n = 8 + (threadIdx.x < 78 ? 1 : 0);
    uint32_t it[kNP];
    uint32_t x = mix(threadIdx.x * 2654435761u + blockIdx.x * 40503u + t * 977u + 1u);
#pragma unroll
    for(int r = 0; r < kNP; ++r) { x = x * 1664525u + 1013904223u; uint32_t v = mix(x); if(pool) v = mix((v % pool) * 2654435761u + t * 31u + blockIdx.x); it[r] = v & ((1u << 22) - 1); }
    if(MODE == 0 || MODE == 1) {
#pragma unroll
      for(int r = 0; r < kNP; ++r) if(r < n) {
        const uint32_t half = it[r] >> kTagBits, tag = it[r] & (kOcc - 1), idx0 = tag >> kRem;
        const uint32_t low = kOcc | tag, neww = kInc | low;
        uint32_t* tl = s_tile + half * 8192;
        for(uint32_t p = 0; p < 1024; ++p) {
          const uint32_t slot = MODE == 0 ? ((idx0 + p * (p + 1) / 2) & 8191u) : (((idx0 & ~7u) + p) & 8191u);
          const uint32_t old = atomicCAS(&tl[slot], 0u, neww);
          if(old == 0u) break;
          if((old & kLow) == low) { atomicAdd(&tl[slot], kInc); break; }
        }
      }
    } else if(MODE == 4) {
#pragma unroll
      for(int r = 0; r < kNP; ++r) if(r < n) acc ^= it[r];
    } else if(MODE >= 5 && MODE <= 8) {
      // bucket = 4 slots (one 16-byte vector); counters: two 16-bit ranks per word
      uint32_t old[kNP];
#pragma unroll
      for(int r = 0; r < kNP; ++r) if(r < n) {
        const uint32_t b = (it[r] >> (kRem + 2)) & 4095u;
        old[r] = atomicAdd(&s_cnt[b >> 1], 1u << ((b & 1) * 16));
      }
#pragma unroll
      for(int r = 0; r < kNP; ++r) if(r < n) {
        const uint32_t b = (it[r] >> (kRem + 2)) & 4095u;
        const uint32_t rank = (old[r] >> ((b & 1) * 16)) & 0xFFFFu;
        if(MODE == 8) { acc ^= rank; continue; }
        if(rank < 4) s_tile[b * 4 + rank] = kInc | kOcc | (it[r] & (kOcc - 1));
        else { const uint32_t q = atomicAdd(&s_q[0], 1u); if(q < 2047) s_q[1 + q] = it[r]; }
      }
      lds_barrier();
      if(MODE == 5) {
        for(uint32_t b2 = threadIdx.x; b2 < 2048; b2 += kThreads) {       // two buckets per lane-iteration
          const uint32_t cc = s_cnt[b2];
          if(((cc & 0xFFFFu) < 2) && ((cc >> 16) < 2)) continue;
          uint4 v[2] = {*reinterpret_cast<const uint4*>(s_tile + b2 * 8), *reinterpret_cast<const uint4*>(s_tile + b2 * 8 + 4)};
#pragma unroll
          for(int h = 0; h < 2; ++h) {
            uint32_t w[4] = {v[h].x, v[h].y, v[h].z, v[h].w};
            // identity inside a bucket = the low 16 bits of the word (rem + low idx0 bits); empty slots get distinct sentinels
            uint32_t s0 = w[0] << 10, s1 = w[1] ? w[1] << 10 : 1u, s2 = w[2] ? w[2] << 10 : 2u, s3 = w[3] ? w[3] << 10 : 3u;
            uint32_t m = min(min(s0 ^ s1, s0 ^ s2), min(s0 ^ s3, s1 ^ s2));
            m = min(m, min(s1 ^ s3, s2 ^ s3));
            if(m == 0 && w[0]) {
              for(int j = 1; j < 4; ++j) for(int i = 0; i < j; ++i)
                if(w[j] != 0 && w[i] != 0 && ((w[i] ^ w[j]) & kLow) == 0) { w[i] += w[j] & ~kLow; w[j] = 0; }
              uint32_t o[4] = {0, 0, 0, 0}; int mm = 0;
              for(int j = 0; j < 4; ++j) if(w[j]) { o[mm++] = w[j]; }
              *reinterpret_cast<uint4*>(s_tile + b2 * 8 + 4 * h) = make_uint4(o[0], o[1], o[2], o[3]);
            }
          }
        }
        lds_barrier();
      }
      const uint32_t nq = MODE >= 7 ? 0u : (s_q[0] < 2047 ? s_q[0] : 2047);
      for(uint32_t i = threadIdx.x; i < nq; i += kThreads) {
        const uint32_t xq = s_q[1 + i];
        const uint32_t half = xq >> kTagBits, tag = xq & (kOcc - 1), idx0 = tag >> kRem;
        const uint32_t low = kOcc | tag, neww = kInc | low;
        uint32_t* tl = s_tile + half * 8192;
        for(uint32_t p = 0; p < 8192; ++p) {
          const uint32_t slot = ((idx0 & ~3u) + p) & 8191u;
          const uint32_t o = atomicCAS(&tl[slot], 0u, neww);
          if(o == 0u) break;
          if((o & kLow) == low) { atomicAdd(&tl[slot], kInc); break; }
        }
      }
    } else {
      uint32_t rk[kNP];
#pragma unroll
      for(int r = 0; r < kNP; ++r) if(r < n) rk[r] = atomicAdd(&s_cnt[(it[r] >> (kRem + 3)) & (kBuckets - 1)], 1u);
#pragma unroll
      for(int r = 0; r < kNP; ++r) if(r < n) {
        const uint32_t b = (it[r] >> (kRem + 3)) & (kBuckets - 1);
        if(rk[r] < 8) s_tile[b * 8 + rk[r]] = kInc | kOcc | (it[r] & (kOcc - 1));
        else { const uint32_t q = atomicAdd(&s_q[0], 1u); if(q < 2047) s_q[1 + q] = it[r]; }
      }
      lds_barrier();
      if(MODE == 2) {
        for(uint32_t b = threadIdx.x; b < kBuckets; b += kThreads) {
          const uint32_t c = s_cnt[b];
          if(c < 2) continue;
          uint4 v0 = *reinterpret_cast<const uint4*>(s_tile + b * 8), v1 = *reinterpret_cast<const uint4*>(s_tile + b * 8 + 4);
          uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          bool dup = false;
#pragma unroll
          for(int j = 1; j < 8; ++j)
#pragma unroll
            for(int i = 0; i < j; ++i) dup |= (((w[i] ^ w[j]) & kLow) == 0) & (w[j] != 0);
          if(dup) {
#pragma unroll
            for(int j = 1; j < 8; ++j)
#pragma unroll
              for(int i = 0; i < j; ++i)
                if(w[j] != 0 && ((w[i] ^ w[j]) & kLow) == 0) { w[i] += w[j] & ~kLow; w[j] = 0; }
            uint32_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int m = 0;      // compact to the front
#pragma unroll
            for(int j = 0; j < 8; ++j) if(w[j]) {
#pragma unroll
              for(int q = 0; q < 8; ++q) if(q == m) o[q] = w[j];
              ++m;
            }
            *reinterpret_cast<uint4*>(s_tile + b * 8) = make_uint4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<uint4*>(s_tile + b * 8 + 4) = make_uint4(o[4], o[5], o[6], o[7]);
          }
        }
        lds_barrier();
      }
      const uint32_t nq = s_q[0] < 2047 ? s_q[0] : 2047;
      for(uint32_t i = threadIdx.x; i < nq; i += kThreads) {
        const uint32_t xq = s_q[1 + i];
        const uint32_t half = xq >> kTagBits, tag = xq & (kOcc - 1), idx0 = tag >> kRem;
        const uint32_t low = kOcc | tag, neww = kInc | low;
        uint32_t* tl = s_tile + half * 8192;
        for(uint32_t p = 0; p < 8192; ++p) {
          const uint32_t slot = ((idx0 & ~7u) + p) & 8191u;
          const uint32_t old = atomicCAS(&tl[slot], 0u, neww);
          if(old == 0u) break;
          if((old & kLow) == low) { atomicAdd(&tl[slot], kInc); break; }
        }
      }
    }
    lds_barrier();
    for(uint32_t i = threadIdx.x * 4; i < kSlots; i += kThreads * 4) { const uint4 v = *reinterpret_cast<const uint4*>(s_tile + i); acc += v.x ^ v.y ^ v.z ^ v.w; }
    lds_barrier();
  }
  if(acc == 0xdeadbeef) out[0] = acc;
  if(threadIdx.x == 0 && blockIdx.x == 0) out[1 + MODE] = s_q[0];
}

template <int MODE> double trun(uint32_t* d_out, int blocks, uint32_t dup_shift) {
  const size_t lds = (kSlots + kBuckets + 2048) * 4;
  hipFuncSetAttribute((const void*)tprobe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  tprobe<MODE><<<blocks, kThreads, lds>>>(d_out, dup_shift); hipDeviceSynchronize();
  hipEventRecord(a); tprobe<MODE><<<blocks, kThreads, lds>>>(d_out, dup_shift); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

// ---- hash ----
constexpr int kHIter = 256;
template <int MODE>
__global__ __launch_bounds__(kThreads) void hprobe(const uint64_t* __restrict__ tbl, uint32_t* out) {
  extern __shared__ __align__(16) uint32_t s_raw[];
  uint64_t* s64 = reinterpret_cast<uint64_t*>(s_raw);
  // fill the tables with something (content does not matter for timing, layout does)
  const uint32_t words = MODE == 0 ? 6 * 256 * 2 : MODE == 1 ? 6 * 256 : MODE == 2 ? (2 * 2048 + 2 * 1024) : MODE == 3 ? (2 * 16384 + 1024) : 0;
  for(uint32_t i = threadIdx.x; i < words; i += kThreads) s_raw[i] = mix(i + 7);
  __syncthreads();
  uint32_t acc = 0;
  uint64_t key = ((uint64_t)mix(threadIdx.x + blockIdx.x * 1024u) << 20) ^ mix(threadIdx.x * 77u + 5u);
  const uint64_t row0 = tbl[0], row1 = tbl[1];
  for(int itn = 0; itn < kHIter; ++itn) {
#pragma unroll
    for(int j = 0; j < 16; ++j) {
      key = ((key << 2) | ((key >> 40) & 3)) & ((1ull << 42) - 1); key ^= (uint64_t)(j * 0x9E37u + itn);
      const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
      if(MODE == 0) {
        uint64_t pos = 0;
#pragma unroll
        for(int b = 0; b < 6; ++b) { const uint32_t w = b < 4 ? lo : hi; pos ^= s64[b * 256 + ((w >> (8 * (b & 3))) & 0xFFu)]; }
        acc ^= (uint32_t)pos ^ (uint32_t)(pos >> 32);
      } else if(MODE == 1) {
        uint32_t pos = 0;
#pragma unroll
        for(int b = 0; b < 6; ++b) { const uint32_t w = b < 4 ? lo : hi; pos ^= s_raw[b * 256 + ((w >> (8 * (b & 3))) & 0xFFu)]; }
        const uint32_t p0 = __popcll(key & row0) & 1, p1 = __popcll(key & row1) & 1;
        acc ^= pos ^ (p0 << 3) ^ (p1 << 7);
      } else if(MODE == 2) {
        uint32_t pos = s_raw[lo & 2047u] ^ s_raw[2048 + ((lo >> 11) & 2047u)] ^ s_raw[4096 + ((uint32_t)(key >> 22) & 1023u)] ^ s_raw[5120 + ((uint32_t)(key >> 32) & 1023u)];
        const uint32_t p0 = __popcll(key & row0) & 1, p1 = __popcll(key & row1) & 1;
        acc ^= pos ^ (p0 << 3) ^ (p1 << 7);
      } else if(MODE == 3) {
        uint32_t pos = s_raw[lo & 16383u] ^ s_raw[16384 + ((lo >> 14) & 16383u)] ^ s_raw[32768 + ((uint32_t)(key >> 28) & 1023u)];
        const uint32_t p0 = __popcll(key & row0) & 1, p1 = __popcll(key & row1) & 1;
        acc ^= pos ^ (p0 << 3) ^ (p1 << 7);
      } else {
        uint32_t plo = 0, phi = 0;
#pragma unroll
        for(int c = 0; c < 42; ++c) {
          const uint64_t col = tbl[2 + c];                   // uniform: scalar registers
          const uint32_t m = 0u - (uint32_t)((key >> c) & 1);
          plo ^= m & (uint32_t)col; phi ^= m & (uint32_t)(col >> 32);
        }
        acc ^= plo ^ phi;
      }
    }
  }
  if(acc == 0xdeadbeef) out[0] = acc;
}
template <int MODE> double hrun(const uint64_t* tbl, uint32_t* d_out, int blocks) {
  const size_t lds = 140 * 1024;
  hipFuncSetAttribute((const void*)hprobe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hprobe<MODE><<<blocks, kThreads, lds>>>(tbl, d_out); hipDeviceSynchronize();
  hipEventRecord(a); hprobe<MODE><<<blocks, kThreads, lds>>>(tbl, d_out); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
  uint32_t* d_out; hipMalloc(&d_out, 256); hipMemset(d_out, 0, 256);
  uint64_t h_tbl[64]; for(int i = 0; i < 64; ++i) h_tbl[i] = 0x9E3779B97F4A7C15ull * (i + 3);
  uint64_t* d_tbl; hipMalloc(&d_tbl, sizeof h_tbl); hipMemcpy(d_tbl, h_tbl, sizeof h_tbl, hipMemcpyHostToDevice);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("CUs %d clockRate %d kHz\n", cus, p.clockRate);
  const char* tn[] = {"T0 CAS + triangular probing", "T1 CAS + bucket-linear probing", "T2 rank placement + merge pass", "T3 rank placement, no merge pass", "T4 floor: zero, items, read-out", "T5 rank, buckets of 4, merge pass", "T6 rank, buckets of 4, no merge pass", "T7 = T6 without the overflow queue's inserts", "T8 = only the returning adds"};
  for(uint32_t dup_shift : {0u, 1034u}) {     // pool of distinct items per pair: 0 = all distinct; 4135: two copies each; 1034: eight; 129: sixty-four
    double ms[9] = {trun<0>(d_out, cus * 2, dup_shift), trun<1>(d_out, cus * 2, dup_shift), trun<2>(d_out, cus * 2, dup_shift), trun<3>(d_out, cus * 2, dup_shift), trun<4>(d_out, cus * 2, dup_shift), trun<5>(d_out, cus * 2, dup_shift), trun<6>(d_out, cus * 2, dup_shift), trun<7>(d_out, cus * 2, dup_shift), trun<8>(d_out, cus * 2, dup_shift)};
    uint32_t h_out[12]; hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    for(int m = 0; m < 9; ++m)
      printf("pool %5u  %-36s %8.3f ms  -> %7.2f us per pair per block (2 blocks/CU), %6.3f items/ns/CU-pair   overflow queue %u\n", dup_shift, tn[m], ms[m],
             ms[m] * 1e3 / kTiles, 2.0 * 8270.0 * kTiles / (ms[m] * 1e6), h_out[1 + m]);
  }
  const char* hn[] = {"H0 6 x b64 byte tables", "H1 6 x b32 byte tables + 2 parity rows", "H2 4 x b32 11-bit tables + 2 parity rows", "H3 2 x 14-bit + 1 x 10-bit b32 tables + 2 parity rows", "H4 42 select-XORs (VALU)"};
  double hm[5] = {hrun<0>(d_tbl, d_out, cus), hrun<1>(d_tbl, d_out, cus), hrun<2>(d_tbl, d_out, cus), hrun<3>(d_tbl, d_out, cus), hrun<4>(d_tbl, d_out, cus)};
  for(int m = 0; m < 5; ++m)
    printf("%-56s %8.3f ms  -> %7.2f ns per wave-round (64 keys) per CU, %7.2f G keys/s\n", hn[m], hm[m],
           hm[m] * 1e6 / (kHIter * 16.0 * 16.0), (double)cus * kThreads * kHIter * 16.0 / (hm[m] * 1e6));
  return 0;
}
