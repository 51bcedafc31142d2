// Micro-benchmark: LDS atomic throughput on gfx950 (per CU), to price the ranking step of the
// partition kernels.  hipcc --offload-arch=gfx950 -O3 -o lds_atomic_probe lds_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr int kThreads = 1024, kIters = 512;

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(kThreads) void probe(uint32_t* out, uint32_t nb_mask) {
  __shared__ uint32_t s[8192];
  __shared__ unsigned long long s64[4096];
  for(int i = threadIdx.x; i < 8192; i += kThreads) s[i] = 0;
  for(int i = threadIdx.x; i < 4096; i += kThreads) s64[i] = 0;
  __syncthreads();
  uint32_t acc = 0, x = mix(threadIdx.x * 977u + blockIdx.x * 131071u + 1u);
#pragma unroll 4
  for(int it = 0; it < kIters; ++it) {
    x = x * 1664525u + 1013904223u;
    const uint32_t b = (x >> 10) & nb_mask;
    if(MODE == 0) atomicAdd(&s[b], 1u);                                  // non-returning
    else if(MODE == 1) acc += atomicAdd(&s[b], 1u);                      // returning
    else if(MODE == 2) { const uint32_t v = s[b]; s[b] = v + 1; acc += v; }   // plain read + write (racy; timing only)
    else if(MODE == 3) acc += (uint32_t)atomicCAS(&s64[b & 4095], 0ull, (unsigned long long)x);   // 64-bit CAS returning
    else if(MODE == 4) {                                                 // ballot match-any on 10 bits + mbcnt
      uint64_t peers = ~0ull;
#pragma unroll
      for(int bit = 0; bit < 10; ++bit) {
        const uint64_t m = __ballot((b >> bit) & 1);
        peers &= ((b >> bit) & 1) ? m : ~m;
      }
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0));
      const uint32_t cnt = __popcll(peers);
      uint32_t base = 0;
      if(rank == 0) { base = s[b]; s[b] = base + cnt; }                  // leader: wave-private row in the real kernel
      acc += rank + cnt + base;
    } else if(MODE == 5) {                                               // 6 byte-table lookups of 8 B (the GF(2) hash)
      const uint64_t key = ((uint64_t)x << 20) ^ mix(x);
      uint64_t h = 0;
#pragma unroll
      for(int t = 0; t < 6; ++t) h ^= s64[t * 256 + ((key >> (8 * t)) & 255)];
      acc += (uint32_t)h ^ (uint32_t)(h >> 32);
    } else if(MODE == 6) {                                               // rolling GF(2^42) update, both strands, VALU only
      uint64_t hf = x, hr = mix(x);
      const uint64_t P = 0x40000000801ull, top = 1ull << 42;
#pragma unroll
      for(int r = 0; r < 1; ++r) {
        hf <<= 1; if(hf & top) hf ^= P; hf <<= 1; if(hf & top) hf ^= P;
        hf ^= (x & 1 ? 0x123456789ull : 0) ^ (x & 2 ? 0x2468ace13ull : 0) ^ (x & 4 ? 0x3579bdf01ull : 0) ^ (x & 8 ? 0x0f0f0f0f0full : 0);
        if(hr & 1) hr ^= P; hr >>= 1; if(hr & 1) hr ^= P; hr >>= 1;
        hr ^= (x & 16 ? 0x123456789ull : 0) ^ (x & 32 ? 0x2468ace13ull : 0) ^ (x & 64 ? 0x3579bdf01ull : 0) ^ (x & 128 ? 0x0f0f0f0f0full : 0);
      }
      acc += (uint32_t)(hf ^ hr) ^ (uint32_t)((hf ^ hr) >> 32);
    }
  }
  if(acc == 0xdeadbeef) out[0] = acc;
  __syncthreads();
  if(threadIdx.x == 0 && blockIdx.x == 0) out[1] = s[0] + (uint32_t)s64[0];
}

template <int MODE>
double run(uint32_t* d_out, uint32_t mask, int blocks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<MODE><<<blocks, kThreads>>>(d_out, mask);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<MODE><<<blocks, kThreads>>>(d_out, mask);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  uint32_t* d_out; hipMalloc(&d_out, 64);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, blocks = cus * 8;
  const double clk = p.clockRate * 1e3;   // Hz
  printf("CUs %d clock %.0f MHz; %d blocks x %d threads x %d iters\n", cus, clk / 1e6, blocks, kThreads, kIters);
  const char* names[] = {"atomicAdd no-return", "atomicAdd returning", "plain read+write", "64-bit CAS returning", "ballot match-any(10b)+leader rmw", "6x ds_read_b64 byte tables", "rolling GF(2^42) both strands"};
  for(uint32_t mask : {1023u}) {
    double ms[7] = {run<0>(d_out, mask, blocks), run<1>(d_out, mask, blocks), run<2>(d_out, mask, blocks), run<3>(d_out, mask, blocks), run<4>(d_out, mask, blocks), run<5>(d_out, mask, blocks), run<6>(d_out, mask, blocks)};
    for(int m = 0; m < 7; ++m) {
      const double ops = (double)blocks * kThreads * kIters;
      const double per_cu_per_clk = ops / (ms[m] * 1e-3) / cus / clk;
      printf("buckets %5u  %-34s %8.3f ms  %7.2f G ops/s  %6.3f lane-ops/clk/CU\n", mask + 1, names[m], ms[m], ops / ms[m] / 1e6, per_cu_per_clk);
    }
  }
  return 0;
}
