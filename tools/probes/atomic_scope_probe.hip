// Random read-modify-write rates on gfx950 by working-set size and by who touches a region.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_scope_probe atomic_scope_probe.hip
// Question behind it: global atomics retire at ~20 G/s on tables >= 1 GB.  Is that the
// memory-side atomic unit (then small working sets are no faster) or DRAM line traffic (then a working
// set that fits the 256 MB Infinity Cache / a 4 MB XCD L2 is)?  And what does a plain load+store RMW
// reach when one workgroup (or one XCD) owns the region -- the non-atomic alternative to LDS tiles?
//   op 0: global_atomic_add_x2, no return (agent scope; workgroup scope emits the same instruction)
//   op 1: 32-bit global_atomic_cmpswap with return
//   op 2: plain 64-bit load, add, store (correct only when nobody else touches the word)
//   op 3: 32-bit byte-granular RMW by CAS loop (the Bloom counter's update)
//   own 0: every block addresses the whole table
//   own 1: the table is cut into regions of `region` bytes; block b only addresses region b % n_regions
//          (b % 8 = XCD under the observed dispatch order, so a region stays on one XCD's L2)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <int OP, int OWN>
__global__ __launch_bounds__(256) void rmw(uint64_t* __restrict__ tab, uint64_t words, uint64_t region_words, uint64_t per_thread,
                                           uint64_t seed, unsigned long long* sink) {
  uint64_t acc = 0;
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t base = 0, span = words;
  if(OWN) { const uint64_t nreg = words / region_words; base = (blockIdx.x % nreg) * region_words; span = region_words; }
#pragma unroll 4
  for(uint64_t i = 0; i < per_thread; ++i) {
    const uint64_t r = mix64(seed + (tid * per_thread + i) * 0x9E3779B97F4A7C15ull);
    const uint64_t w = base + (r % span);
    if(OP == 0) __hip_atomic_fetch_add((unsigned long long*)&tab[w], 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if(OP == 1) acc += atomicCAS((unsigned int*)&tab[w], 0u, (unsigned int)(i | 1));
    else if(OP == 2) { const uint64_t v = tab[w]; tab[w] = v + (1ull << 32) + (v & 1); }
    else {
      unsigned int* p = (unsigned int*)&tab[w];
      const uint32_t sh = 8 * (uint32_t)((r >> 40) & 3);
      unsigned int old = *p;
      while(true) {
        const uint32_t b = (old >> sh) & 0xFF;
        if(b >= 200) break;
        const unsigned int seen = atomicCAS(p, old, old + (1u << sh));
        if(seen == old) break;
        old = seen;
      }
    }
  }
  if(acc == 0x123456789ull) atomicAdd(sink, 1ull);
}

template <int OP, int OWN>
double run(uint64_t* tab, uint64_t words, uint64_t region_words, uint64_t n, unsigned long long* sink) {
  const int blocks = 256 * 8 * 2;
  const uint64_t per = n / ((uint64_t)blocks * 256);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  rmw<OP, OWN><<<blocks, 256>>>(tab, words, region_words, per / 8 + 1, 1, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  rmw<OP, OWN><<<blocks, 256>>>(tab, words, region_words, per, 42, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return (double)per * blocks * 256 / (ms * 1e-3) / 1e9;
}

int main() {
  const uint64_t max_bytes = 16ull << 30;
  uint64_t* tab; unsigned long long* sink;
  if(hipMalloc(&tab, max_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&sink, 64);
  hipMemset(tab, 0, max_bytes);
  const uint64_t n = 1ull << 30;
  printf("# G updates/s; %llu updates per run, 4096 blocks x 256 threads\n", (unsigned long long)n);
  printf("%-14s %10s %10s %10s %10s\n", "working set", "atomic64", "cas32", "ld+st64", "byteCAS");
  for(uint64_t bytes : {1ull << 20, 8ull << 20, 32ull << 20, 128ull << 20, 512ull << 20, 2ull << 30, 16ull << 30}) {
    const uint64_t w = bytes / 8;
    printf("%10llu KiB %10.1f %10.1f %10.1f %10.1f\n", (unsigned long long)(bytes >> 10), run<0, 0>(tab, w, w, n, sink), run<1, 0>(tab, w, w, n, sink),
           run<2, 0>(tab, w, w, n, sink), run<3, 0>(tab, w, w, n, sink));
    fflush(stdout);
  }
  printf("# owned regions inside a 16 GiB table (block b -> region b %% n_regions)\n");
  for(uint64_t region : {8ull << 10, 16ull << 10, 64ull << 10, 512ull << 10, 2ull << 20, 4ull << 20, 16ull << 20}) {
    const uint64_t w = max_bytes / 8, rw = region / 8;
    // only the first 4096 regions are addressed (one per block): the working set is 4096 x region
    printf("region %6llu KiB %7.1f %10.1f %10.1f %10.1f\n", (unsigned long long)(region >> 10), run<0, 1>(tab, w, rw, n, sink), run<1, 1>(tab, w, rw, n, sink),
           run<2, 1>(tab, w, rw, n, sink), run<3, 1>(tab, w, rw, n, sink));
    fflush(stdout);
  }
  return 0;
}
