// What does the partition kernels' write pattern cost?  Every block owns NR regions; per iteration it writes one run of
// R 4-byte items to each of them (consecutive lanes -> consecutive addresses inside a run, the runs of one iteration go
// to NR different regions), the next iteration appends.  R * 4 bytes is the run length; `mis` shifts every region's
// cursor by half a run so runs straddle their natural alignment.  Reported: payload bytes per second.
//   hipcc --offload-arch=gfx950 -O3 -o scatter_write_probe scatter_write_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__global__ __launch_bounds__(1024) void scatter_runs(uint32_t* __restrict__ out, uint32_t nr, uint32_t R, uint32_t iters, uint32_t mis, uint64_t region_items) {
  // chunk = 16384 items per iteration = (16384 / R) runs, spread round-robin over the nr regions
  const uint32_t runs = 16384u / R;
  uint32_t* base = out + (uint64_t)blockIdx.x * nr * region_items;
  for(uint32_t it = 0; it < iters; ++it) {
#pragma unroll 4
    for(uint32_t i = threadIdx.x; i < 16384u; i += 1024u) {
      const uint32_t run = i / R, w = i % R;
      const uint64_t g = (uint64_t)it * runs + run;                          // this block's g-th run
      const uint32_t region = ((uint32_t)(g % nr) * 40503u) & (nr - 1);      // nr is a power of two: a bijection
      const uint64_t cur = (g / nr) * R + (mis ? R / 2 : 0);
      base[(uint64_t)region * region_items + cur + w] = i ^ it;
    }
  }
}

int main(int argc, char** argv) {
  const uint32_t blocks = argc > 1 ? atoi(argv[1]) : 256;
  const uint64_t total_items = 1ull << 31;      // 8 GiB of payload
  uint32_t* d; const uint64_t alloc_items = total_items * 2 + (1ull << 24);
  if(hipMalloc((void**)&d, alloc_items * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 0, alloc_items * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(uint32_t nr : {1024u, 2048u})
    for(uint32_t R : {4u, 8u, 16u, 32u, 64u, 128u, 512u})
      for(uint32_t mis : {0u, 1u}) {
        const uint32_t runs = 16384u / R;
        const uint32_t iters = (uint32_t)(total_items / ((uint64_t)blocks * 16384u));
        const uint64_t region_items = ((uint64_t)iters * runs / nr + 2) * R;
        if((uint64_t)blocks * nr * region_items > alloc_items) { printf("nr %u R %u: skipped (needs %.1f GiB)\n", nr, R, blocks * (double)nr * region_items * 4 / (1ull << 30)); continue; }
        float best = 1e30f;
        for(int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          hipLaunchKernelGGL(scatter_runs, dim3(blocks), dim3(1024), 0, 0, d, nr, R, iters, mis, region_items);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1); if(ms < best) best = ms;
        }
        const double bytes = (double)blocks * iters * 16384.0 * 4;
        printf("blocks %u regions/block %4u run %4u B %s  %8.3f ms  %6.2f TB/s payload\n", blocks, nr, R * 4, mis ? "misaligned" : "aligned   ", best, bytes / best / 1e9);
        fflush(stdout);
      }
  return 0;
}
