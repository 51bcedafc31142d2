// HBM bandwidth of plain streaming kernels on gfx950: read, write zeros, write data, copy.
// hipcc --offload-arch=gfx950 -O3 -o hbm_probe hbm_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint4* __restrict__ a, const uint4* __restrict__ b, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if(MODE == 0) { const uint4 v = b[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    else if(MODE == 1) a[i] = make_uint4(0, 0, 0, 0);
    else if(MODE == 2) { const uint32_t x = (uint32_t)i * 2654435761u; a[i] = make_uint4(x, x ^ 0x9e3779b9u, x * 31u, ~x); }
    else if(MODE == 3) a[i] = b[i];
  }
  if(acc == 0x12345678u) *sink = acc;
}
template <int MODE> double run(uint4* a, uint4* b, size_t n, uint32_t* sink, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 1024>>>(a, b, n, sink); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 1024>>>(a, b, n, sink); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  const size_t bytes = (size_t)32 << 30, n = bytes / 16;
  uint4 *a, *b; uint32_t* sink;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 64);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  k<2><<<4096, 1024>>>(b, a, n, sink); hipDeviceSynchronize();          // b holds non-trivial data
  const char* names[] = {"read 32 GiB", "write zeros 32 GiB", "write data 32 GiB", "copy 32 GiB (read + write)"};
  for(int blocks : {2048, 8192}) {
    double ms[4] = {run<0>(a, b, n, sink, blocks), run<1>(a, b, n, sink, blocks), run<2>(a, b, n, sink, blocks), run<3>(a, b, n, sink, blocks)};
    for(int m = 0; m < 4; ++m) printf("blocks %5d  %-28s %8.3f ms  %7.2f TB/s\n", blocks, names[m], ms[m], (m == 3 ? 2.0 : 1.0) * bytes / ms[m] / 1e9);
  }
  return 0;
}
