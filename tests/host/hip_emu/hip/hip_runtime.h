// tests/host/hip_emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE, never shipped, never measured.
//
// A host-side stand-in for <hip/hip_runtime.h>, just large enough to compile jellyfish_amd/csrc/jfgpu.hip
// with g++ (-DJFGPU_EMU) and run its kernels' SOURCE on a CPU for debugging before a GPU call is spent:
//   * one workgroup at a time; its work-items are fibers (own stack, hand-written context switch) that run
//     until they reach a barrier, so __syncthreads() / LDS-only barriers have real rendezvous semantics and a
//     divergent barrier is reported as a deadlock instead of hanging;
//   * wave64 cross-lane operations (__shfl_*) rendezvous the live lanes of one wave;
//   * __shared__ is block-lifetime static storage, dynamic LDS a per-block arena (poisoned at block start);
//   * device memory is host memory (poisoned at allocation: nothing may rely on fresh pages being zero);
//   * streams and events are synchronous.
// What it cannot show: data races, memory-model and visibility bugs, occupancy, performance.  The product
// library (jellyfish_amd/lib/libjfgpu.so) is built by hipcc from the same sources and has no CPU path; the
// library built against this header is loaded only by tests that ask for it explicitly (JFGPU_EMU=1).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#if !defined(__x86_64__)
#error "hip_emu: the fiber switch is written for x86-64"
#endif

#define __host__
#define __device__
#define __global__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __align__(x) __attribute__((aligned(x)))

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { uint32_t x, y, z, w; } __attribute__((aligned(16)));
struct uint2 { uint32_t x, y; } __attribute__((aligned(8)));
struct ulonglong2 { unsigned long long x, y; } __attribute__((aligned(16)));
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 v; v.x = x; v.y = y; return v; }
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { ulonglong2 v; v.x = x; v.y = y; return v; }

// ---------------------------------------------------------------- host API (synchronous)
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorNoDevice = 100, hipErrorInvalidValue = 1 };
typedef struct hip_emu_stream_* hipStream_t;
struct hip_emu_event_ { double t_ms; };
typedef hip_emu_event_* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char name[64]; int multiProcessorCount; size_t totalGlobalMem; };

namespace hip_emu {
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline size_t& mem_in_use() { static size_t v = 0; return v; }
inline size_t mem_total() { const char* e = getenv("JFGPU_EMU_MEM_MB"); return (size_t)(e ? atol(e) : 49152) << 20; }
inline int n_cus() { const char* e = getenv("JFGPU_EMU_CUS"); const int v = e ? atoi(e) : 2; return v > 0 ? v : 2; }
struct AllocHeader { size_t bytes; size_t magic; char pad[256 - 2 * sizeof(size_t)]; };
}  // namespace hip_emu

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory (emu)" : "error (emu)"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof *p); strcpy(p->name, "hip_emu (host fibers)"); p->multiProcessorCount = hip_emu::n_cus(); p->totalGlobalMem = hip_emu::mem_total();
  return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
  *total_b = hip_emu::mem_total(); *free_b = *total_b > hip_emu::mem_in_use() ? *total_b - hip_emu::mem_in_use() : 0; return hipSuccess;
}
template <typename T> static inline hipError_t hipMalloc(T** p, size_t bytes) {
  using hip_emu::AllocHeader;
  if(hip_emu::mem_in_use() + bytes > hip_emu::mem_total()) { *p = nullptr; return hipErrorOutOfMemory; }
  const size_t total = sizeof(AllocHeader) + bytes + 256;      // slack: kernels read whole 16-byte vectors at buffer ends
  void* raw = nullptr;
  if(total > ((size_t)64 << 20)) {                             // big tables: untouched pages stay virtual
    raw = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if(raw == MAP_FAILED) { *p = nullptr; return hipErrorOutOfMemory; }
  } else {
    if(posix_memalign(&raw, 256, total) != 0) { *p = nullptr; return hipErrorOutOfMemory; }
    memset(raw, 0xAB, total);                                  // poison: device memory is not zero-initialised
  }
  AllocHeader* h = (AllocHeader*)raw; h->bytes = bytes; h->magic = total > ((size_t)64 << 20) ? 0x6D6D6170 : 0x68656170;
  hip_emu::mem_in_use() += bytes;
  *p = (T*)((char*)raw + sizeof(AllocHeader));
  return hipSuccess;
}
static inline hipError_t hipFree(void* p) {
  using hip_emu::AllocHeader;
  if(!p) return hipSuccess;
  AllocHeader* h = (AllocHeader*)((char*)p - sizeof(AllocHeader));
  hip_emu::mem_in_use() -= h->bytes;
  if(h->magic == 0x6D6D6170) munmap(h, sizeof(AllocHeader) + h->bytes + 256);
  else if(h->magic == 0x68656170) { h->magic = 0; free(h); }
  else { fprintf(stderr, "hip_emu: hipFree of a pointer that was not allocated (or freed twice)\n"); abort(); }
  return hipSuccess;
}
template <typename T> static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned = 0) { *p = (T*)malloc(bytes ? bytes : 16); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if(n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if(n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) {
  // zeroing a large (mmap-backed) range: give the interior pages back instead of touching them
  if(v == 0 && n >= ((size_t)64 << 20)) {
    const uintptr_t a = ((uintptr_t)d + 4095) & ~(uintptr_t)4095, b = ((uintptr_t)d + n) & ~(uintptr_t)4095;
    if(b > a && madvise((void*)a, b - a, MADV_DONTNEED) == 0) {
      memset(d, 0, a - (uintptr_t)d); memset((void*)b, 0, (uintptr_t)d + n - b);
      return hipSuccess;
    }
  }
  if(n) memset(d, v, n);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { return hipMemset(d, v, n); }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)(uintptr_t)1; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)(uintptr_t)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hip_emu_event_{0.0}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t_ms = hip_emu::now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

// ---------------------------------------------------------------- fibers
extern "C" void hip_emu_switch(void** save_sp, void* new_sp);
asm(R"(
    .text
    .globl hip_emu_switch
    .type hip_emu_switch,@function
hip_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hip_emu_switch, .-hip_emu_switch
)");

namespace hip_emu {

enum FiberState { F_READY = 0, F_WAIT_BLOCK = 1, F_WAIT_WAVE = 2, F_DONE = 3 };
struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  int state = F_DONE;
  dim3 thread_idx;
  unsigned linear = 0;
};
constexpr size_t kStackBytes = 256 << 10;
constexpr size_t kDynLdsBytes = 160 << 10;

struct BlockRun {
  std::vector<Fiber> fibers;
  void* sched_sp = nullptr;
  Fiber* cur = nullptr;
  const void* body = nullptr;                 // pointer to the launch lambda
  void (*invoke)(const void*) = nullptr;
  unsigned char* dyn_lds = nullptr;
  unsigned long long xch[1024][2];            // per-lane exchange slots for __shfl_*
  dim3 block_idx, block_dim, grid_dim;
};
inline BlockRun*& tls_run() { static thread_local BlockRun* r = nullptr; return r; }

inline void yield_to_scheduler() { BlockRun* R = tls_run(); hip_emu_switch(&R->cur->sp, R->sched_sp); }
inline void block_barrier() { tls_run()->cur->state = F_WAIT_BLOCK; yield_to_scheduler(); }
inline void wave_barrier() { tls_run()->cur->state = F_WAIT_WAVE; yield_to_scheduler(); }

extern "C" inline void hip_emu_fiber_main() {
  BlockRun* R = tls_run();
  R->invoke(R->body);
  R = tls_run();
  R->cur->state = F_DONE;
  yield_to_scheduler();
  abort();                                    // a finished fiber is never resumed
}

inline void prepare_fiber(Fiber& f) {
  if(!f.stack) {
    f.stack = (char*)mmap(nullptr, kStackBytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_STACK, -1, 0);
    if(f.stack == (char*)MAP_FAILED) { perror("hip_emu: fiber stack"); abort(); }
  }
  // the switch pops r15 r14 r13 r12 rbx rbp and returns into hip_emu_fiber_main with rsp = 8 (mod 16), as after a call
  uintptr_t top = ((uintptr_t)f.stack + kStackBytes) & ~(uintptr_t)15;
  void** sp = (void**)top;
  *--sp = nullptr;                            // fake return address of fiber_main (keeps the alignment rule)
  *--sp = (void*)&hip_emu_fiber_main;
  for(int i = 0; i < 6; ++i) *--sp = nullptr;
  f.sp = sp;
  f.state = F_READY;
}

inline void run_block(BlockRun& R, unsigned nthreads) {
  unsigned live = nthreads;
  memset(R.dyn_lds, 0xCD, kDynLdsBytes);
  for(unsigned i = 0; i < nthreads; ++i) prepare_fiber(R.fibers[i]);
  const unsigned nwaves = (nthreads + 63) / 64;
  while(live) {
    bool ran = false;
    for(unsigned i = 0; i < nthreads; ++i) {
      Fiber& f = R.fibers[i];
      if(f.state != F_READY) continue;
      R.cur = &f;
      hip_emu_switch(&R.sched_sp, f.sp);
      ran = true;
      if(f.state == F_DONE) --live;
    }
    // release what can be released
    bool released = false;
    for(unsigned w = 0; w < nwaves; ++w) {
      unsigned alive = 0, waiting = 0;
      const unsigned lo = w * 64, hi = lo + 64 < nthreads ? lo + 64 : nthreads;
      for(unsigned i = lo; i < hi; ++i) { alive += R.fibers[i].state != F_DONE; waiting += R.fibers[i].state == F_WAIT_WAVE; }
      if(waiting && waiting == alive) { for(unsigned i = lo; i < hi; ++i) if(R.fibers[i].state == F_WAIT_WAVE) R.fibers[i].state = F_READY; released = true; }
    }
    if(!released && live) {
      unsigned at_block = 0;
      for(unsigned i = 0; i < nthreads; ++i) at_block += R.fibers[i].state == F_WAIT_BLOCK;
      if(at_block == live) { for(unsigned i = 0; i < nthreads; ++i) if(R.fibers[i].state == F_WAIT_BLOCK) R.fibers[i].state = F_READY; released = true; }
    }
    if(live && !ran && !released) {
      unsigned nb = 0, nw = 0;
      for(unsigned i = 0; i < nthreads; ++i) { nb += R.fibers[i].state == F_WAIT_BLOCK; nw += R.fibers[i].state == F_WAIT_WAVE; }
      fprintf(stderr, "hip_emu: deadlock in block (%u,%u,%u): %u live work-items, %u at a block barrier, %u at a wave operation "
                      "(divergent barrier or cross-lane operation under divergent control flow)\n",
              R.block_idx.x, R.block_idx.y, R.block_idx.z, live, nb, nw);
      abort();
    }
  }
}

template <typename F> inline void invoke_thunk(const void* p) { (*(const F*)p)(); }

inline BlockRun* my_run() {
  static thread_local BlockRun* R = nullptr;
  if(!R) { R = new BlockRun; R->fibers.resize(1024); R->dyn_lds = (unsigned char*)aligned_alloc(256, kDynLdsBytes); }
  return R;
}

// One launch = its blocks handed out to a few OS threads (each runs one block at a time as fibers).
struct Job {
  dim3 grid, block; unsigned nthreads = 0;
  const void* body = nullptr; void (*invoke)(const void*) = nullptr;
  std::atomic<uint64_t> next{0}; uint64_t total = 0;
};
inline void work_on(Job& J) {
  BlockRun* R = my_run();
  BlockRun* prev = tls_run();
  tls_run() = R;
  R->body = J.body; R->invoke = J.invoke; R->block_dim = J.block; R->grid_dim = J.grid;
  for(unsigned i = 0; i < J.nthreads; ++i) {
    Fiber& f = R->fibers[i];
    f.linear = i; f.thread_idx = dim3(i % J.block.x, (i / J.block.x) % J.block.y, i / (J.block.x * J.block.y));
  }
  for(uint64_t b; (b = J.next.fetch_add(1)) < J.total; ) {
    R->block_idx = dim3((unsigned)(b % J.grid.x), (unsigned)((b / J.grid.x) % J.grid.y), (unsigned)(b / ((uint64_t)J.grid.x * J.grid.y)));
    run_block(*R, J.nthreads);
  }
  tls_run() = prev;
}
struct Pool {
  std::vector<std::thread> workers;
  std::mutex mu; std::condition_variable cv_work, cv_done;
  Job* job = nullptr; uint64_t generation = 0; unsigned busy = 0;
  Pool() {
    const char* e = getenv("JFGPU_EMU_THREADS");
    unsigned n = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
    if(n < 1) n = 1;
    if(n > 16) n = 16;
    for(unsigned i = 1; i < n; ++i) workers.emplace_back([this]() {
      uint64_t seen = 0;
      std::unique_lock<std::mutex> lk(mu);
      while(true) {
        cv_work.wait(lk, [&]() { return generation != seen; });
        seen = generation;
        Job* j = job;
        if(!j) continue;
        ++busy; lk.unlock();
        work_on(*j);
        lk.lock(); --busy;
        if(busy == 0) cv_done.notify_all();
      }
    });
    for(auto& w : workers) w.detach();
  }
  void run(Job& J) {
    if(workers.empty() || J.total < 2) { work_on(J); return; }
    { std::lock_guard<std::mutex> lk(mu); job = &J; ++generation; }
    cv_work.notify_all();
    work_on(J);
    std::unique_lock<std::mutex> lk(mu);
    job = nullptr;                              // late wakers find nothing to do
    cv_done.wait(lk, [&]() { return busy == 0; });
  }
};
inline Pool& pool() { static Pool* p = new Pool; return *p; }

template <typename F>
inline void launch(dim3 grid, dim3 block, size_t dyn_lds_bytes, const F& body) {
  const unsigned nthreads = block.x * block.y * block.z;
  if(nthreads == 0 || nthreads > 1024) { fprintf(stderr, "hip_emu: bad block size %u\n", nthreads); abort(); }
  if(dyn_lds_bytes > kDynLdsBytes) { fprintf(stderr, "hip_emu: %zu bytes of dynamic LDS requested (> 160 KiB)\n", dyn_lds_bytes); abort(); }
  Job J;
  J.grid = grid; J.block = block; J.nthreads = nthreads; J.body = &body; J.invoke = &invoke_thunk<F>;
  J.total = (uint64_t)grid.x * grid.y * grid.z;
  if(J.total == 0) return;
  if(tls_run()) { fprintf(stderr, "hip_emu: kernel launch from inside a kernel\n"); abort(); }
  pool().run(J);
}

inline unsigned char* dyn_lds() { return tls_run()->dyn_lds; }

template <typename T> inline T lane_exchange(T v, int src_lane_delta_kind, unsigned arg) {
  // kind 0: down (read lane + arg), 1: up (read lane - arg), 2: xor
  static_assert(sizeof(T) <= 16, "shuffle operand too wide");
  BlockRun* R = tls_run();
  const unsigned me = R->cur->linear, lane = me & 63, base = me & ~63u;
  const unsigned nthreads = R->block_dim.x * R->block_dim.y * R->block_dim.z;
  memcpy(R->xch[me], &v, sizeof(T));
  wave_barrier();
  R = tls_run();
  int src = (int)lane;
  if(src_lane_delta_kind == 0) src = (int)lane + (int)arg; else if(src_lane_delta_kind == 1) src = (int)lane - (int)arg; else src = (int)(lane ^ arg);
  T out = v;
  if(src >= 0 && src < 64 && base + (unsigned)src < nthreads && R->fibers[base + src].state != F_DONE) memcpy(&out, R->xch[base + src], sizeof(T));
  wave_barrier();
  return out;
}

// __ballot: the predicate bits of the wave's live lanes; __shfl: the value of an absolute lane.
inline unsigned long long wave_ballot(int pred) {
  BlockRun* R = tls_run();
  const unsigned me = R->cur->linear, base = me & ~63u;
  const unsigned nthreads = R->block_dim.x * R->block_dim.y * R->block_dim.z;
  const unsigned char p = pred ? 1 : 0;
  memcpy(R->xch[me], &p, 1);
  wave_barrier();
  R = tls_run();
  unsigned long long m = 0;
  for(unsigned l = 0; l < 64 && base + l < nthreads; ++l)
    if(R->fibers[base + l].state != F_DONE && *(unsigned char*)R->xch[base + l]) m |= 1ull << l;
  wave_barrier();
  return m;
}
template <typename T> inline T lane_read(T v, int src) {
  BlockRun* R = tls_run();
  const unsigned me = R->cur->linear, base = me & ~63u;
  const unsigned nthreads = R->block_dim.x * R->block_dim.y * R->block_dim.z;
  memcpy(R->xch[me], &v, sizeof(T));
  wave_barrier();
  R = tls_run();
  T out = v;
  if(src >= 0 && src < 64 && base + (unsigned)src < nthreads && R->fibers[base + src].state != F_DONE) memcpy(&out, R->xch[base + src], sizeof(T));
  wave_barrier();
  return out;
}

}  // namespace hip_emu

#define threadIdx (::hip_emu::tls_run()->cur->thread_idx)
#define blockIdx (::hip_emu::tls_run()->block_idx)
#define blockDim (::hip_emu::tls_run()->block_dim)
#define gridDim (::hip_emu::tls_run()->grid_dim)

#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) \
  ::hip_emu::launch(dim3(grid), dim3(block), (size_t)(lds), [&]() { kern(__VA_ARGS__); })

// ---------------------------------------------------------------- device builtins
static inline void __syncthreads() { ::hip_emu::block_barrier(); }
template <typename T> static inline T __shfl_down(T v, unsigned o, int = 64) { return ::hip_emu::lane_exchange(v, 0, o); }
template <typename T> static inline T __shfl_up(T v, unsigned o, int = 64) { return ::hip_emu::lane_exchange(v, 1, o); }
template <typename T> static inline T __shfl_xor(T v, unsigned o, int = 64) { return ::hip_emu::lane_exchange(v, 2, o); }
static inline unsigned long long __ballot(int pred) { return ::hip_emu::wave_ballot(pred); }
template <typename T> static inline T __shfl(T v, int src, int = 64) { return ::hip_emu::lane_read(v, src); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline long long clock64() { return (long long)__builtin_ia32_rdtsc(); }

#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
template <typename T, typename U> static inline T __hip_atomic_fetch_add(T* p, U v, int, int) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> static inline T __hip_atomic_fetch_or(T* p, U v, int, int) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> static inline T __hip_atomic_fetch_max(T* p, U v, int, int) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while(old < (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <typename T> static inline T __hip_atomic_load(const T* p, int, int) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
template <typename T, typename U> static inline void __hip_atomic_store(T* p, U v, int, int) { __atomic_store_n(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> static inline T atomicAdd(T* p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> static inline T atomicSub(T* p, U v) { return __atomic_fetch_sub(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> static inline T atomicOr(T* p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> static inline T atomicXor(T* p, U v) { return __atomic_fetch_xor(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> static inline T atomicAnd(T* p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> static inline T atomicExch(T* p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_RELAXED); }
template <typename T, typename U> static inline T atomicMax(T* p, U v) { return __hip_atomic_fetch_max(p, v, 0, 0); }
template <typename T, typename U> static inline T atomicMin(T* p, U v) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while(old > (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <typename T, typename U, typename V> static inline T atomicCAS(T* p, U cmp, V val) {
  T expected = (T)cmp;
  __atomic_compare_exchange_n(p, &expected, (T)val, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return expected;
}
