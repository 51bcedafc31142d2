// Where does the host side of the feed go?  A file in /dev/shm read into a 320 MiB buffer by T threads (pread of equal
// slices, like device_sequence_parser::read_slices), for: plain malloc'd memory, hipHostMalloc default / non-coherent /
// write-combined / NUMA-user memory, and hipHostRegister'ed malloc memory; then the upload of that buffer.
//   hipcc -O2 -o feed_probe feed_probe.hip -pthread;  feed_probe <file>
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double read_into(int fd, size_t file_size, char* dst, size_t len, unsigned nt, int rounds) {
  double best = 1e30;
  for(int r = 0; r < rounds; ++r) {
    const size_t off0 = ((size_t)r * len) % (file_size - len);
    const double t0 = now();
    std::vector<std::thread> th;
    const size_t per = (len + nt - 1) / nt;
    for(unsigned i = 0; i < nt; ++i)
      th.emplace_back([=]() {
        size_t o = (size_t)i * per; const size_t e = std::min(len, o + per);
        while(o < e) { const ssize_t g = pread(fd, dst + o, e - o, (off_t)(off0 + o)); if(g <= 0) break; o += (size_t)g; }
      });
    for(auto& t : th) t.join();
    best = std::min(best, now() - t0);
  }
  return len / best / 1e9;
}

int main(int argc, char** argv) {
  if(argc < 2) return 2;
  const int fd = open(argv[1], O_RDONLY);
  struct stat st; fstat(fd, &st);
  const size_t len = (size_t)320 << 20;
  if((size_t)st.st_size < 2 * len) { printf("file too small\n"); return 1; }
  char* d_buf; hipMalloc((void**)&d_buf, len);
  hipStream_t s; hipStreamCreate(&s);
  struct Kind { const char* name; char* p; };
  std::vector<Kind> kinds;
  { char* p = (char*)aligned_alloc(4096, len); memset(p, 1, len); kinds.push_back({"malloc (pageable)", p}); }
  { char* p = nullptr; if(hipHostMalloc((void**)&p, len, hipHostMallocDefault) == hipSuccess) kinds.push_back({"hipHostMalloc default", p}); }
  { char* p = nullptr; if(hipHostMalloc((void**)&p, len, hipHostMallocNonCoherent) == hipSuccess) kinds.push_back({"hipHostMalloc non-coherent", p}); }
  { char* p = nullptr; if(hipHostMalloc((void**)&p, len, hipHostMallocNumaUser) == hipSuccess) kinds.push_back({"hipHostMalloc numa-user", p}); }
  { char* p = (char*)aligned_alloc(4096, len); memset(p, 1, len); if(hipHostRegister(p, len, hipHostRegisterDefault) == hipSuccess) kinds.push_back({"malloc + hipHostRegister", p}); }
  for(auto& k : kinds) {
    printf("%-28s", k.name);
    for(unsigned nt : {8u, 16u, 32u, 64u, 128u}) printf("  %3ut %6.1f GB/s", nt, read_into(fd, st.st_size, k.p, len, nt, 3));
    double best = 1e30;
    for(int r = 0; r < 3; ++r) { const double t0 = now(); hipMemcpyAsync(d_buf, k.p, len, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); best = std::min(best, now() - t0); }
    printf("   upload %6.1f GB/s\n", len / best / 1e9);
    fflush(stdout);
  }
  // registering a window of the file's own pages (no copy at all)
  void* m = mmap(nullptr, st.st_size, PROT_READ, MAP_SHARED, fd, 0);
  if(m != MAP_FAILED) {
    for(unsigned flags : {(unsigned)hipHostRegisterDefault, (unsigned)hipHostRegisterReadOnly}) {
      const double t0 = now();
      const hipError_t e = hipHostRegister(m, len, flags);
      const double t1 = now();
      if(e != hipSuccess) { printf("hipHostRegister(mmap, flags %u): %s\n", flags, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
      hipMemcpyAsync(d_buf, m, len, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
      const double t2 = now();
      hipHostUnregister(m);
      const double t3 = now();
      printf("mmap window, flags %u: register %.1f GB/s, upload %.1f GB/s, unregister %.1f GB/s\n", flags, len / (t1 - t0) / 1e9, len / (t2 - t1) / 1e9, len / (t3 - t2) / 1e9);
    }
  }
  return 0;
}
