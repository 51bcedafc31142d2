// How fast can ONE file on /dev/shm be written from user space?  (the Writing phase of `jellyfish-amd count`: the device
// delivers sorted records at 21 GB/s, the file takes 4.5)   g++ -O2 -pthread -o tmpfs_write_probe tmpfs_write_probe.cc
//   pwrite by N threads | fallocate alone | memcpy by N threads into a shared mapping, fresh pages | ... of pages fallocate made
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const size_t total = (size_t)(argc > 1 ? atof(argv[1]) : 8) << 30, piece = 32u << 20;
  const char* path = argc > 2 ? argv[2] : "/dev/shm/jf_probe.bin";
  std::vector<char> src(piece, 'x');
  auto run = [&](const char* name, int nthreads, int mode) {
    unlink(path);
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if(ftruncate(fd, total) != 0) { perror("ftruncate"); exit(1); }
    char* map = nullptr;
    const double t0 = now();
    double t_alloc = 0;
    if(mode == 1 || mode == 3) { const double a = now(); if(fallocate(fd, 0, 0, total) != 0) perror("fallocate"); t_alloc = now() - a; }
    if(mode >= 2) map = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if(mode != 1) {
      std::vector<std::thread> th;
      for(int t = 0; t < nthreads; ++t) th.emplace_back([&, t]() {
        for(size_t off = (size_t)t * piece; off < total; off += (size_t)nthreads * piece) {
          if(mode == 0) { size_t d = 0; while(d < piece) { ssize_t w = pwrite(fd, src.data() + d, piece - d, off + d); if(w <= 0) return; d += w; } }
          else memcpy(map + off, src.data(), piece);
        }
      });
      for(auto& x : th) x.join();
    }
    const double dt = now() - t0;
    printf("%-46s %2d threads: %6.2f GB/s  (fallocate %.2f s of %.2f s)\n", name, nthreads, total / dt / 1e9, t_alloc, dt);
    if(map) munmap(map, total);
    close(fd); unlink(path);
  };
  for(int n : {1, 4, 16}) run("pwrite", n, 0);
  run("fallocate alone", 1, 1);
  for(int n : {4, 16}) run("memcpy into a shared mapping, fresh pages", n, 2);
  for(int n : {4, 16, 32}) run("fallocate, then memcpy into the mapping", n, 3);
  return 0;
}
