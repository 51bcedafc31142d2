// jellyfish_amd/include/jellyfish_amd/device_parser.hpp
//
// File -> device feed without a per-character host loop: the file is mmap'ed, cut into chunks at
// line (FASTA) or record (FASTQ) boundaries, each chunk is copied to HBM and turned into a contract
// buffer by the device parser (jfgpu_parser_parse, csrc/kernels_parse.hip.hpp).  Same observable
// behaviour as sequence_parser (the host restatement of the reference's
// mer_overlap_sequence_parser.hpp); what the device parser refuses -- FASTQ that is not in strict
// 4-line layout -- is handed to sequence_parser from that point of the file on, so wrapped records
// and the "Invalid fastq sequence" error behave as in the reference (:292-309).
#pragma once
#include <jfgpu.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <thread>

#include "sequence_parser.hpp"

namespace jellyfish_amd {

class device_sequence_parser {
public:
  typedef std::function<void(const char* d_buf, size_t n)> dev_sink_type;   // contract buffer in device memory
  typedef sequence_parser::sink_type host_sink_type;                        // contract buffer in host memory (fallback)
  typedef std::function<void()> fence_type;                                 // "buffers handed out earlier may be reused"

  device_sequence_parser(unsigned mer_len, int device, size_t chunk_bytes = (size_t)256 << 20)
      : k_(mer_len), chunk_(std::min<size_t>(std::max<size_t>(chunk_bytes, 1 << 16), (size_t)1 << 30)), host_(mer_len) {
    if(const char* e = getenv("JFGPU_PARSE_CHUNK")) {          // testing / tuning knob
      const size_t v = strtoull(e, nullptr, 10);
      if(v) chunk_ = std::min<size_t>(std::max<size_t>(v, 1 << 12), (size_t)1 << 30);
    }
    if(const char* e = getenv("JFGPU_FEED_PINNED")) pinned_ = atoi(e) ? 1 : 0;
    if(jfgpu_parser_create(device, mer_len, &p_)) throw std::runtime_error(jfgpu_last_error());
  }
  ~device_sequence_parser() { jfgpu_parser_destroy(p_); }
  device_sequence_parser(const device_sequence_parser&) = delete;
  device_sequence_parser& operator=(const device_sequence_parser&) = delete;

  size_t nb_files() const { return files_read_; }
  size_t nb_reads() const { return reads_read_ + host_.nb_reads(); }
  // count -Q / --min-quality: FASTQ bases below this quality character count as 'N' (device parser and host fallback alike)
  void min_quality(int c) {
    host_.min_quality(c);
    if(jfgpu_parser_set_min_quality(p_, c)) throw std::runtime_error(jfgpu_last_error());
  }
  size_t host_fallback_bytes() const { return fallback_bytes_; }
  double device_ms() const { return device_ms_; }

  // Allocate the pinned staging buffers now if a file of this size will take the pinned path (an Init-phase cost).
  void prepare(size_t largest_file_bytes) {
    const bool pinned = pinned_ < 0 ? largest_file_bytes >= ((size_t)64 << 20) : pinned_ != 0;
    if(!pinned) return;
    const size_t cap = chunk_ + (chunk_ >> 2);
    for(int w = 0; w < 2; ++w)
      if(jfgpu_parser_host_buffer(p_, w, cap, &pin_[w])) throw std::runtime_error(jfgpu_last_error());
  }

  void parse_file(const char* path, const dev_sink_type& dev_sink, const host_sink_type& host_sink, const fence_type& fence) {
    parse_file_part(path, 0, 1, dev_sink, host_sink, fence);
  }

  // Part `part` of `parts` of a regular file, cut at record boundaries (a line starting with '>'; for FASTQ a line starting
  // with '@' whose second-next line starts with '+'): what one rank of a multi-GPU job reads.  Every rank finds the same
  // cut points on its own, the parts are disjoint and cover the file.  Records are never split, so a file of a few long
  // sequences divides unevenly.  Pipes cannot be divided: part 0 reads them whole.
  void parse_file_part(const char* path, unsigned part, unsigned parts, const dev_sink_type& dev_sink, const host_sink_type& host_sink,
                       const fence_type& fence) {
    int fd = open(path, O_RDONLY);
    if(fd < 0) throw std::runtime_error(std::string("Can't open file '") + path + "'");
    struct stat st;
    if(fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) {   // pipes, empty files: the host reader copes
      close(fd);
      if(part == 0) host_.parse_file(path, host_sink);
      return;
    }
    const size_t size = (size_t)st.st_size;
    size_t begin = 0, end = size;
    if(parts > 1) {
      char first = 0;
      if(pread(fd, &first, 1, 0) != 1 || (first != '>' && first != '@')) { close(fd); throw std::runtime_error("Unsupported format"); }
      try {
        begin = part == 0 ? 0 : record_boundary(fd, size, (size_t)((unsigned __int128)size * part / parts), first);
        end = part + 1 == parts ? size : record_boundary(fd, size, (size_t)((unsigned __int128)size * (part + 1) / parts), first);
      } catch(...) { close(fd); throw; }
      if(begin >= end) { close(fd); ++files_read_; return; }
    }
    const bool pinned = pinned_ < 0 ? end - begin >= ((size_t)64 << 20) : pinned_ != 0;
    if(pinned) {
      try { parse_fd_pinned(fd, end, dev_sink, host_sink, fence, begin); } catch(...) { close(fd); throw; }
      close(fd);
      return;
    }
    void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if(m == MAP_FAILED) {
      if(parts > 1) throw std::runtime_error("Can't mmap file");
      host_.parse_file(path, host_sink);
      return;
    }
    madvise(m, size, MADV_SEQUENTIAL);
    try { parse_memory((const char*)m + begin, end - begin, dev_sink, host_sink, fence); } catch(...) { munmap(m, size); throw; }
    munmap(m, size);
  }

  // Offset of the first record that starts at or after `target` (size: none).
  static size_t record_boundary(int fd, size_t size, size_t target, char marker) {
    if(target == 0) return 0;
    const size_t W = (size_t)4 << 20;
    std::vector<char> buf(W + 1);
    size_t pos = target - 1;                     // buf[0] is the character before the first candidate
    while(pos + 1 < size) {
      const size_t len = std::min(W + 1, size - pos);
      if(!read_exact(fd, pos, buf.data(), len)) throw std::runtime_error("Error reading the sequence file");
      for(const char* q = buf.data(); (q = (const char*)memchr(q, '\n', buf.data() + len - 1 - q)) != nullptr; ++q) {
        if(q[1] != marker) continue;
        const size_t at = pos + (size_t)(q - buf.data()) + 1;
        if(marker == '>' || fastq_record_at(fd, size, at)) return at;
      }
      pos += len - 1;
    }
    return size;
  }

  void parse_memory(const char* data, size_t n, const dev_sink_type& dev_sink, const host_sink_type& host_sink, const fence_type& fence) {
    if(n == 0) { ++files_read_; return; }
    unsigned fmt;
    if(data[0] == '>') fmt = JFGPU_PARSE_FASTA;
    else if(data[0] == '@') fmt = JFGPU_PARSE_FASTQ;
    else throw std::runtime_error("Unsupported format");
    ++files_read_;
    parse_pageable(data, n, fmt, dev_sink, host_sink, fence);
  }

private:
  void parse_pageable(const char* data, size_t n, unsigned fmt, const dev_sink_type& dev_sink, const host_sink_type& host_sink,
                      const fence_type& fence) {
    size_t a = 0;
    bool first = true;
    while(a < n) {
      const size_t b = next_cut(data, a, n, fmt);
      if(b == npos && fmt == JFGPU_PARSE_FASTA) throw std::runtime_error("FASTA line longer than 1 GiB");
      int rc = JFGPU_E_FORMAT;
      const char* d_out = nullptr; size_t n_out = 0; uint64_t recs = 0;
      if(b != npos) {
        fence();
        rc = jfgpu_parser_parse(p_, data + a, b - a, fmt | (first ? 0u : JFGPU_PARSE_CONTINUE), &d_out, &n_out, &recs);
      }
      if(rc == JFGPU_E_FORMAT) { fallback_bytes_ += n - a; host_.parse_memory(data + a, n - a, host_sink); return; }
      if(rc) throw std::runtime_error(jfgpu_last_error());
      double ms = 0; jfgpu_parser_last_ms(p_, &ms); device_ms_ += ms;
      reads_read_ += recs;
      if(n_out) dev_sink(d_out, n_out);
      a = b; first = false;
    }
  }

  static bool read_exact(int fd, size_t off, char* dst, size_t len) {
    size_t done = 0;
    while(done < len) {
      const ssize_t r = pread(fd, dst + done, len - done, (off_t)(off + done));
      if(r <= 0) return false;
      done += (size_t)r;
    }
    return true;
  }
  // does a FASTQ record start at file offset `at` ('@' there, and the second-next line starts with '+')?
  static bool fastq_record_at(int fd, size_t size, size_t at) {
    std::vector<char> b(std::min<size_t>((size_t)1 << 20, size - at));
    if(b.empty() || !read_exact(fd, at, b.data(), b.size()) || b[0] != '@') return false;
    const char* e1 = (const char*)memchr(b.data(), '\n', b.size());
    const char* e2 = e1 ? (const char*)memchr(e1 + 1, '\n', b.data() + b.size() - (e1 + 1)) : nullptr;
    return e2 && e2 + 1 < b.data() + b.size() && e2[1] == '+';
  }

  static constexpr size_t npos = ~(size_t)0;
  int pinned_ = -1;             // JFGPU_FEED_PINNED: -1 auto (files >= 64 MiB), 0 never (pageable copy of the mapping), 1 always
  unsigned k_;
  size_t chunk_;
  jfgpu_parser* p_ = nullptr;
  char* pin_[2] = {nullptr, nullptr};

  // end of the chunk that starts at a (npos: none that the device parser could take)
  size_t next_cut(const char* d, size_t a, size_t n, unsigned fmt) const {
    size_t b = n;
    if(n - a > chunk_) b = fmt == JFGPU_PARSE_FASTA ? fasta_cut(d, a, n) : fastq_cut(d, a, n);
    if(b != npos && b - a > ((size_t)1 << 30)) b = npos;
    return b;
  }
  // file[off, off + len) -> dst, in slices read by a few threads (one pread stream copies out of the page
  // cache at ~10 GB/s, PCIe wants five times that; page faults on a shared mapping do not scale either)
  bool read_slices(int fd, size_t off, char* dst, size_t len) const {
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>(copy_threads_, len >> 22));
    std::vector<std::thread> th;
    std::vector<int> ok(nt, 1);
    const size_t per = (len + nt - 1) / nt;
    auto work = [&](unsigned i) {
      size_t o = (size_t)i * per; const size_t e = std::min(len, o + per);
      while(o < e) {
        const ssize_t r = pread(fd, dst + o, e - o, (off_t)(off + o));
        if(r <= 0) { ok[i] = 0; return; }
        o += (size_t)r;
      }
    };
    for(unsigned i = 1; i < nt; ++i) th.emplace_back(work, i);
    work(0);
    for(auto& t : th) t.join();
    for(int v : ok) if(!v) return false;
    return true;
  }

  // Regular files: two pinned buffers of chunk_ bytes, three overlapping stages.  Buffer w holds [carried tail of the
  // previous buffer][fresh bytes]; the part up to the last line / record boundary is uploaded on the parser's copy
  // stream (jfgpu_parser_upload) and parsed + counted on the device, while a background task moves the tail to the
  // other buffer and reads the next chunk with a few dozen parallel pread()s (one stream copies out of the page cache
  // at ~2-3 GB/s; a pageable copy of the mapping measured 2.9 GB/s end to end on a 10 GB file, profiles/r02_*).
  void parse_fd_pinned(int fd, size_t n, const dev_sink_type& dev_sink, const host_sink_type& host_sink, const fence_type& fence, size_t start = 0) {
    ++files_read_;
    // JFGPU_FEED_TRACE=1: where the wall time of the feed goes (stderr), for tools/cli_feed_bench.py
    const bool trace = getenv("JFGPU_FEED_TRACE") != nullptr;
    typedef std::chrono::steady_clock clk;
    double t_alloc = 0, t_fill0 = 0, t_fence = 0, t_parse = 0, t_sink = 0, t_wait_fill = 0, t_bg_fill = 0;
    auto since = [](clk::time_point a) { return std::chrono::duration<double>(clk::now() - a).count(); };
    auto ta = clk::now();
    const size_t cap = chunk_ + (chunk_ >> 2);
    for(int w = 0; w < 2; ++w)
      if(jfgpu_parser_host_buffer(p_, w, cap, &pin_[w])) throw std::runtime_error(jfgpu_last_error());
    t_alloc = since(ta);
    size_t pos = start;             // file offset of the first byte not yet read (the part read is [start, n))
    size_t have[2] = {0, 0};        // valid bytes in each buffer
    auto fill = [&](int w, size_t carried) -> bool {      // append fresh bytes behind `carried`
      const size_t want = std::min(n - pos, cap - carried > chunk_ ? chunk_ : cap - carried);
      if(want && !read_slices(fd, pos, pin_[w] + carried, want)) return false;
      pos += want; have[w] = carried + want;
      return true;
    };
    ta = clk::now();
    if(!fill(0, 0)) throw std::runtime_error("Error reading the sequence file");
    t_fill0 = since(ta);
    unsigned fmt;
    if(pin_[0][0] == '>') fmt = JFGPU_PARSE_FASTA;
    else if(pin_[0][0] == '@') fmt = JFGPU_PARSE_FASTQ;
    else throw std::runtime_error("Unsupported format");
    bool first = true;
    size_t consumed = start;        // file offset up to which a parser has taken the bytes (for the host fallback)
    for(int w = 0;; w ^= 1) {
      const char* buf = pin_[w];
      const size_t len = have[w];
      const bool last = pos >= n;
      size_t cut = len;
      if(!last) {
        if(fmt == JFGPU_PARSE_FASTA) { const void* q = memrchr(buf, '\n', len); cut = q ? (size_t)((const char*)q - buf) + 1 : npos; }
        else cut = fastq_cut_in(buf, len);
      }
      if(cut == npos || cut == 0) {           // no boundary in a whole buffer: the general reader takes the rest of the file
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if(m == MAP_FAILED) throw std::runtime_error("Can't mmap file");
        if(fmt == JFGPU_PARSE_FASTA && consumed > start) { munmap(m, n); throw std::runtime_error("FASTA line longer than the staging buffer"); }
        fallback_bytes_ += n - consumed;
        try { host_.parse_memory((const char*)m + consumed, n - consumed, host_sink); } catch(...) { munmap(m, n); throw; }
        munmap(m, n);
        return;
      }
      std::future<bool> next_ready;
      const size_t tail = len - cut;
      const bool more = !last || tail;
      if(jfgpu_parser_upload(p_, w, buf, cut)) throw std::runtime_error(jfgpu_last_error());   // asynchronous, copy stream
      if(more) {
        const int o = w ^ 1;
        next_ready = std::async(std::launch::async, [&, o, tail, cut, buf]() {
          const auto tb = clk::now();
          memcpy(pin_[o], buf + cut, tail);       // pin_[o]'s own upload finished before its parse returned (previous turn)
          const bool ok = fill(o, tail);
          t_bg_fill += since(tb);
          return ok;
        });
      }
      // (the upload of this buffer was enqueued before the background task started: see below)
      ta = clk::now();
      fence();
      t_fence += since(ta); ta = clk::now();
      const char* d_out = nullptr; size_t n_out = 0; uint64_t recs = 0;
      const int rc = jfgpu_parser_parse_uploaded(p_, w, fmt | (first ? 0u : JFGPU_PARSE_CONTINUE), &d_out, &n_out, &recs);
      t_parse += since(ta); ta = clk::now();
      if(next_ready.valid() && !next_ready.get()) throw std::runtime_error("Error reading the sequence file");
      t_wait_fill += since(ta);
      if(rc == JFGPU_E_FORMAT) {
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if(m == MAP_FAILED) throw std::runtime_error("Can't mmap file");
        fallback_bytes_ += n - consumed;
        try { host_.parse_memory((const char*)m + consumed, n - consumed, host_sink); } catch(...) { munmap(m, n); throw; }
        munmap(m, n);
        return;
      }
      if(rc) throw std::runtime_error(jfgpu_last_error());
      double ms = 0; jfgpu_parser_last_ms(p_, &ms); device_ms_ += ms;
      reads_read_ += recs;
      ta = clk::now();
      if(n_out) dev_sink(d_out, n_out);
      t_sink += since(ta);
      consumed += cut; first = false;
      if(!more) {                             // that was the last buffer and nothing is left over
        if(trace) fprintf(stderr, "[feed] %.2f GB: pinned buffers %.3f s, first fill %.3f, fence %.3f, upload+parse %.3f, wait for next fill %.3f "
                                  "(background fills %.3f, %u threads), sink %.3f\n", n / 1e9, t_alloc, t_fill0, t_fence, t_parse, t_wait_fill, t_bg_fill,
                          copy_threads_, t_sink);
        break;
      }
    }
  }

  // FASTQ record boundary inside a buffer: start of the last line that begins with '@' and whose
  // second-next line begins with '+' (see fastq_cut)
  static size_t fastq_cut_in(const char* d, size_t len) {
    size_t p = len;
    for(int tries = 0; tries < 256 && p > 0; ++tries) {
      const void* q = memrchr(d, '\n', p);
      if(!q) break;
      const size_t s = (const char*)q - d + 1;
      if(s < len && d[s] == '@') {
        const void* e1 = memchr(d + s, '\n', len - s);
        const void* e2 = e1 ? memchr((const char*)e1 + 1, '\n', len - ((const char*)e1 + 1 - d)) : nullptr;
        if(e2 && (size_t)((const char*)e2 + 1 - d) < len && ((const char*)e2)[1] == '+') return s;
      }
      p = (const char*)q - d;
    }
    return npos;
  }

  unsigned copy_threads_ = feed_threads();
  // pread streams per fill: a quarter of the hardware threads, at most 32 -- and no more than the CPU time the cgroup
  // grants (cpu.max): the copy out of the page cache costs 0.3-0.7 CPU-seconds per GB (tools/probes/feed_probe.hip), so on
  // a box limited to 16 CPUs the feed tops out near 20 GB/s whatever the thread count, and more threads only get throttled.
  // JFGPU_FEED_THREADS overrides.
  static unsigned feed_threads() {
    if(const char* e = getenv("JFGPU_FEED_THREADS")) return (unsigned)std::max(1, atoi(e));
    unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency() / 4));
    if(FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      long long quota = 0, period = 0;
      if(fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) nt = std::min<unsigned>(nt, (unsigned)std::max<long long>(1, quota / period));
      fclose(f);
    }
    return nt;
  }
  sequence_parser host_;
  size_t files_read_ = 0, reads_read_ = 0, fallback_bytes_ = 0;
  double device_ms_ = 0;

  // end of the chunk starting at a: just after the last '\n' within the next chunk_ bytes, else after
  // the first one beyond them
  size_t fasta_cut(const char* d, size_t a, size_t n) const {
    const void* q = memrchr(d + a, '\n', chunk_);
    if(q) return (const char*)q - d + 1;
    q = memchr(d + a + chunk_, '\n', n - a - chunk_);
    return q ? (size_t)((const char*)q - d + 1) : n;
  }
  // Start of the last record that begins within the next chunk_ bytes: a line starting with '@'
  // whose second-next line starts with '+'.  (A quality line may start with '@', but then the
  // second-next line is a sequence line.)
  size_t fastq_cut(const char* d, size_t a, size_t n) const {
    size_t p = a + chunk_;
    for(int tries = 0; tries < 256 && p > a; ++tries) {
      const void* q = memrchr(d + a, '\n', p - a);
      if(!q) break;
      const size_t s = (const char*)q - d + 1;
      if(s > a && s < n && d[s] == '@') {
        const void* e1 = memchr(d + s, '\n', n - s);
        const void* e2 = e1 ? memchr((const char*)e1 + 1, '\n', n - ((const char*)e1 + 1 - d)) : nullptr;
        if(e2 && (size_t)((const char*)e2 + 1 - d) < n && ((const char*)e2)[1] == '+') return s;
      }
      p = (const char*)q - d;
    }
    return npos;
  }
};

}  // namespace jellyfish_amd
