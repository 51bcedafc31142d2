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
#include <cstdlib>

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
    if(jfgpu_parser_create(device, mer_len, &p_)) throw std::runtime_error(jfgpu_last_error());
  }
  ~device_sequence_parser() { jfgpu_parser_destroy(p_); }
  device_sequence_parser(const device_sequence_parser&) = delete;
  device_sequence_parser& operator=(const device_sequence_parser&) = delete;

  size_t nb_files() const { return files_read_; }
  size_t nb_reads() const { return reads_read_ + host_.nb_reads(); }
  size_t host_fallback_bytes() const { return fallback_bytes_; }
  double device_ms() const { return device_ms_; }

  void parse_file(const char* path, const dev_sink_type& dev_sink, const host_sink_type& host_sink, const fence_type& fence) {
    int fd = open(path, O_RDONLY);
    if(fd < 0) throw std::runtime_error(std::string("Can't open file '") + path + "'");
    struct stat st;
    if(fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) {   // pipes, empty files: the host reader copes
      close(fd);
      host_.parse_file(path, host_sink);
      return;
    }
    void* m = mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if(m == MAP_FAILED) { host_.parse_file(path, host_sink); return; }
    madvise(m, st.st_size, MADV_SEQUENTIAL);
    try { parse_memory((const char*)m, st.st_size, dev_sink, host_sink, fence); } catch(...) { munmap(m, st.st_size); throw; }
    munmap(m, st.st_size);
  }

  void parse_memory(const char* data, size_t n, const dev_sink_type& dev_sink, const host_sink_type& host_sink, const fence_type& fence) {
    if(n == 0) { ++files_read_; return; }
    unsigned fmt;
    if(data[0] == '>') fmt = JFGPU_PARSE_FASTA;
    else if(data[0] == '@') fmt = JFGPU_PARSE_FASTQ;
    else throw std::runtime_error("Unsupported format");
    ++files_read_;
    size_t a = 0;
    bool first = true;
    while(a < n) {
      size_t b = n;
      if(n - a > chunk_) b = fmt == JFGPU_PARSE_FASTA ? fasta_cut(data, a, n) : fastq_cut(data, a, n);
      if(b == npos || b - a > ((size_t)1 << 31)) {
        if(fmt == JFGPU_PARSE_FASTA) throw std::runtime_error("FASTA line longer than 2 GiB");
        b = npos;
      }
      int rc = JFGPU_E_FORMAT;
      const char* d_out = nullptr; size_t n_out = 0; uint64_t recs = 0;
      if(b != npos) {
        fence();
        rc = jfgpu_parser_parse(p_, data + a, b - a, fmt | (first ? 0u : JFGPU_PARSE_CONTINUE), &d_out, &n_out, &recs);
      }
      if(rc == JFGPU_E_FORMAT) {             // FASTQ outside the strict layout: the general reader takes the rest of the file
        fallback_bytes_ += n - a;
        host_.parse_memory(data + a, n - a, host_sink);
        return;
      }
      if(rc) throw std::runtime_error(jfgpu_last_error());
      double ms = 0; jfgpu_parser_last_ms(p_, &ms); device_ms_ += ms;
      reads_read_ += recs;
      if(n_out) dev_sink(d_out, n_out);
      a = b; first = false;
    }
  }

private:
  static constexpr size_t npos = ~(size_t)0;
  unsigned k_;
  size_t chunk_;
  jfgpu_parser* p_ = nullptr;
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
