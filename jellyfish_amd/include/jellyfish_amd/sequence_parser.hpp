// jellyfish_amd/include/jellyfish_amd/sequence_parser.hpp
//
// Host-side feed: FASTA / FASTQ file -> "contract buffers" for the device
// (jfgpu_count_ascii).  Observable behaviour follows the reference's
// mer_overlap_sequence_parser (include/jellyfish/mer_overlap_sequence_parser.hpp):
//   * format sniffed from the first byte, '>' FASTA / '@' FASTQ, anything else is
//     "Unsupported format" (:134-148); empty files are skipped (:135)
//   * header lines dropped, sequence lines of a record concatenated with '\n' and
//     trailing '\r' removed (:260-274), one 'N' written between records (:173-176)
//   * FASTQ: sequence lines up to a line starting with '+', then exactly seq_len
//     quality characters are skipped, the next record must start with '@'
//     (:187-217, :290-307), else "Invalid fastq sequence"
//   * consecutive buffers of one file overlap by k-1 characters (the "seam",
//     :164-167,182-184) so no k-mer is lost or seen twice; nothing is carried
//     across files (:111).
// Unlike the reference (4 KiB buffers handed to CPU threads through a lock-free
// pool) the buffers here are tens of MiB: one buffer = one kernel launch.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

namespace jellyfish_amd {

class sequence_parser {
public:
  // sink(buffer, length): one contract buffer; called from the parsing thread.
  typedef std::function<void(const char*, size_t)> sink_type;

  sequence_parser(unsigned mer_len, size_t buf_size = (size_t)32 << 20)
      : k_(mer_len), buf_size_(std::max<size_t>(buf_size, 4 * (size_t)mer_len + 64)) { buf_.reserve(buf_size_ + 4096); }

  size_t nb_files() const { return files_read_; }
  size_t nb_reads() const { return reads_read_; }
  // count -Q / --min-quality (mer_qual_iterator.hpp:75-84): FASTQ bases with a lower quality character become 'N'; 0 = off
  void min_quality(int c) { min_qual_ = c; }

  void parse_file(const char* path, const sink_type& sink) {
    int fd = open(path, O_RDONLY);
    if(fd < 0) throw std::runtime_error(std::string("Can't open file '") + path + "'");
    struct stat st;
    if(fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
      void* p = mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if(p != MAP_FAILED) {
        madvise(p, st.st_size, MADV_SEQUENTIAL);
        try { parse_memory((const char*)p, st.st_size, sink); } catch(...) { munmap(p, st.st_size); close(fd); throw; }
        munmap(p, st.st_size);
        close(fd);
        return;
      }
    }
    // pipes / stdin / zero-size special files: slurp
    std::string all;
    char tmp[1 << 16];
    ssize_t n;
    while((n = read(fd, tmp, sizeof tmp)) > 0) all.append(tmp, n);
    close(fd);
    parse_memory(all.data(), all.size(), sink);
  }

  void parse_memory(const char* data, size_t n, const sink_type& sink) {
    ++files_read_;
    buf_.clear();
    if(n == 0) return;
    const char* p = data;
    const char* end = data + n;
    if(*p == '>') parse_fasta(p, end, sink);
    else if(*p == '@') parse_fastq(p, end, sink);
    else throw std::runtime_error("Unsupported format");
    flush(sink, true);
  }

private:
  unsigned k_;
  size_t buf_size_;
  std::string buf_;
  size_t files_read_ = 0, reads_read_ = 0;
  int min_qual_ = 0;

  static const char* line_end(const char* p, const char* end) {
    const char* nl = (const char*)memchr(p, '\n', end - p);
    return nl ? nl : end;
  }
  static const char* skip_newlines(const char* p, const char* end) {
    while(p < end && (*p == '\n' || *p == '\r')) ++p;
    return p;
  }
  void append_line(const char* p, const char* e) {
    while(e > p && e[-1] == '\r') --e;
    buf_.append(p, e - p);
  }
  void flush(const sink_type& sink, bool final) {
    if(buf_.empty()) return;
    sink(buf_.data(), buf_.size());
    if(final || buf_.size() < k_ - 1) { buf_.clear(); return; }
    std::string seam = buf_.substr(buf_.size() - (k_ - 1));   // next buffer starts with the last k-1 chars
    buf_.swap(seam);
  }
  void maybe_flush(const sink_type& sink) { if(buf_.size() >= buf_size_) flush(sink, false); }

  void parse_fasta(const char* p, const char* end, const sink_type& sink) {
    p = line_end(p, end); if(p < end) ++p;   // first header
    ++reads_read_;
    bool any = false;                        // something written for this file so far
    while(p < end) {
      p = skip_newlines(p, end);
      if(p >= end) break;
      if(*p == '>') {
        if(any) buf_.push_back('N');
        p = line_end(p, end); if(p < end) ++p;
        ++reads_read_;
        continue;
      }
      const char* e = line_end(p, end);
      append_line(p, e);
      any = any || e > p;
      p = e < end ? e + 1 : end;
      maybe_flush(sink);
    }
  }

  void parse_fastq(const char* p, const char* end, const sink_type& sink) {
    p = line_end(p, end); if(p < end) ++p;   // first '@' header
    ++reads_read_;
    while(p < end) {
      // sequence lines until a line starting with '+'
      size_t seq_len = 0;
      const size_t rec_start = buf_.size();                  // this record's bases are buf_[rec_start, rec_start + seq_len)
      while(true) {
        p = skip_newlines(p, end);
        if(p >= end || *p == '+') break;
        const char* e = line_end(p, end);
        const size_t before = buf_.size();
        append_line(p, e);
        seq_len += buf_.size() - before;
        p = e < end ? e + 1 : end;
      }
      if(p >= end) break;
      // '+' line, then exactly seq_len quality characters (line breaks tolerated)
      p = line_end(p, end); if(p < end) ++p;
      size_t quals = 0;
      while(p < end && quals < seq_len) {
        p = skip_newlines(p, end);
        const char* e = line_end(p, end);
        const char* le = e;
        while(le > p && le[-1] == '\r') --le;
        if(min_qual_)
          for(const char* q = p; q < le && quals + (size_t)(q - p) < seq_len; ++q)
            if(*q < min_qual_) buf_[rec_start + quals + (size_t)(q - p)] = 'N';
        quals += le - p;
        p = e < end ? e + 1 : end;
      }
      p = skip_newlines(p, end);
      if(quals != seq_len || (p < end && *p != '@')) throw std::runtime_error("Invalid fastq sequence");
      if(p < end) {
        buf_.push_back('N');
        p = line_end(p, end); if(p < end) ++p;
        ++reads_read_;
      }
      maybe_flush(sink);
    }
  }
};

}  // namespace jellyfish_amd
