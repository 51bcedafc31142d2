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
#include <cstdlib>
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
    // pipes / stdin / special files: in pieces of whole records
    try { parse_stream(fd, sink); } catch(...) { close(fd); throw; }
    close(fd);
  }

  // A stream that cannot be mapped (a pipe, e.g. `zcat reads.fa.gz |`, generator commands): read about `piece` bytes at a
  // time, hand the records that are complete to parse_memory, carry the rest.  Memory is bounded by the piece or by the
  // longest record, not by the stream.  (Pieces end where a record starts: a line beginning with '>'; for FASTQ a line
  // beginning with '@' whose second-next line begins with '+'.)
  void parse_stream(int fd, const sink_type& sink, size_t piece = (size_t)64 << 20) {
    if(const char* e = getenv("JFGPU_STREAM_PIECE")) piece = std::max<size_t>(1, strtoull(e, nullptr, 10));   // (tests: many small pieces)
    std::string data;
    data.reserve(piece + (piece >> 2));
    const size_t files_before = files_read_;
    bool eof = false, any = false;
    while(!eof) {
      const size_t goal = data.size() + piece;
      while(data.size() < goal) {
        const size_t old = data.size();
        const size_t want = std::min<size_t>((size_t)1 << 20, goal - old);
        data.resize(old + want);
        const ssize_t r = read(fd, &data[old], want);
        if(r < 0) { data.resize(old); throw std::runtime_error("Error reading the sequence stream"); }
        data.resize(old + (size_t)r);
        if(r == 0) { eof = true; break; }
      }
      if(data.empty()) break;
      if(data[0] != '>' && data[0] != '@') throw std::runtime_error("Unsupported format");
      size_t cut = data.size();
      if(!eof) {
        cut = data[0] == '>' ? last_fasta_record(data) : last_fastq_record(data);
        if(cut == 0) continue;                      // one record longer than everything read so far: read on
      }
      parse_memory(data.data(), cut, sink);
      any = true;
      data.erase(0, cut);
    }
    (void)any;
    files_read_ = files_before + 1;                 // one stream = one file, however many pieces
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

  // start of the last record that begins inside d (0: none but the first)
  static size_t last_fasta_record(const std::string& d) {
    size_t end = d.size();
    while(end > 0) {
      const void* q = memrchr(d.data(), '\n', end);           // the last newline before `end`
      if(!q) return 0;
      const size_t nl = (size_t)((const char*)q - d.data());
      if(nl + 1 < d.size() && d[nl + 1] == '>') return nl + 1;
      end = nl;
    }
    return 0;
  }
  static size_t last_fastq_record(const std::string& d) {
    size_t end = d.size();
    while(end > 0) {
      const void* q = memrchr(d.data(), '\n', end);
      if(!q) return 0;
      const size_t nl = (size_t)((const char*)q - d.data()), s = nl + 1;
      if(s < d.size() && d[s] == '@') {                        // a header, unless it is a quality line: then the
        const char* e1 = (const char*)memchr(d.data() + s, '\n', d.size() - s);                       // second-next line
        const char* e2 = e1 ? (const char*)memchr(e1 + 1, '\n', d.data() + d.size() - (e1 + 1)) : nullptr;   // is a sequence,
        if(e2 && e2 + 1 < d.data() + d.size() && e2[1] == '+') return s;                               // not the '+' line
      }
      end = nl;
    }
    return 0;
  }
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
