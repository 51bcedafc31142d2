// jellyfish_amd/include/jellyfish_amd/dumpers.hpp
//
// Results path on the host, API-shaped like the reference's dumpers and readers:
//   binary_dumper / text_dumper   include/jellyfish/binary_dumper.hpp:46-76,
//                                 text_dumper.hpp, sorted_dumper.hpp:57-101, dumper.hpp:68-91
//   binary_reader / text_reader   binary_dumper.hpp:82-109, text_dumper.hpp
//   binary_query                  binary_dumper.hpp:112-213 (interpolation search on pos)
// The (pos, key) sort the reference does with a per-thread min-heap happens on
// the GPU, tile by tile (jfgpu_dump_next); the host only streams bytes to the file.
#pragma once
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cmath>
#include <fstream>
#include <future>
#include <thread>
#include <iostream>
#include <limits>
#include <sstream>

#include "hash_counter.hpp"

namespace jellyfish_amd {

struct ErrorWriting : public std::runtime_error {   // dumper_t::ErrorWriting, dumper.hpp:37
  explicit ErrorWriting(const std::string& s) : std::runtime_error(s) {}
};

class dumper_base {
public:
  dumper_base(const char* file_prefix, file_header* header)
      : prefix_(file_prefix), header_(header), min_(0), max_(std::numeric_limits<uint64_t>::max()) {}
  virtual ~dumper_base() {}
  void one_file(bool v) { one_file_ = v; }
  void min(uint64_t m) { min_ = m; }
  void max(uint64_t m) { max_ = m; }
  int nb_files() const { return nb_files_; }
  const std::vector<std::string>& file_names() const { return file_names_; }    // dumper_t::file_names, dumper.hpp:80-87
  virtual void dump(hash_counter* ary) = 0;

protected:
  std::string prefix_;
  file_header* header_;
  uint64_t min_, max_;
  bool one_file_ = true;
  int nb_files_ = 0;
  std::vector<std::string> file_names_;
  std::string next_path() {   // dumper.hpp:45-61: prefix itself when one_file, else prefix + index
    std::string p = prefix_;
    if(!one_file_) p += std::to_string(nb_files_);
    ++nb_files_;
    file_names_.push_back(p);
    return p;
  }
};

// binary/sorted: header, then fixed-width records key (ceil(2k/8) bytes LE) + count
// (val_len bytes LE, saturated) in ascending (pos, key) order.
class binary_dumper : public dumper_base {
public:
  static constexpr const char* format = "binary/sorted";
  binary_dumper(int val_len /* bytes */, int key_len /* bits */, int nb_threads, const char* file_prefix, file_header* header = 0)
      : dumper_base(file_prefix, header), val_len_(val_len), key_len_(key_len) { (void)nb_threads; }

  void dump(hash_counter* ary) override {
    if((int)ary->info().out_counter_len != val_len_)
      throw std::length_error("binary_dumper: table was created with a different out_counter_len");
    ary->flush();                                              // pending adds first: they may still double the table
    const std::string path = next_path();
    // One shard of a multi-GPU table (hash_counter::attach_comm): the shards' sorted dumps concatenated in rank order are
    // the globally (pos, key)-sorted body, so every rank writes its records into the same file at the offset the record
    // counts of the ranks before it give; rank 0 writes the header and sizes the file first.
    jfgpu_comm* comm = ary->comm();
    int world = 1, rank = 0;
    if(comm) jf_check(jfgpu_comm_world(comm, &world, &rank));
    uint64_t n = 0; uint32_t rec = 0;
    jf_check(jfgpu_dump_begin(ary->handle(), min_, max_, &n, &rec));
    uint64_t first_record = 0, all_records = n;
    if(comm) {
      std::vector<uint64_t> counts(world);
      if(jfgpu_comm_allgather_u64(comm, n, counts.data())) { jfgpu_dump_end(ary->handle()); throw std::runtime_error(jfgpu_last_error()); }
      all_records = 0;
      for(int r = 0; r < world; ++r) { if(r < rank) first_record += counts[r]; all_records += counts[r]; }
    }
    off_t body = 0;
    if(rank == 0) {
      std::ofstream out(path, std::ios::binary | std::ios::trunc);
      if(!out.good()) { jfgpu_dump_end(ary->handle()); throw ErrorWriting("Can't open file '" + path + "'"); }
      if(header_) {
        ary->update_header(*header_);
        header_->format(format);
        header_->counter_len(val_len_);
        header_->write(out);
      }
      out.flush();
      if(!out.good()) { jfgpu_dump_end(ary->handle()); throw ErrorWriting("Error while writing '" + path + "'"); }
      body = (off_t)out.tellp();
      out.close();
      // the body's size is known (fixed-width records, counted by jfgpu_dump_begin): the file is sized now, every rank and every
      // writer thread then fills its own part of it
      if(::truncate(path.c_str(), body + (off_t)(all_records * rec)) != 0) {
        // ranks write at offsets of a file sized here: without it they cannot; a single writer only loses the pre-size
        if(comm) { jfgpu_dump_end(ary->handle()); throw ErrorWriting("Can't size '" + path + "'"); }
        static bool said = false;
        if(!said) { said = true; fprintf(stderr, "jellyfish-amd: could not pre-size '%s' (%s): writing without\n", path.c_str(), strerror(errno)); }
      }
    }
    if(comm) {                                                 // where the body starts; also: the file exists from here on
      std::vector<uint64_t> bodies(world);
      if(jfgpu_comm_allgather_u64(comm, (uint64_t)body, bodies.data())) { jfgpu_dump_end(ary->handle()); throw std::runtime_error(jfgpu_last_error()); }
      body = (off_t)bodies[0] + (off_t)(first_record * rec);
    }
    // Records have a fixed width and arrive in file order, so every chunk's place in the file is known: while
    // the device sorts and ships chunk i+1 into one buffer, a few threads pwrite() the slices of chunk i from
    // the other (a single write() stream copies into the page cache at 2-3 GB/s; the device delivers faster).
    const uint64_t cap = std::max<uint64_t>(std::min<uint64_t>((uint64_t)16 << 20, std::max<uint64_t>(n, (uint64_t)1 << 16)), ary->info().tile_slots);
    struct pinned {                                            // device -> host at PCIe speed needs pinned memory
      char* p = nullptr; std::vector<char> fallback;
      explicit pinned(size_t n) { void* q = nullptr; if(jfgpu_malloc_host(n, &q) == JFGPU_OK) p = (char*)q; else { fallback.resize(n); p = fallback.data(); } }
      ~pinned() { if(fallback.empty()) jfgpu_free_host(p); }
      char* data() { return p; }
    } bufs[2] = {pinned(cap * rec), pinned(cap * rec)};
    const int fd = ::open(path.c_str(), O_WRONLY);
    if(fd < 0) { jfgpu_dump_end(ary->handle()); throw ErrorWriting("Can't reopen '" + path + "' for writing"); }
    // Where the Writing phase goes (profiles/r04_cli_writing.log, JFGPU_DUMP_TRACE=1): the device delivers sorted records at
    // 21 GB/s (sort + copy to pinned memory); ONE file on tmpfs takes 4.5 - 6.8 GB/s whatever is tried from user space --
    // write() / pwrite() hold the file's inode lock for the whole call (one thread 6.8 GB/s, four or sixteen 4.0 - 4.4
    // between them), stores into a shared mapping fault page by page under per-inode accounting (fresh pages 3.7 - 5.9,
    // pages made by fallocate() ahead of the stores no better inside this process); tools/probes/tmpfs_write_probe.cc.
    // So: two pwrite() streams (one would idle while its chunk is replaced), which is what the file system gives.
    std::vector<std::future<bool>> pending[2];
    auto drain = [&](int b) { bool ok = true; for(auto& f : pending[b]) ok = f.get() && ok; pending[b].clear(); return ok; };
    unsigned nw = 2;
    if(const char* e = getenv("JFGPU_DUMP_WRITERS")) nw = (unsigned)std::max(1, atoi(e));
    uint64_t written = 0;
    bool ok = true;
    // JFGPU_DUMP_TRACE=1: where the Writing phase goes (waiting for the file writers / for the device's next chunk)
    const bool trace = getenv("JFGPU_DUMP_TRACE") != nullptr;
    double t_drain = 0, t_next = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    try {
      for(int b = 0;; b ^= 1) {
        const double a0 = now();
        ok = drain(b) && ok;                                   // this buffer's previous slices are on their way to disk
        const double a1 = now();
        uint64_t got = 0;
        jf_check(jfgpu_dump_next(ary->handle(), bufs[b].data(), cap, &got));
        t_drain += a1 - a0; t_next += now() - a1;
        if(!got) break;
        const size_t bytes = got * rec, per = (bytes / nw + 4095) / 4096 * 4096 + 4096;
        for(size_t o = 0; o < bytes; o += per) {
          const char* src = bufs[b].data() + o;
          const size_t len = std::min(per, bytes - o);
          const off_t at = body + (off_t)(written * rec + o);
          pending[b].push_back(std::async(std::launch::async, [fd, src, len, at]() {
            size_t done = 0;
            while(done < len) {
              const ssize_t w = ::pwrite(fd, src + done, len - done, at + (off_t)done);
              if(w <= 0) return false;
              done += (size_t)w;
            }
            return true;
          }));
        }
        written += got;
      }
      const double a0 = now();
      ok = drain(0) && ok; ok = drain(1) && ok;
      t_drain += now() - a0;
      if(trace) std::cerr << "[dump] " << written << " records of " << rec << " bytes: " << t_next << " s in jfgpu_dump_next (device sort + copy), " << t_drain
                          << " s waiting for " << nw << " file writers\n";
    } catch(...) { drain(0); drain(1); ::close(fd); jfgpu_dump_end(ary->handle()); throw; }
    ::close(fd);
    jf_check(jfgpu_dump_end(ary->handle()));
    if(!ok) throw ErrorWriting("Error while writing '" + path + "'");
  }

private:
  int val_len_, key_len_;
};

inline void write_text_records(hash_counter* ary, uint64_t min, uint64_t max, std::ostream& out) {
  uint64_t n = 0; uint32_t rec = 0;
  jf_check(jfgpu_dump_begin(ary->handle(), min, max, &n, &rec));
  const uint64_t cap = std::max<uint64_t>((uint64_t)1 << 20, ary->info().tile_slots);
  std::vector<unsigned char> buf(cap * rec);
  const unsigned k = ary->info().k, kb = (2 * k + 7) / 8, vb = rec - kb;
  mer_dna m(k);
  std::string line;
  try {
    while(true) {
      uint64_t got = 0;
      jf_check(jfgpu_dump_next(ary->handle(), buf.data(), cap, &got));
      if(!got) break;
      line.clear();
      for(uint64_t i = 0; i < got; ++i) {
        const unsigned char* r = &buf[i * rec];
        uint64_t val = 0;
        memset(m.data__(), 0, m.nb_words() * sizeof(uint64_t));
        memcpy(m.data__(), r, kb); memcpy(&val, r + kb, vb);
        line += m.to_str(); line += ' '; line += std::to_string(val); line += '\n';
      }
      out.write(line.data(), line.size());
    }
  } catch(...) { jfgpu_dump_end(ary->handle()); throw; }
  jf_check(jfgpu_dump_end(ary->handle()));
}

// text/sorted: same header, then "KMER count\n" lines (text_dumper.hpp:18-20); counts are not
// saturated in text, so the table must be created with out_counter_len = 8 for exactness.
class text_dumper : public dumper_base {
public:
  static constexpr const char* format = "text/sorted";
  text_dumper(int nb_threads, const char* file_prefix, file_header* header = 0) : dumper_base(file_prefix, header) { (void)nb_threads; }
  void dump(hash_counter* ary) override {
    ary->flush();
    const std::string path = next_path();
    // One shard of a multi-GPU table (hash_counter::attach_comm): the shards' sorted lines concatenated in rank order are the
    // globally (pos, key)-sorted body, but a text record has no fixed width -- every rank writes its lines to a part file
    // beside the output, rank 0 (which wrote the header and its own lines into the output itself) appends the parts in rank
    // order once everybody is through, and removes them.
    jfgpu_comm* comm = ary->comm();
    int world = 1, rank = 0;
    if(comm) jf_check(jfgpu_comm_world(comm, &world, &rank));
    auto part_name = [&](int r) { return path + ".rank" + std::to_string(r); };
    {
      std::ofstream out(rank == 0 ? path : part_name(rank), std::ios::binary | std::ios::trunc);
      if(!out.good()) throw ErrorWriting("Can't open file '" + (rank == 0 ? path : part_name(rank)) + "'");
      if(header_ && rank == 0) {
        ary->update_header(*header_);
        header_->format(format);
        header_->write(out);
      }
      write_text_records(ary, min_, max_, out);
      out.flush();
      if(!out.good()) throw ErrorWriting("Error while writing '" + path + "'");
    }
    if(comm && world > 1) {
      std::vector<uint64_t> done(world);
      if(jfgpu_comm_allgather_u64(comm, 1, done.data())) throw std::runtime_error(jfgpu_last_error());      // (everybody's part is on disk)
      if(rank == 0) {
        std::ofstream out(path, std::ios::binary | std::ios::app);
        for(int r = 1; r < world; ++r) {
          std::ifstream in(part_name(r), std::ios::binary);
          if(!in.good()) throw ErrorWriting("Can't read the part '" + part_name(r) + "'");
          if(in.peek() != std::ifstream::traits_type::eof()) out << in.rdbuf();
          in.close();
          ::unlink(part_name(r).c_str());
        }
        out.flush();
        if(!out.good()) throw ErrorWriting("Error while writing '" + path + "'");
      }
      if(jfgpu_comm_allgather_u64(comm, 1, done.data())) throw std::runtime_error(jfgpu_last_error());      // (the file is whole when any rank returns)
    }
  }
};

// Sequential reader of binary/sorted (binary_dumper.hpp:82-109).
class binary_reader {
public:
  binary_reader(std::istream& is, file_header* header)
      : is_(is), val_len_(header->counter_len()), key_(header->key_len() / 2), m_(header->matrix()),
        size_mask_(header->size() - 1) {}
  const mer_dna& key() const { return key_; }
  const uint64_t& val() const { return val_; }
  size_t pos() const { return m_.times(key_.data()) & size_mask_; }
  bool next() {
    key_.read_bytes(is_);
    val_ = 0;
    is_.read((char*)&val_, val_len_);
    return is_.good();
  }
private:
  std::istream& is_;
  const int val_len_;
  mer_dna key_;
  uint64_t val_ = 0;
  const header_matrix m_;
  const size_t size_mask_;
};

class text_reader {
public:
  text_reader(std::istream& is, file_header* header) : is_(is), key_(header->key_len() / 2) {}
  const mer_dna& key() const { return key_; }
  const uint64_t& val() const { return val_; }
  bool next() {
    std::string s;
    is_ >> s >> val_;
    if(!is_.good() && s.empty()) return false;
    try { key_ = s; } catch(std::length_error&) { return false; }
    return !is_.fail();
  }
private:
  std::istream& is_;
  mer_dna key_;
  uint64_t val_ = 0;
};

// Read-only memory map of a whole file (include/jellyfish/mapped_file.hpp).
class mapped_file {
public:
  explicit mapped_file(const char* path) {
    int fd = open(path, O_RDONLY);
    if(fd < 0) throw std::runtime_error(std::string("Can't open file '") + path + "'");
    struct stat st;
    if(fstat(fd, &st) < 0) { close(fd); throw std::runtime_error("Can't stat file"); }
    length_ = st.st_size;
    base_ = length_ ? (char*)mmap(nullptr, length_, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    close(fd);
    if(base_ == (char*)MAP_FAILED) throw std::runtime_error("Can't mmap file");
  }
  ~mapped_file() { if(base_) munmap(base_, length_); }
  char* base() const { return base_; }
  size_t length() const { return length_; }
private:
  char* base_ = nullptr;
  size_t length_ = 0;
};

// Random access into a binary/sorted body (binary_dumper.hpp:112-213): interpolation
// search on pos = matrix * key & mask, then a short linear scan.
class binary_query {
public:
  binary_query(const char* data, unsigned key_len /* bits */, unsigned val_len /* bytes */, const header_matrix& m,
               size_t mask, size_t size)
      : data_(data), val_len_(val_len), key_len_(key_len / 8 + (key_len % 8 != 0)), m_(m), mask_(mask),
        record_len_(val_len + key_len_), last_id_(size / record_len_), k_(key_len / 2) {
    if(size % record_len_ != 0)
      throw std::length_error("Size of database (" + std::to_string(size) + ") must be a multiple of the length of a record (" +
                              std::to_string(record_len_) + ")");
    if(last_id_) {
      first_key_ = key_at(0); first_pos_ = key_pos(first_key_);
      last_key_ = key_at(last_id_ - 1); last_pos_ = key_pos(last_key_);
    }
  }

  // Where is `key` in the body?  Records ascend by (pos, key) and the positions of hashed keys are uniform, so the record
  // number is guessed from the position (one interpolation over the whole file), the guess is widened by doubling steps
  // until it brackets the key, and the bracket is bisected.  (Same file order as the reference's binary_query relies on,
  // binary_dumper.hpp:147-189; its search re-interpolates inside a shrinking window and ends in a linear scan.)
  bool val_id(const mer_dna& key, uint64_t* res, uint64_t* id) const {
    if(last_id_ == 0) return false;
    const uint64_t pos = key_pos(key);
    if(pos < first_pos_ || pos > last_pos_) return false;
    // -1 / 0 / +1: record i sorts before / is / sorts after the key
    auto side = [&](uint64_t i) -> int {
      const mer_dna m = key_at(i);
      if(m == key) return 0;
      const uint64_t p = key_pos(m);
      return (p < pos || (p == pos && m < key)) ? -1 : 1;
    };
    auto found = [&](uint64_t i) { *res = val_at(i); *id = i; return true; };
    const uint64_t n = last_id_;
    uint64_t guess = last_pos_ > first_pos_ ? (uint64_t)((long double)(n - 1) * (long double)(pos - first_pos_) / (long double)(last_pos_ - first_pos_)) : 0;
    if(guess >= n) guess = n - 1;
    int s = side(guess);
    if(s == 0) return found(guess);
    uint64_t lo, hi;                                          // the key, if present, is a record of the open interval (lo, hi) -- lo may be "before 0", hi == n "after the last"
    bool lo_open = false;
    if(s < 0) {                                               // the guess is before the key: gallop upwards
      lo = guess; hi = n;
      for(uint64_t step = 1; lo + step < n; step <<= 1) {
        const int t = side(lo + step);
        if(t == 0) return found(lo + step);
        if(t > 0) { hi = lo + step; break; }
        lo += step;
      }
    } else {                                                  // after the key: gallop downwards
      hi = guess; lo = 0; lo_open = true;
      for(uint64_t step = 1; hi >= step; step <<= 1) {
        const int t = side(hi - step);
        if(t == 0) return found(hi - step);
        if(t < 0) { lo = hi - step; lo_open = false; break; }
        hi -= step;
      }
      if(lo_open) {                                           // ran off the front: record 0 is still unseen when hi > 0
        if(hi == 0) return false;
        const int t = side(0);
        if(t == 0) return found(0);
        if(t > 0) return false;
        lo = 0;
      }
    }
    while(hi - lo > 1) {
      const uint64_t mid = lo + (hi - lo) / 2;
      const int t = side(mid);
      if(t == 0) return found(mid);
      if(t < 0) lo = mid; else hi = mid;
    }
    return false;
  }
  uint64_t operator[](const mer_dna& key) const { uint64_t r, id; return val_id(key, &r, &id) ? r : 0; }
  uint64_t check(const mer_dna& key) const { return (*this)[key]; }

private:
  const char* data_;
  unsigned val_len_, key_len_;
  header_matrix m_;
  size_t mask_, record_len_, last_id_;
  unsigned k_;
  mer_dna first_key_{1u}, last_key_{1u};
  uint64_t first_pos_ = 0, last_pos_ = 0;

  mer_dna key_at(size_t id) const {
    mer_dna k(k_);
    memcpy(k.data__(), data_ + id * record_len_, key_len_);
    k.clean_msw();
    return k;
  }
  uint64_t val_at(size_t id) const {
    uint64_t v = 0;
    memcpy(&v, data_ + id * record_len_ + key_len_, val_len_);
    return v;
  }
  uint64_t key_pos(const mer_dna& key) const { return m_.times(key.data()) & mask_; }
};

// Read-only view of a bloomcounter file body (bloom_counter2::check__, bloom_counter2.hpp:109-142, with the
// hash pair of mer_dna_bloom_counter.hpp:19-34): cell_i = (h0 % m + i * (h1 % m)) % m, base-3 digit (p % 5) of
// byte (p / 5), result = the minimum digit over the nb_hashes cells (0, 1 or 2).
class bloom_query {
public:
  bloom_query(const char* data, size_t nbytes, uint64_t m, unsigned nb_hashes, const header_matrix& m1, const header_matrix& m2)
      : data_((const unsigned char*)data), m_(m), k_(nb_hashes), m1_(m1), m2_(m2) {
    if(nbytes < m / 5 + (m % 5 != 0)) throw std::length_error("Bloom counter file is truncated");
  }
  unsigned check(const mer_dna& key) const {
    static const unsigned pow3[5] = {1, 3, 9, 27, 81};
    const uint64_t base = m1_.times(key.data()) % m_, inc = m2_.times(key.data()) % m_;
    uint64_t p = base;
    unsigned res = 2;
    for(unsigned i = 0; i < k_; ++i) {
      const unsigned d = (data_[p / 5] / pow3[p % 5]) % 3;
      res = std::min(res, d);
      p += inc; if(p >= m_) p -= m_;
    }
    return res;
  }
private:
  const unsigned char* data_;
  uint64_t m_;
  unsigned k_;
  header_matrix m1_, m2_;
};

}  // namespace jellyfish_amd
