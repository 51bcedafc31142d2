// jellyfish_amd/include/jellyfish_amd/hash_counter.hpp
//
// C++ facade over the C ABI (include/jfgpu.h) with the API shape of the
// reference's jellyfish::cooperative::hash_counter<mer_dna> and its array
// (include/jellyfish/hash_counter.hpp:50-172, large_hash_array.hpp:196-226,
// 354-372): the table lives in one GPU's HBM, add() calls are batched on the
// host and retired by kernels, done() == jfgpu_sync.  Errors cross the C ABI as
// codes and are re-thrown here as the exception types the reference throws
// (std::runtime_error("Hash full"), hash_counter.hpp:194-195; allocation failure,
// large_hash_array.hpp:169-172).
#pragma once
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/jfgpu.h"
#include "file_header.hpp"
#include "mer_dna.hpp"

namespace jellyfish_amd {

struct ErrorAllocation : public std::runtime_error {
  explicit ErrorAllocation(const std::string& s) : std::runtime_error(s) {}
};

inline void jf_check(int rc) {
  if(rc == JFGPU_OK) return;
  const std::string msg = jfgpu_last_error();
  switch(rc) {
  case JFGPU_E_ALLOC: throw ErrorAllocation(msg);
  case JFGPU_E_INVALID: throw std::length_error(msg);
  default: throw std::runtime_error(msg);
  }
}

class hash_counter {
public:
  // hash_counter(size, key_len (bits), val_len (bits), nb_threads, reprobe_limit)
  // (hash_counter.hpp:56-64).  val_len / nb_threads / reprobe_limit are CPU-table
  // knobs: accepted for source compatibility, the device slot format fixes them.
  // shard_bits / shard_id: this object is one shard of a table spread over 2^shard_bits GPUs (size stays the GLOBAL size);
  // see attach_comm.
  hash_counter(size_t size, uint16_t key_len, uint16_t val_len, uint16_t nb_threads, uint16_t reprobe_limit = 126,
               bool canonical = false, int device = -1, uint32_t out_counter_len = 4, uint64_t matrix_seed = 0,
               uint32_t shard_bits = 0, uint32_t shard_id = 0, uint32_t matrix_kind = 0 /* jfgpu.h: JFGPU_MATRIX_* */)
      : nb_threads_(nb_threads) {
    (void)val_len; (void)reprobe_limit;
    if(key_len == 0 || key_len % 2) throw std::length_error("key_len must be an even number of bits");
    jfgpu_params p;
    memset(&p, 0, sizeof p);
    p.k = key_len / 2; p.canonical = canonical; p.size = size; p.device = device;
    p.matrix_seed = matrix_seed; p.out_counter_len = out_counter_len;
    p.shard_bits = shard_bits; p.shard_id = shard_id; p.matrix_kind = matrix_kind;
    jf_check(jfgpu_create(&p, &t_));
    jf_check(jfgpu_get_info(t_, &info_));
    kw_ = (key_len + 63) / 64;
    pending_.reserve(kBatch * kw_);
  }
  ~hash_counter() { if(stage_) jfgpu_free_dev(t_, stage_); if(t_) jfgpu_destroy(t_); }

  // Multi-GPU (one process per GPU): with a communicator attached, every sequence buffer is one collective
  // step -- this rank's k-mers are routed to the shards that own them and what arrives is inserted here.  Ranks read
  // different amounts of input, so each step starts by agreeing that somebody still has some; done() keeps stepping with
  // nothing until nobody has (every rank makes the same number of steps), then completes the exchange.
  void attach_comm(jfgpu_comm* c) { comm_ = c; }
  jfgpu_comm* comm() const { return comm_; }
  hash_counter(const hash_counter&) = delete;
  hash_counter& operator=(const hash_counter&) = delete;

  jfgpu_table* handle() { return t_; }
  size_t size() const { return info_.size; }
  uint16_t key_len() const { return (uint16_t)info_.key_len; }
  uint16_t val_len() const { return (uint16_t)info_.val_len; }
  uint16_t lsize() const { return (uint16_t)info_.lsize; }
  uint16_t nb_threads() const { return nb_threads_; }
  uint16_t max_reprobe() const { return (uint16_t)std::min<uint32_t>(info_.max_reprobe, 65535); }
  const jfgpu_info& info() const { return info_; }
  hash_counter* ary() { return this; }   // hash_counter::ary() (hash_counter.hpp:70)

  header_matrix matrix() const {
    header_matrix m;
    m.r = info_.lsize; m.c = info_.key_len; m.identity = info_.matrix_identity;
    m.columns.assign(m.c, 0);
    jf_check(jfgpu_get_matrix(t_, m.columns.data()));
    return m;
  }

  // The table may have doubled itself since construction (size is a hint): re-read the geometry.
  void refresh_info() { jf_check(jfgpu_get_info(t_, &info_)); }
  // hash_counter::do_size_doubling (hash_counter.hpp:78-79)
  void do_size_doubling(bool v) { jf_check(jfgpu_set_growth(t_, v ? 1 : 0)); }
  // hash_counter::dumper(d) (hash_counter.hpp:81): with doubling off, a table that fills up is handed to `spill`
  // (which writes it out as one sorted run) and emptied, instead of failing with "Hash full".
  void on_full(std::function<void()> spill) {
    spill_ = std::move(spill);
    jf_check(jfgpu_set_spill(t_, spill_ ? &hash_counter::spill_trampoline : nullptr, this));
  }

  // file_header::update_from_ary (file_header.hpp:25-33)
  void update_header(file_header& h) {
    refresh_info();
    h.size(info_.size);
    h.key_len(info_.key_len);
    h.val_len(info_.val_len);
    h.matrix(matrix());
    // The probing schedule is never serialised; readers only need the fields to exist and
    // merge only compares reprobes[max_reprobe] across files (merge_files.cc:128,145-146).
    // Advertise triangular numbers capped to the tile, like the reference's quadratic_reprobes.
    const unsigned mr = std::min<unsigned>(info_.max_reprobe, 126);
    h.max_reprobe(mr);
    std::vector<size_t> rp(mr + 1);
    for(unsigned i = 0; i <= mr; ++i) rp[i] = i == 0 ? 1 : (size_t)i * (i + 1) / 2;
    h.set_reprobes(rp);
  }

  // ---- hot path -----------------------------------------------------------
  // One parser-contract buffer (count_main.cc:152-163 loop body on the device).
  void count_sequence(const char* bases, size_t n) {
    flush();
    if(!comm_) { jf_check(jfgpu_count_ascii(t_, bases, n)); return; }
    if(n > stage_cap_) {                           // sharded: the routing kernels read device memory
      if(stage_) jf_check(jfgpu_free_dev(t_, stage_));
      stage_ = nullptr; stage_cap_ = 0;
      jf_check(jfgpu_malloc_dev(t_, n + (n >> 2) + 64, &stage_));
      stage_cap_ = n + (n >> 2);
    }
    jf_check(jfgpu_memcpy_h2d(t_, stage_, bases, n));
    count_sequence_dev((const char*)stage_, n);
  }
  // The same for a buffer already in device memory (device_sequence_parser); wait_consumed() says
  // when buffers handed over so far may be overwritten.
  void count_sequence_dev(const char* d_bases, size_t n) {
    flush();
    if(!comm_) { jf_check(jfgpu_count_ascii_dev(t_, d_bases, n)); return; }
    uint64_t any = 1;
    jf_check(jfgpu_comm_allreduce_u64(comm_, &any, 1, 0));
    jf_check(jfgpu_comm_count_ascii_dev(comm_, t_, d_bases, n));    // returns when the buffer has been read
  }
  void wait_consumed() { jf_check(jfgpu_wait(t_)); }
  // What count_sequence does with a k-mer: COUNT add(m, 1), PRIME set(m), UPDATE update_add(m, 1)
  // (the OPERATION of mer_counter_base, count_main.cc:133,152-184).
  enum operation { COUNT = JFGPU_OP_COUNT, PRIME = JFGPU_OP_PRIME, UPDATE = JFGPU_OP_UPDATE };
  void set_operation(operation op) { flush(); jf_check(jfgpu_set_operation(t_, (int)op)); }
  // Expected amount of sequence before the next done(): lets the engine size its partition workspace once
  // instead of growing it batch by batch (each growth applies what is pending first).  Best effort.
  void expect_input(uint64_t bytes) { flush(); jfgpu_reserve(t_, bytes); }

  // hash_counter::add(k, v) (hash_counter.hpp:122-126): batched, thread-safe.
  void add(const mer_dna& k, uint64_t v) {
    std::lock_guard<std::mutex> lock(mu_);
    if(v != pending_val_ && !pending_.empty()) flush_locked();
    pending_val_ = v;
    for(unsigned i = 0; i < kw_; ++i) pending_.push_back(k.word(i));
    if(pending_.size() >= kBatch * kw_) flush_locked();
  }
  // a batch of (key, value) pairs at once (keys: key_words() words each, little-endian): loading a sorted file
  void add_batch(const uint64_t* keys, const uint64_t* vals, size_t n) {
    flush();
    jf_check(jfgpu_add_key_vals(t_, keys, vals, n));
  }
  // add(k, v, &is_new, &id) (hash_counter.hpp:91-115): synchronous (SWIG HashCounter.add).
  void add(const mer_dna& k, uint64_t v, bool* is_new, size_t* id = nullptr) {
    flush();
    uint8_t nw = 0;
    jf_check(jfgpu_add_keys(t_, k.data(), 1, v, &nw));
    if(is_new) *is_new = nw != 0;
    if(id) *id = 0;
  }
  // set(k) (hash_counter.hpp:132-145): make the key present with value 0.
  void set(const mer_dna& k, bool* is_new = nullptr, size_t* id = nullptr) { add(k, 0, is_new, id); }
  // update_add(k, v) (hash_counter.hpp:150-166): add only if the key is present.
  bool update_add(const mer_dna& k, uint64_t v) {
    uint64_t val = 0;
    if(!get_val_for_key(k, &val)) return false;
    bool is_new;
    add(k, v, &is_new);
    return true;
  }
  // done() (hash_counter.hpp:169-172): everything retired; throws "Hash full" if it did not fit.
  void done() {
    flush();
    if(comm_) {
      while(true) {                                // other ranks may still have input: step with nothing until nobody has
        uint64_t any = 0;
        jf_check(jfgpu_comm_allreduce_u64(comm_, &any, 1, 0));
        if(!any) break;
        jf_check(jfgpu_comm_count_ascii_dev(comm_, t_, nullptr, 0));
      }
      uint64_t sent = 0, received = 0;
      jf_check(jfgpu_comm_finish(comm_, &sent, &received));
      uint64_t v[2] = {sent, received};
      jf_check(jfgpu_comm_allreduce_u64(comm_, v, 2, 0));
      if(v[0] != v[1]) throw std::runtime_error("multi-GPU exchange lost k-mers: " + std::to_string(v[0]) + " sent, " + std::to_string(v[1]) + " received");
    }
    jf_check(jfgpu_sync(t_));
    refresh_info();
  }
  void clear() { std::lock_guard<std::mutex> lock(mu_); pending_.clear(); jf_check(jfgpu_clear(t_)); }

  // array::get_val_for_key (large_hash_array.hpp:354-372)
  bool get_val_for_key(const mer_dna& k, uint64_t* val) {
    flush();
    uint64_t v = 0;
    uint8_t f = 0;
    jf_check(jfgpu_lookup(t_, k.data(), 1, &v, &f));
    if(val) *val = v;
    return f != 0;
  }
  bool has_key(const mer_dna& k) { return get_val_for_key(k, nullptr); }

  void flush() { if(in_spill_) return; std::lock_guard<std::mutex> lock(mu_); flush_locked(); }   // (a spill runs inside an engine call)

private:
  std::function<void()> spill_;
  bool in_spill_ = false;
  static int spill_trampoline(void* self) {
    hash_counter* h = static_cast<hash_counter*>(self);
    h->in_spill_ = true;
    int rc = 0;
    try { h->spill_(); } catch(std::exception& e) { fprintf(stderr, "%s\n", e.what()); rc = 1; }
    h->in_spill_ = false;
    return rc;
  }

  static constexpr size_t kBatch = 1 << 20;
  jfgpu_comm* comm_ = nullptr;
  void* stage_ = nullptr; size_t stage_cap_ = 0;
  jfgpu_table* t_ = nullptr;
  jfgpu_info info_;
  uint16_t nb_threads_;
  unsigned kw_ = 1;                 // 64-bit words per key
  std::mutex mu_;
  std::vector<uint64_t> pending_;
  uint64_t pending_val_ = 1;

  void flush_locked() {
    if(pending_.empty()) return;
    std::vector<uint64_t> batch;
    batch.swap(pending_);
    pending_.reserve(kBatch * kw_);
    jf_check(jfgpu_add_keys(t_, batch.data(), batch.size() / kw_, pending_val_, nullptr));
  }
};

typedef hash_counter mer_hash;
typedef hash_counter mer_array;

}  // namespace jellyfish_amd
