// jellyfish/hash_counter.hpp (compat): jellyfish::cooperative::hash_counter<jellyfish::mer_dna>
// (include/jellyfish/hash_counter.hpp:50-172) and the part of its array (large_hash_array.hpp:354-372,477-499,
// iterators :798-925) client programs use -- examples/jf_count_dump/jf_count_dump.cc, unit_tests/test_hash_counter.cc
// compile against this directory unchanged -- on top of the engine's facade (jellyfish_amd::hash_counter -> C ABI).
//
// Differences a program can observe: the table lives in HBM, so add() batches (a k-mer becomes visible at done() or at
// the first read), iteration walks a snapshot taken when the iterator is made (in the reference's (position, key) order),
// val_len() reports what the constructor was given (in-memory counts are exact 64-bit), and the id of add(k, v, &is_new,
// &id) is not a slot address (0).
#pragma once
#include <iterator>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>
#include <jellyfish_amd/hash_counter.hpp>
#include <jellyfish/mer_dna.hpp>

namespace jellyfish {
namespace cooperative {

template <typename Key> class hash_counter;

template <>
class hash_counter<mer_dna> {
  typedef jellyfish_amd::hash_counter engine_type;

public:
  typedef mer_dna key_type;
  typedef uint64_t mapped_type;
  typedef std::pair<key_type, mapped_type> value_type;

  // large_hash::array as seen by readers of a counted table
  class array {
  public:
    typedef mer_dna key_type;
    typedef uint64_t mapped_type;
    typedef std::pair<key_type, mapped_type> value_type;
  private:
    typedef std::vector<value_type> snapshot_type;
  public:

    // eager_iterator (large_hash_array.hpp:881-925): a forward iterator over (key, value) pairs
    class eager_iterator : public std::iterator<std::forward_iterator_tag, value_type> {
    public:
      eager_iterator() : pos_(0) {}
      eager_iterator(std::shared_ptr<const snapshot_type> s, size_t pos) : snap_(std::move(s)), pos_(pos) {}
      bool operator==(const eager_iterator& o) const { return at_end() ? o.at_end() : (!o.at_end() && pos_ == o.pos_); }
      bool operator!=(const eager_iterator& o) const { return !(*this == o); }
      const value_type& operator*() const { return (*snap_)[pos_]; }
      const value_type* operator->() const { return &(*snap_)[pos_]; }
      eager_iterator& operator++() { ++pos_; return *this; }
      eager_iterator operator++(int) { eager_iterator r(*this); ++pos_; return r; }
      // the reference's own accessors
      const key_type& key() const { return (*snap_)[pos_].first; }
      const mapped_type& val() const { return (*snap_)[pos_].second; }
    private:
      bool at_end() const { return !snap_ || pos_ >= snap_->size(); }
      std::shared_ptr<const snapshot_type> snap_;
      size_t pos_;
    };
    typedef eager_iterator iterator;
    typedef eager_iterator const_iterator;

    // lazy_iterator / region iterators (large_hash_array.hpp:798-879): while(it.next()) { it.key(); it.val(); }
    class lazy_iterator {
    public:
      lazy_iterator() : pos_(0), started_(false) {}
      lazy_iterator(std::shared_ptr<const snapshot_type> s, size_t first, size_t last)
          : snap_(std::move(s)), pos_(first), last_(last), started_(false) {}
      bool next() {
        if(started_) ++pos_; else started_ = true;
        return snap_ && pos_ < last_;
      }
      const key_type& key() const { return (*snap_)[pos_].first; }
      mapped_type val() const { return (*snap_)[pos_].second; }
      size_t id() const { return pos_; }
      size_t pos() const { return pos_; }
    private:
      std::shared_ptr<const snapshot_type> snap_;
      size_t pos_, last_ = 0;
      bool started_;
    };
    typedef lazy_iterator region_iterator;

    explicit array(hash_counter* owner) : owner_(owner) {}
    size_t size() const { return owner_->size(); }
    uint16_t key_len() const { return owner_->key_len(); }
    uint16_t val_len() const { return owner_->val_len(); }

    // get_val_for_key (large_hash_array.hpp:354-372)
    bool get_val_for_key(const key_type& k, mapped_type* val) const { return owner_->eng_.get_val_for_key(k, val); }
    bool get_val_for_key(const key_type& k, mapped_type* val, key_type&, size_t* id) const { if(id) *id = 0; return get_val_for_key(k, val); }
    bool has_key(const key_type& k) const { return owner_->eng_.has_key(k); }

    eager_iterator begin() const { return eager_iterator(snapshot(), 0); }
    eager_iterator end() const { return eager_iterator(); }
    template <typename It> It iterator_all() const { auto s = snapshot(); const size_t n = s->size(); return It(std::move(s), 0, n); }
    // slice i of n (large_hash_array.hpp:477-499)
    template <typename It> It iterator_slice(size_t index, size_t nb_slices) const {
      auto s = snapshot();
      const size_t n = s->size(), a = n * index / nb_slices, b = n * (index + 1) / nb_slices;
      return It(std::move(s), a, b);
    }
    lazy_iterator lazy_slice(size_t i, size_t n) const { return iterator_slice<lazy_iterator>(i, n); }
    lazy_iterator region_slice(size_t i, size_t n) const { return iterator_slice<lazy_iterator>(i, n); }

  private:
    // every record of the table, in (position, key) order, keys decoded: what sorted_dumper writes (sorted_dumper.hpp:57-101)
    std::shared_ptr<const snapshot_type> snapshot() const {
      engine_type& e = owner_->eng_;
      e.flush();
      std::shared_ptr<snapshot_type> s(new snapshot_type);
      uint64_t n = 0; uint32_t rec = 0;
      jellyfish_amd::jf_check(jfgpu_dump_begin(e.handle(), 0, ~(uint64_t)0, &n, &rec));
      s->reserve(n);
      const uint64_t cap = std::max<uint64_t>((uint64_t)1 << 18, e.info().tile_slots);
      std::vector<unsigned char> buf(cap * rec);
      const unsigned k = e.info().k, kb = (2 * k + 7) / 8, vb = rec - kb;
      try {
        while(true) {
          uint64_t got = 0;
          jellyfish_amd::jf_check(jfgpu_dump_next(e.handle(), buf.data(), cap, &got));
          if(!got) break;
          for(uint64_t i = 0; i < got; ++i) {
            const unsigned char* r = &buf[i * rec];
            value_type v(mer_dna(k), 0);
            memset(v.first.data__(), 0, v.first.nb_words() * sizeof(uint64_t));
            memcpy(v.first.data__(), r, kb); memcpy(&v.second, r + kb, vb < 8 ? vb : 8);
            s->push_back(std::move(v));
          }
        }
      } catch(...) { jfgpu_dump_end(e.handle()); throw; }
      jellyfish_amd::jf_check(jfgpu_dump_end(e.handle()));
      return s;
    }
    hash_counter* owner_;
  };

  // hash_counter(size, key_len, val_len, nb_threads, reprobe_limit) (hash_counter.hpp:56-64)
  hash_counter(size_t size, uint16_t key_len, uint16_t val_len, uint16_t nb_threads, uint16_t reprobe_limit = 126)
      : eng_(size, key_len, val_len, nb_threads, reprobe_limit, false, -1, 8), ary_(this), val_len_(val_len), doubling_(true) {}

  array* ary() { return &ary_; }
  const array* ary() const { return &ary_; }
  size_t size() { eng_.refresh_info(); return eng_.size(); }
  uint16_t key_len() const { return eng_.key_len(); }
  uint16_t val_len() const { return val_len_; }
  uint16_t nb_threads() const { return eng_.nb_threads(); }
  uint16_t reprobe_limit() const { return eng_.max_reprobe(); }
  bool do_size_doubling() const { return doubling_; }
  void do_size_doubling(bool v) { doubling_ = v; eng_.do_size_doubling(v); }

  // add / set / update_add (hash_counter.hpp:91-166)
  void add(const mer_dna& k, uint64_t v) { eng_.add(k, v); }
  void add(const mer_dna& k, uint64_t v, bool* is_new, size_t* id) { eng_.add(k, v, is_new, id); }
  void set(const mer_dna& k) { eng_.add(k, 0); }
  void set(const mer_dna& k, bool* is_new, size_t* id) { eng_.set(k, is_new, id); }
  bool update_add(const mer_dna& k, uint64_t v) { return eng_.update_add(k, v); }
  bool update_add(const mer_dna& k, uint64_t v, mer_dna&) { return eng_.update_add(k, v); }
  // done() (hash_counter.hpp:169-172): called by every worker thread when it is finished; whatever is batched goes in
  void done() { std::lock_guard<std::mutex> l(done_mu_); eng_.done(); }

  engine_type& engine() { return eng_; }

private:
  friend class array;
  mutable engine_type eng_;
  array ary_;
  uint16_t val_len_;
  bool doubling_;
  std::mutex done_mu_;
};

}  // namespace cooperative
}  // namespace jellyfish
