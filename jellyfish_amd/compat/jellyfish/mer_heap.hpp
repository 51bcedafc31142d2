// jellyfish/mer_heap.hpp (compat): the k-way merge heap over sorted readers (include/jellyfish/mer_heap.hpp:17-106): items
// ordered by (position under the readers' matrix, key), smallest on top -- the order of binary/sorted files.
#pragma once
#include <algorithm>
#include <vector>
namespace jellyfish {
namespace mer_heap {
template <typename Key, typename Iterator>
struct heap_item {
  Key key_;
  uint64_t val_ = 0;
  size_t pos_ = 0;
  Iterator* it_ = nullptr;
  heap_item() {}
  explicit heap_item(Iterator& iter) { initialize(iter); }
  void initialize(Iterator& iter) { key_ = iter.key(); val_ = iter.val(); pos_ = iter.pos(); it_ = &iter; }
  bool operator>(const heap_item& o) const { return pos_ == o.pos_ ? key_ > o.key_ : pos_ > o.pos_; }
};
template <typename Key, typename Iterator>
struct heap_item_comp {
  bool operator()(const heap_item<Key, Iterator>* a, const heap_item<Key, Iterator>* b) const { return *a > *b; }
};
template <typename Key, typename Iterator>
class heap {
public:
  typedef heap_item<Key, Iterator> item_type;
  typedef const item_type* const_item_t;
  heap() : h_(0) {}
  explicit heap(size_t capacity) : h_(0) { initialize(capacity); }
  void initialize(size_t capacity) { store_.assign(capacity, item_type()); elts_.resize(capacity); for(size_t i = 0; i < capacity; ++i) elts_[i] = &store_[i]; h_ = 0; }
  bool is_empty() const { return h_ == 0; }
  bool is_not_empty() const { return h_ > 0; }
  size_t size() const { return h_; }
  size_t capacity() const { return elts_.size(); }
  const_item_t head() const { return elts_[0]; }
  void pop() { std::pop_heap(elts_.begin(), elts_.begin() + h_--, comp_); }
  void push(Iterator& item) { elts_[h_]->initialize(item); std::push_heap(elts_.begin(), elts_.begin() + ++h_, comp_); }
private:
  std::vector<item_type> store_;
  std::vector<item_type*> elts_;
  size_t h_;
  heap_item_comp<Key, Iterator> comp_;
};
}  // namespace mer_heap
}  // namespace jellyfish
