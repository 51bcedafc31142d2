// jellyfish/mer_iterator.hpp (compat): the k-mers of a parser's buffers, one at a time, optionally canonical
// (include/jellyfish/mer_iterator.hpp:28-104): any character outside [ACGTacgt] restarts the window.
#pragma once
#include <iterator>
#include <memory>
#include <string>
namespace jellyfish {
template <typename SequencePool, typename MerType>
class mer_iterator : public std::iterator<std::input_iterator_tag, MerType> {
public:
  typedef MerType mer_type;
  typedef SequencePool sequence_parser_type;
  mer_iterator(SequencePool& seq, bool canonical = false)
      : pool_(&seq), m_(), rcm_(), filled_(0), canonical_(canonical), pos_(0) {
    buf_ = pool_->next();
    if(buf_) this->operator++(); else pool_ = nullptr;
  }
  mer_iterator() : pool_(nullptr), filled_(0), canonical_(false), pos_(0) {}

  bool operator==(const mer_iterator& rhs) const { return pool_ == rhs.pool_; }
  bool operator!=(const mer_iterator& rhs) const { return pool_ != rhs.pool_; }
  operator void*() const { return (void*)pool_; }
  const mer_type& operator*() const { return !canonical_ || m_ < rcm_ ? m_ : rcm_; }
  const mer_type* operator->() const { return &this->operator*(); }
  mer_iterator& operator++() {
    while(true) {
      while(pos_ < buf_->size()) {
        const int code = mer_type::code((*buf_)[pos_++]);
        if(code >= 0) {
          m_.shift_left(code);
          if(canonical_) rcm_.shift_right(mer_type::complement(code));
          filled_ = filled_ < m_.mer_k() ? filled_ + 1 : filled_;
          if(filled_ >= m_.mer_k()) return *this;
        } else
          filled_ = 0;
      }
      buf_ = pool_->next();                    // k-mers never span buffers
      pos_ = 0; filled_ = 0;
      if(!buf_) { pool_ = nullptr; return *this; }
    }
  }
  mer_iterator operator++(int) { mer_iterator res(*this); ++*this; return res; }

private:
  SequencePool* pool_;
  std::shared_ptr<std::string> buf_;
  mer_type m_, rcm_;
  unsigned int filled_;
  bool canonical_;
  size_t pos_;
};
}  // namespace jellyfish
