// jellyfish/cpp_array.hpp (compat): fixed-size array of objects constructed in place (include/jellyfish/cpp_array.hpp:19-110).
#pragma once
#include <cstddef>
#include <memory>
#include <new>
#include <utility>
#include <vector>
namespace jellyfish {
template <typename T>
class cpp_array {
public:
  explicit cpp_array(size_t size) : data_(static_cast<T*>(::operator new(sizeof(T) * size))), init_(size, false), size_(size) {}
  ~cpp_array() { clear(); ::operator delete(data_); }
  cpp_array(const cpp_array&) = delete;
  cpp_array& operator=(const cpp_array&) = delete;
  template <typename... Args> void init(size_t i, Args&&... args) { release(i); new(&data_[i]) T(std::forward<Args>(args)...); init_[i] = true; }
  void release(size_t i) { if(init_[i]) { data_[i].~T(); init_[i] = false; } }
  void clear() { for(size_t i = 0; i < size_; ++i) release(i); }
  size_t size() const { return size_; }
  bool empty() const { return size_ == 0; }
  bool initialized(size_t i) const { return init_[i]; }
  T& operator[](size_t i) { return data_[i]; }
  const T& operator[](size_t i) const { return data_[i]; }
  T* begin() { return data_; }
  T* end() { return data_ + size_; }
  T* data() { return data_; }
private:
  T* data_;
  std::vector<bool> init_;
  size_t size_;
};
}  // namespace jellyfish
