// jellyfish/err.hpp (compat): err::msg / err::die as client programs use them (include/jellyfish/err.hpp:27-110).
#pragma once
#include <cstdlib>
#include <exception>
#include <iostream>
#include <sstream>
#include <string>
namespace jellyfish {
namespace err {
struct msg {
  std::ostringstream msg_;
  msg() {}
  template <typename T> explicit msg(const T& x) { *this << x; }
  operator std::string() const { return msg_.str(); }
  template <typename T> msg& operator<<(const T& x) { msg_ << x; return *this; }
  msg& operator<<(const std::exception& e) { msg_ << e.what(); return *this; }
};
inline std::ostream& operator<<(std::ostream& os, const msg& m) { return os << m.msg_.str(); }
[[noreturn]] inline void die(const std::string& m) { std::cerr << m << std::endl; exit(1); }
[[noreturn]] inline void die(const msg& m) { die(m.msg_.str()); }
[[noreturn]] inline void die(const char* m) { die(std::string(m)); }
}  // namespace err
}  // namespace jellyfish
