// jellyfish/misc.hpp (compat): the few helpers of include/jellyfish/misc.hpp client programs touch.
#pragma once
#include <stdint.h>
#include <cstdlib>
#include <string>
namespace jellyfish {
inline uint64_t bitsize(uint64_t n) { uint64_t b = 0; while(n) { ++b; n >>= 1; } return b; }          // misc.hpp floorLog2 + 1
inline unsigned ceilLog2(uint64_t n) { unsigned b = 0; while(((uint64_t)1 << b) < n) ++b; return b; }
inline uint64_t random_bits(int length) { uint64_t r = 0; for(int i = 0; i < length; i += 16) r = (r << 16) ^ (uint64_t)(random() & 0xFFFF); return length >= 64 ? r : r & (((uint64_t)1 << length) - 1); }
inline uint64_t random_bits() { return random_bits(64); }
}  // namespace jellyfish
