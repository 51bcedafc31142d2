// jellyfish_amd/csrc/kernels_bloom.hip.hpp -- Bloom counter of `jellyfish bc` (BASELINE config 3).
//
// Reference (paths relative to /root/reference):
//   include/jellyfish/bloom_counter2.hpp:56-107  insert__: nb_hashes cells, cell_i = (h0 % m + i * (h1 % m)) % m,
//                                                cell = base-3 digit (p % 5) of byte (p / 5), CAS-increment while < 2,
//                                                returns the minimum previous digit
//   include/jellyfish/bloom_counter2.hpp:109-142 check__: minimum digit
//   include/jellyfish/mer_dna_bloom_counter.hpp:19-34  h0 = M1 * key, h1 = M2 * key, two 64-row GF(2) matrices
//   sub_commands/bc_main.cc:67-71                the loop: filter.insert(*mers) for every (canonical) k-mer
//   sub_commands/count_main.cc:115-118           count --bc: a k-mer is admitted iff check(m) > 1
// The final byte array is independent of the insertion order (saturating commutative increments), so the
// device result is byte-identical to the reference's for the same matrices.
#pragma once
#include "kernels.hip.hpp"

namespace jfgpu {

__device__ __constant__ const uint32_t kPow3[5] = {1, 3, 9, 27, 81};

// p / 5 and p % 5 without a 64-bit divide.
__device__ inline void divmod5(uint64_t p, uint64_t& q, uint32_t& r) {
  q = __umul64hi(p, 0xCCCCCCCCCCCCCCCDull) >> 2;
  r = (uint32_t)(p - q * 5);
}

// Increment digit `boff` of the byte at byte index `byte` unless it is already 2.  Returns the previous digit.
__device__ inline uint32_t bloom_bump(uint32_t* words, uint64_t byte, uint32_t boff) {
  uint32_t* w = words + (byte >> 2);
  const uint32_t sh = 8 * (uint32_t)(byte & 3);
  uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while(true) {
    const uint32_t v = (old >> sh) & 0xFFu;
    const uint32_t d = (v / kPow3[boff]) % 3;
    if(d == 2) return 2;
    const uint32_t nw = old + (kPow3[boff] << sh);      // v + 3^boff <= 242: never carries out of the byte
    const uint32_t seen = atomicCAS(w, old, nw);
    if(seen == old) return d;
    old = seen;
  }
}

__device__ inline uint32_t bloom_insert(const DevBloom& B, uint64_t h0, uint64_t h1) {
  const uint64_t base = h0 % B.m, inc = h1 % B.m;
  uint64_t p = base;
  uint32_t res = 2;
  for(uint32_t i = 0; i < B.nh; ++i) {
    uint64_t byte; uint32_t boff;
    divmod5(p, byte, boff);
    const uint32_t d = bloom_bump(B.data, byte, boff);
    res = d < res ? d : res;
    p += inc; if(p >= B.m) p -= B.m;                     // == (base + (i+1) * inc) % m, base, inc < m
  }
  return res;
}

__device__ inline uint32_t bloom_check(const DevBloom& B, uint64_t h0, uint64_t h1) {
  const uint64_t base = h0 % B.m, inc = h1 % B.m;
  uint64_t p = base;
  uint32_t res = 2;
  for(uint32_t i = 0; i < B.nh; ++i) {
    uint64_t byte; uint32_t boff;
    divmod5(p, byte, boff);
    const uint32_t v = (B.data[byte >> 2] >> (8 * (uint32_t)(byte & 3))) & 0xFFu;
    const uint32_t d = (v / kPow3[boff]) % 3;
    res = d < res ? d : res;
    p += inc; if(p >= B.m) p -= B.m;
  }
  return res;
}

// count --bc filter (count_main.cc:115-118); tables read through the caches (12-16 KiB hot set).
__device__ inline bool bloom_admits(const DevBloom& B, uint64_t key) {
  const uint64_t h0 = hash_tables(B.tbl1, key, B.nbytes), h1 = hash_tables(B.tbl2, key, B.nbytes);
  return bloom_check(B, h0, h1) > 1;
}

// K4: insert every (canonical) k-mer of a contract buffer into the Bloom counter.
__global__ __launch_bounds__(kBlock) void bloom_insert_ascii_kernel(DevBloom B, TableGeom g, const uint8_t* __restrict__ base,
                                                                    int64_t lo, int64_t hi, unsigned long long* __restrict__ mers) {
  __shared__ uint64_t s_t1[8 * 256];
  __shared__ uint64_t s_t2[8 * 256];
  __shared__ uint32_t s_codes[kBlock + 2];
  __shared__ uint32_t s_inv[kBlock + 2];
  load_tables_lds(s_t1, B.tbl1, B.nbytes);
  load_tables_lds(s_t2, B.tbl2, B.nbytes);
  const int64_t n_tiles = (hi + kTilePos - 1) / kTilePos;
  uint32_t my = 0;
  for(int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    __syncthreads();
    const LaneWords L = stage_tile(base, tile * kTilePos, lo, hi, s_codes, s_inv);
    for_each_kmer(g, L, [&](int, uint64_t key) {
      ++my;
      bloom_insert(B, hash_tables(s_t1, key, B.nbytes), hash_tables(s_t2, key, B.nbytes));
    });
  }
  uint64_t w = my;
  for(int o = 32; o > 0; o >>= 1) w += __shfl_down(w, o, 64);
  if((threadIdx.x & 63) == 0 && w) atomicAdd(mers, (unsigned long long)w);
}

// check() / insert() on encoded keys (query_main.cc Bloom branch; unit-test style access).
__global__ __launch_bounds__(kBlock) void bloom_keys_kernel(DevBloom B, const uint64_t* __restrict__ keys, uint64_t n,
                                                            uint8_t* __restrict__ out, int do_insert) {
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t h0 = hash_tables(B.tbl1, keys[i], B.nbytes), h1 = hash_tables(B.tbl2, keys[i], B.nbytes);
    const uint32_t r = do_insert ? bloom_insert(B, h0, h1) : bloom_check(B, h0, h1);
    if(out) out[i] = (uint8_t)r;
  }
}

}  // namespace jfgpu
