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

// p / 5 and p % 5 without a 64-bit divide.
__device__ inline void divmod5(uint64_t p, uint64_t& q, uint32_t& r) {
  q = __umul64hi(p, 0xCCCCCCCCCCCCCCCDull) >> 2;
  r = (uint32_t)(p - q * 5);
}

// digit j (0..4) of a byte holding five base-3 cells, and 3^j, without tables or variable divisors
__device__ inline uint32_t bloom_digit(uint32_t v, uint32_t j) {
  const uint32_t q = j == 0 ? v : j == 1 ? v / 3u : j == 2 ? v / 9u : j == 3 ? v / 27u : v / 81u;
  return q % 3u;
}
__device__ inline uint32_t bloom_pow3(uint32_t j) { return j == 0 ? 1u : j == 1 ? 3u : j == 2 ? 9u : j == 3 ? 27u : 81u; }

// Increment digit `boff` of the byte at byte index `byte` unless it is already 2.  Returns the previous digit.
__device__ inline uint32_t bloom_bump(uint32_t* words, uint64_t byte, uint32_t boff) {
  uint32_t* w = words + (byte >> 2);
  const uint32_t sh = 8 * (uint32_t)(byte & 3);
  uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while(true) {
    const uint32_t v = (old >> sh) & 0xFFu;
    const uint32_t d = bloom_digit(v, boff);
    if(d == 2) return 2;
    const uint32_t nw = old + (bloom_pow3(boff) << sh);  // v + 3^boff <= 242: never carries out of the byte
    const uint32_t seen = atomicCAS(w, old, nw);
    if(seen == old) return d;
    old = seen;
  }
}

// x % m by a precomputed reciprocal (bloom_counter2.hpp:60-64 uses divisor64 for the same reason): the estimate
// floor(x * floor(2^64 / m) / 2^64) is the quotient or one less, so at most one correction is needed (two are allowed).
__device__ __host__ inline uint64_t bloom_mod(uint64_t x, uint64_t m, uint64_t recip) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint64_t q = __umul64hi(x, recip);
#else
  const uint64_t q = (uint64_t)(((unsigned __int128)x * recip) >> 64);
#endif
  uint64_t r = x - q * m;
  if(r >= m) r -= m;
  if(r >= m) r -= m;
  return r;
}
inline uint64_t bloom_recip(uint64_t m) { return m <= 1 ? ~0ull : (uint64_t)((((unsigned __int128)1) << 64) / m); }

__device__ inline uint32_t bloom_insert(const DevBloom& B, uint64_t h0, uint64_t h1) {
  const uint64_t base = bloom_mod(h0, B.m, B.recip), inc = bloom_mod(h1, B.m, B.recip);
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
  const uint64_t base = bloom_mod(h0, B.m, B.recip), inc = bloom_mod(h1, B.m, B.recip);
  uint64_t p = base;
  uint32_t res = 2;
  for(uint32_t i = 0; i < B.nh; ++i) {
    uint64_t byte; uint32_t boff;
    divmod5(p, byte, boff);
    const uint32_t v = (B.data[byte >> 2] >> (8 * (uint32_t)(byte & 3))) & 0xFFu;
    const uint32_t d = bloom_digit(v, boff);
    res = d < res ? d : res;
    p += inc; if(p >= B.m) p -= B.m;
  }
  return res;
}

// check(m) > 1, i.e. every one of the nh cells holds 2, decided as early as possible: the first cell below 2 ends the
// search (same answer as bloom_check() > 1).  A k-mer seen once fails at its first or second cell with high
// probability, so the filter pass of `count --bc` costs ~1.2 random reads per singleton instead of nh; cells are
// read two at a time so that a k-mer that passes does not pay nh dependent round trips.
__device__ inline bool bloom_all_two(const DevBloom& B, uint64_t h0, uint64_t h1) {
  const uint64_t base = bloom_mod(h0, B.m, B.recip), inc = bloom_mod(h1, B.m, B.recip);
  uint64_t p = base;
  for(uint32_t i = 0; i < B.nh; i += 2) {
    uint64_t p2 = p + inc; if(p2 >= B.m) p2 -= B.m;
    uint64_t b0, b1; uint32_t o0, o1;
    divmod5(p, b0, o0); divmod5(p2, b1, o1);
    const bool two = i + 1 < B.nh;
    const uint32_t w0 = B.data[b0 >> 2];
    const uint32_t w1 = two ? B.data[b1 >> 2] : 0u;
    if(bloom_digit((w0 >> (8 * (uint32_t)(b0 & 3))) & 0xFFu, o0) < 2) return false;
    if(two && bloom_digit((w1 >> (8 * (uint32_t)(b1 & 3))) & 0xFFu, o1) < 2) return false;
    p = p2 + inc; if(p >= B.m) p -= B.m;
  }
  return true;
}

// count --bf-size (bloom_filter.hpp:44-68, filter_bf count_main.cc:121-131): set the nh bits of the key, admit it iff all of
// them were set already -- the first sighting of a k-mer only marks it, later ones are counted.  One atomic OR per bit,
// like the reference's __sync_fetch_and_or; concurrent first sightings of one k-mer can both come out "new" there and
// here alike, which is why the reference only bounds the result statistically (tests/bloom_filter.sh).
__device__ inline bool bloom_filter_insert(const DevBloom& B, uint64_t h0, uint64_t h1) {
  const uint64_t base = bloom_mod(h0, B.m, B.recip), inc = bloom_mod(h1, B.m, B.recip);
  uint64_t p = base;
  bool present = true;
  for(uint32_t i = 0; i < B.nh; ++i) {
    const uint32_t bit = 1u << (uint32_t)(p & 31);
    const uint32_t old = atomicOr(&B.data[p >> 5], bit);
    present = present && (old & bit);
    p += inc; if(p >= B.m) p -= B.m;
  }
  return present;
}

// The filter for the 16 windows of one lane: bit j of the result = the window ending at the lane's position j is valid
// and check(m) > 1.  The cells are read in rounds: round r reads cell r of every window still undecided (all of a
// lane's loads of a round are in flight together), a window drops out at its first cell below 2.  With one decision
// loop per window a wave paid up to 16 x nh dependent memory round trips per tile; here it pays at most 2 x nh,
// usually 6-10 (a window survives a round with probability ~0.16 on a filter at its design load, unless its k-mer
// really was seen twice).  Eight windows at a time: sixteen would not fit the registers of a 1024-thread block.
// The cache of admitted k-mers (DevBloom::cache).  On high-coverage input most windows hold a k-mer the counter has seen
// twice: all nh cells read 2, ten random 128-byte line fetches per OCCURRENCE (config 3 on BASELINE.md's secondary
// distribution: 80 G cell reads, 1.4 s).  The answer is a function of the key alone while the counter is read-only, so
// it is asked once per distinct k-mer and remembered: a hit costs one 16-byte read of a 4 GB array and skips the two
// hashes as well.  Exactness: an entry is written only after the counter itself admitted that very key (full key
// compared, no fingerprints), entries are 8-byte words written whole (a torn pair of ways is two valid words), and a
// miss falls through to the counter -- the mask is the counter's, always.  The host turns the cache on when the first
// batches of a filtered pass admit a sizeable fraction of their windows (host_partition.inl: bloom_cache_decide).
__device__ __forceinline__ uint64_t bloom_cache_set(const DevBloom& B, uint64_t key) { return ((key * 0x9E3779B97F4A7C15ull) >> 23) & B.cache_mask; }
__device__ __forceinline__ bool bloom_cache_hit(const DevBloom& B, uint64_t key) {
  const uint64_t* set = B.cache + 2 * bloom_cache_set(B, key);
  const uint64_t a = set[0], b = set[1], tag = key + 1;
  // (tag 0 = the all-T 32-mer without -C: it would read as "an empty way" -- that k-mer is never cached and always asks
  // the counter; round-5 advisor finding)
  return tag != 0 && (a == tag || b == tag);
}
__device__ inline void bloom_cache_insert(const DevBloom& B, uint64_t key) {
  uint64_t* set = B.cache + 2 * bloom_cache_set(B, key);
  const uint64_t a = set[0], b = set[1], tag = key + 1;
  if(tag == 0 || a == tag || b == tag) return;
  const uint32_t way = a == 0 ? 0u : b == 0 ? 1u : (uint32_t)(key >> 9) & 1u;       // both taken: the key picks its victim
  set[way] = tag;
}

template <int J0>
__device__ inline uint32_t bloom_admit_half(const DevBloom& B, const TableGeom& g, const LaneWords& L, uint64_t& fw, uint64_t& rc) {
  constexpr int H = kPerLane / 2;
  const uint32_t k = g.k;
  const uint64_t kwin = k >= 64 ? ~0ull : ((1ull << k) - 1);
  const uint32_t rc_shift = 2 * (k - 1);
  const bool cached = B.cache != nullptr;                        // (uniform over the launch)
  const uint64_t fw_in = fw, rc_in = rc;
  uint64_t p[H], inc[H];
  uint32_t alive = 0, known = 0;
#pragma unroll
  for(int e = 0; e < H; ++e) {
    const int j = J0 + e;
    const uint64_t c = (L.cur >> (2 * (15 - j))) & 3u;
    fw = ((fw << 2) | c) & g.key_mask;
    rc = (rc >> 2) | ((3ull - c) << rc_shift);
    p[e] = 0; inc[e] = 0;
    if(((L.inv48 >> (15 - j)) & kwin) == 0) {
      const uint64_t key = (g.canonical && rc < fw) ? rc : fw;
      if(cached && bloom_cache_hit(B, key)) { known |= 1u << e; continue; }
      p[e] = bloom_mod(hash_tables(B.tbl1, key, B.nbytes), B.m, B.recip);
      inc[e] = bloom_mod(hash_tables(B.tbl2, key, B.nbytes), B.m, B.recip);
      alive |= 1u << e;
    }
  }
  for(uint32_t r = 0; r < B.nh && alive; ++r) {
    uint32_t wv[H];
#pragma unroll
    for(int e = 0; e < H; ++e) {
      wv[e] = 0;
      if((alive >> e) & 1u) { uint64_t byte; uint32_t dig; divmod5(p[e], byte, dig); wv[e] = B.data[byte >> 2]; }
    }
#pragma unroll
    for(int e = 0; e < H; ++e)
      if((alive >> e) & 1u) {
        uint64_t byte; uint32_t dig;
        divmod5(p[e], byte, dig);
        if(bloom_digit((wv[e] >> (8 * (uint32_t)(byte & 3))) & 0xFFu, dig) < 2) alive &= ~(1u << e);
        else { p[e] += inc[e]; if(p[e] >= B.m) p[e] -= B.m; }
      }
  }
  if(cached && alive) {                                          // what the counter has just admitted: remembered (keys rolled again)
    uint64_t f2 = fw_in, r2 = rc_in;
#pragma unroll 1
    for(int e = 0; e < H; ++e) {
      const uint64_t c = (L.cur >> (2 * (15 - (J0 + e)))) & 3u;
      f2 = ((f2 << 2) | c) & g.key_mask;
      r2 = (r2 >> 2) | ((3ull - c) << rc_shift);
      if((alive >> e) & 1u) bloom_cache_insert(B, (g.canonical && r2 < f2) ? r2 : f2);
    }
  }
  return (alive | known) << J0;
}
__device__ inline uint32_t bloom_admit_mask(const DevBloom& B, const TableGeom& g, const LaneWords& L) {
  uint64_t fw = (((uint64_t)L.p2 << 32) | L.p1) & g.key_mask;
  uint64_t rc = revcomp64(fw, g.k);
  if(B.kind == 1) {                      // one-pass Bloom filter: every window inserts, in order
    const uint64_t kwin = g.k >= 64 ? ~0ull : ((1ull << g.k) - 1);
    const uint32_t rc_shift = 2 * (g.k - 1);
    uint32_t adm = 0;
#pragma unroll 1
    for(int j = 0; j < kPerLane; ++j) {
      const uint64_t c = (L.cur >> (2 * (15 - j))) & 3u;
      fw = ((fw << 2) | c) & g.key_mask;
      rc = (rc >> 2) | ((3ull - c) << rc_shift);
      if(((L.inv48 >> (15 - j)) & kwin) != 0) continue;
      const uint64_t key = (g.canonical && rc < fw) ? rc : fw;
      if(bloom_filter_insert(B, hash_tables(B.tbl1, key, B.nbytes), hash_tables(B.tbl2, key, B.nbytes))) adm |= 1u << j;
    }
    return adm;
  }
  const uint32_t lo = bloom_admit_half<0>(B, g, L, fw, rc);
  return lo | bloom_admit_half<kPerLane / 2>(B, g, L, fw, rc);
}

// count --bc filter (count_main.cc:115-118); tables read through the caches (12-16 KiB hot set).
__device__ inline bool bloom_admits(const DevBloom& B, uint64_t key) {
  const uint64_t h0 = hash_tables(B.tbl1, key, B.nbytes), h1 = hash_tables(B.tbl2, key, B.nbytes);
  if(B.kind == 1) return bloom_filter_insert(B, h0, h1);
  return bloom_all_two(B, h0, h1);
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
