// tests/host/core_emu.cc -- TEST-ONLY host emulation of the device tile walk.
// Compiles jellyfish_amd/csrc/kmer_core.hpp with g++ and replays exactly what a
// 256-lane block does per tile (stage -> per-lane for_each_kmer), printing every
// k-mer plus its table address pieces.  tests/test_core_host.py compares the
// output with the oracle.  This is NOT a CPU fallback of the product: it lives
// under tests/ and nothing in jellyfish_amd/ links it.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "../../jellyfish_amd/csrc/kmer_core.hpp"
#include "../../jellyfish_amd/csrc/gf2_matrix.hpp"
using namespace jfgpu;

// usage: core_emu K CANONICAL LSIZE SHARD_BITS LEAD [xs] < sequence   (LEAD = misalignment 0..15; xs: the xor-shift matrix
// family -- the position is then also computed the way the partition kernel does, in registers, and must equal the tables')
//        core_emu xs-family      every (k <= 32, r < 2k): low block invertible, matrix product == xs_hash == its two-dword form
static int xs_family() {
  uint64_t s = 99;
  for(uint32_t k = 1; k <= 32; ++k) for(uint32_t r = 1; r < 2 * k && r < 64; ++r) {
    const Gf2Matrix m = gf2_xorshift_matrix(r, 2 * k);
    std::vector<uint64_t> binv;
    if(!gf2_invert_low_block(m, binv)) { printf("singular low block: k %u r %u\n", k, r); return 1; }
    if(!gf2_is_xorshift(m)) { printf("family test fails: k %u r %u\n", k, r); return 1; }
    const uint64_t km = 2 * k == 64 ? ~0ull : ((1ull << (2 * k)) - 1);
    for(int i = 0; i < 300; ++i) {
      const uint64_t key = splitmix64(s) & km, a = m.times(key), b = xs_hash(key, r, 2 * k);
      if(a != b) { printf("matrix product != xs_hash: k %u r %u\n", k, r); return 1; }
      if(r > 32) {
        uint32_t lo, hi;
        if(xs_folds(2 * k)) xs_hash_halves<true>((uint32_t)key, (uint32_t)(key >> 32), (1u << (r - 32)) - 1u, lo, hi);
        else xs_hash_halves<false>((uint32_t)key, (uint32_t)(key >> 32), (1u << (r - 32)) - 1u, lo, hi);
        if((((uint64_t)hi << 32) | lo) != b) { printf("two-dword form differs: k %u r %u\n", k, r); return 1; }
      }
    }
  }
  // two-word keys: the matrix product over both words == xs_hash_wide
  for(uint32_t k = 33; k <= 64; ++k) for(uint32_t r : {13u, 20u, 31u, 32u, 33u, 40u, 48u, 63u}) {
    const uint32_t c = 2 * k;
    const Gf2Matrix m = gf2_xorshift_matrix(r, c);
    std::vector<uint64_t> binv;
    if(!gf2_invert_low_block(m, binv) || !gf2_is_xorshift(m)) { printf("wide family fails: k %u r %u\n", k, r); return 1; }
    const uint64_t hm = c == 128 ? ~0ull : ((1ull << (c - 64)) - 1);
    for(int i = 0; i < 100; ++i) {
      const uint64_t lo = splitmix64(s), hi = splitmix64(s) & hm;
      uint64_t a = 0;
      for(uint32_t j = 0; j < c; ++j) if(((j < 64 ? lo >> j : hi >> (j - 64)) & 1ull)) a ^= m.col_for_bit(j);
      if(a != xs_hash_wide(lo, hi, r)) { printf("wide matrix product != xs_hash_wide: k %u r %u\n", k, r); return 1; }
    }
  }
  // a few values for the python restatement (jellyfish_amd/capi.py: xs_hash): r, key bits, key, position
  for(uint32_t r : {5u, 22u, 32u, 33u, 34u, 41u, 63u}) for(uint32_t c : {42u, 64u}) for(int i = 0; i < 4; ++i) {
    const uint64_t key = splitmix64(s) & (c == 64 ? ~0ull : ((1ull << c) - 1));
    printf("v %u %u %llu %llu\n", r, c, (unsigned long long)key, (unsigned long long)xs_hash(key, r, c));
  }
  printf("ok\n");
  return 0;
}

int main(int argc, char** argv) {
  if(argc == 2 && !strcmp(argv[1], "xs-family")) return xs_family();
  if(argc < 6) return 2;
  const bool xs = argc > 6 && !strcmp(argv[6], "xs");
  const uint32_t k = atoi(argv[1]), canonical = atoi(argv[2]), lsize = atoi(argv[3]), shard_bits = atoi(argv[4]);
  const int lead = atoi(argv[5]);
  std::string seq; { char buf[65536]; size_t n; while((n = fread(buf, 1, sizeof buf, stdin)) > 0) seq.append(buf, n); }
  TableGeom g;
  if(!geom_init(g, k, lsize, shard_bits, 0, canonical)) { fprintf(stderr, "bad geometry\n"); return 3; }
  Gf2Matrix m = xs ? gf2_xorshift_matrix(lsize, 2 * k) : gf2_random(lsize, 2 * k, 12345);
  std::vector<uint64_t> fwd, inv;
  if(!gf2_build_tables(m, fwd, inv)) return 4;
  // emulate an aligned allocation with `lead` bytes of junk in front
  std::vector<uint8_t> mem(seq.size() + 64 + 4096, 'A');
  uint8_t* base = mem.data();
  while(((uintptr_t)base & 15) != 0) ++base;
  memcpy(base + lead, seq.data(), seq.size());
  const int64_t lo = lead, hi = lead + (int64_t)seq.size();
  const int kBlock = 256, kTile = kBlock * kPerLane;
  printf("matrix");
  for(uint32_t i = 0; i < m.c; ++i) printf(" %llu", (unsigned long long)m.columns[i]);
  printf("\n");
  std::vector<uint32_t> s_codes(kBlock + 2), s_inv(kBlock + 2);
  for(int64_t tile = 0; tile * kTile < hi; ++tile) {
    const int64_t ts = tile * kTile;
    for(int tid = 0; tid < kBlock; ++tid) load_pack16(base, ts + 16 * tid, lo, hi, s_codes[tid + 2], s_inv[tid + 2]);
    for(int tid = 0; tid < 2; ++tid) load_pack16(base, ts - 32 + 16 * tid, lo, hi, s_codes[tid], s_inv[tid]);
    for(int tid = 0; tid < kBlock; ++tid) {
      LaneWords L;
      L.cur = s_codes[tid + 2]; L.p1 = s_codes[tid + 1]; L.p2 = s_codes[tid];
      L.inv48 = ((uint64_t)s_inv[tid] << 32) | ((uint64_t)s_inv[tid + 1] << 16) | s_inv[tid + 2];
      for_each_kmer(g, L, [&](int, uint64_t key) {
        const uint64_t pos = hash_tables(fwd.data(), key, g.nbytes);
        if(xs && lsize < 2 * k && xs_hash(key, lsize, 2 * k) != pos) { fprintf(stderr, "xs_hash != table hash\n"); exit(5); }
        const SlotAddr a = slot_addr(g, pos);
        const uint64_t tag = make_tag(g, key, a.idx0);
        const uint64_t word = (1ull << (g.tag_bits + 1)) | g.occ_bit | tag;
        TableGeom gs = g; gs.shard_id = a.shard;
        const uint64_t back = slot_key(gs, inv.data(), word, a.tile_base);
        printf("%llu %llu %llu\n", (unsigned long long)key, (unsigned long long)pos, (unsigned long long)back);
      });
    }
  }
  return 0;
}
