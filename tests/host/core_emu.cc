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

// usage: core_emu K CANONICAL LSIZE SHARD_BITS LEAD < sequence   (LEAD = misalignment 0..15)
int main(int argc, char** argv) {
  if(argc < 6) return 2;
  const uint32_t k = atoi(argv[1]), canonical = atoi(argv[2]), lsize = atoi(argv[3]), shard_bits = atoi(argv[4]);
  const int lead = atoi(argv[5]);
  std::string seq; { char buf[65536]; size_t n; while((n = fread(buf, 1, sizeof buf, stdin)) > 0) seq.append(buf, n); }
  TableGeom g;
  if(!geom_init(g, k, lsize, shard_bits, 0, canonical)) { fprintf(stderr, "bad geometry\n"); return 3; }
  Gf2Matrix m = gf2_random(lsize, 2 * k, 12345);
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
