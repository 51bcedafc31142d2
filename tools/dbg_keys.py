import sys
sys.path.insert(0, ".")
import numpy as np
from jellyfish_amd import capi
k, L, n_reads = 21, 150, 2_000_000
nbytes = n_reads * (L + 1)
with capi.Table(k, 1 << 30) as t:
    d = t.malloc(nbytes + 16)
    t.gen_reads_dev(d, 0, n_reads, L, 7)
    t.set_mode(1)
    t.count_ascii_dev(d, nbytes); t.sync()
    s0 = t.stats(); print("direct", s0.distinct, s0.total)
    t.clear()
    cap = n_reads * (L - k + 1)
    dk = t.malloc(cap * 8)
    counts = t.partition_ascii_dev(d, nbytes, dk, cap)
    print("partition counts", counts)
    keys = t.d2h(dk, cap * 8).view(np.uint64)
    print("unique keys from partition:", len(np.unique(keys)))
    for mode in (1, 2):
        t.clear(); t.set_mode(mode)
        t.add_keys_dev(dk, cap, 1); t.sync()
        s = t.stats(); print("add_keys mode", mode, s.distinct, s.total)
    # two calls, partitioned
    t.clear(); t.set_mode(2)
    h = cap // 2
    t.add_keys_dev(dk, h, 1); t.wait(); t.add_keys_dev(dk + 8 * h, cap - h, 1); t.sync()
    s = t.stats(); print("add_keys 2 calls partitioned", s.distinct, s.total)
