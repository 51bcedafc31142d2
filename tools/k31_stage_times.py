#!/usr/bin/env python3
"""Stage times of a plain k = 31 count (64-bit items, 8-byte slots: the other instantiation of the one-word kernels) at
5 Gbp into 2^33 slots, next to k = 21 at the same size.  usage: python tools/k31_stage_times.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jellyfish_amd import capi
L, n_reads = 150, 33_333_333
for k in (31, 21):
    with capi.Table(k, 1 << 33, canonical=True) as t:
        buf = t.malloc(n_reads * (L + 1) + 16)
        t.gen_reads_dev(buf, 0, n_reads, L, 42)
        t.reserve(n_reads * (L + 1))
        t.sync()
        for rep in range(2):
            t.clear()
            t.profile_enable(True); t.profile_reset()
            t0 = time.time()
            for i in range(10):
                a, b = n_reads * i // 10, n_reads * (i + 1) // 10
                t.count_ascii_dev(buf + a * (L + 1), (b - a) * (L + 1))
            t.sync()
            dt = time.time() - t0
        st = t.stats()
        print("k", k, "slot bytes", t.info.slot_bytes, "total", st.total, "distinct", st.distinct, "%.1f ms" % (dt * 1e3), "%.1f G k-mers/s" % (st.total / dt / 1e9),
              {name: round(t.profile_get(i)[0], 1) for i, name in enumerate(("count_direct", "add_keys", "shard_partition", "lookup", "p1_partition", "p2_partition", "tile_insert", "items_direct")) if t.profile_get(i)[1]}, flush=True)
        t.free(buf)
