#!/usr/bin/env python3
"""Rate of the direct (global-atomic) path for keys of three and four words -- k = 100, the reference's tests/large_key.sh
length -- on 2 Gbp of 150 bp reads into 2^31 slots of 32 bytes (64 GB), next to k = 63 through the partitioned path at the
same input.  The numbers DESIGN.md quotes for the row that has no partitioned path.  usage: python tools/k100_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jellyfish_amd import capi
L, n_reads = 150, 13_333_333
for k, lsize in ((100, 30), (63, 31)):
    with capi.Table(k, 1 << lsize, canonical=True) as t:
        buf = t.malloc(n_reads * (L + 1) + 16)
        t.gen_reads_dev(buf, 0, n_reads, L, 42)
        t.reserve(n_reads * (L + 1))
        t.sync()
        for rep in range(2):
            t.clear()
            t0 = time.time()
            for i in range(10):
                a, b = n_reads * i // 10, n_reads * (i + 1) // 10
                t.count_ascii_dev(buf + a * (L + 1), (b - a) * (L + 1))
            t.sync()
            dt = time.time() - t0
        st = t.stats()
        print("k", k, "slot bytes", t.info.slot_bytes, "slots 2^%d" % t.info.lsize, "total", st.total, "distinct", st.distinct, "%.1f ms" % (dt * 1e3), "%.2f G k-mers/s" % (st.total / dt / 1e9), flush=True)
        t.free(buf)
