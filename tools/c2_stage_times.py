#!/usr/bin/env python3
"""Stage times of config 2's job (k = 21, 10 Gbp, 2^34 slots) without bench.py's checks: for ablation builds whose
results are wrong on purpose.  usage: [JFGPU_LIB=...] python tools/c2_stage_times.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jellyfish_amd import capi
L, n = 150, 66_666_666
with capi.Table(21, 1 << 34, canonical=True) as t:
    buf = t.malloc(n * (L + 1) + 16)
    t.gen_reads_dev(buf, 0, n, L, 42)
    t.reserve(n * (L + 1))
    t.sync()
    for rep in range(2):
        t.clear()
        t.profile_enable(True); t.profile_reset()
        t0 = time.time()
        for i in range(10):
            a, b = n * i // 10, n * (i + 1) // 10
            t.count_ascii_dev(buf + a * (L + 1), (b - a) * (L + 1))
        t.sync()
        dt = time.time() - t0
    print("k 21 %.1f ms" % (dt * 1e3), {nm: round(t.profile_get(i)[0], 1) for i, nm in ((4, "p1"), (5, "p2"), (6, "tile"))}, "total counted", t.stats().total, flush=True)
