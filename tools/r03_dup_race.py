#!/usr/bin/env python3
"""Heavy-duplication determinism check (round 3): reads sampled from a small random genome at ~100x coverage, counted
several times through the partitioned path and once through the global-atomic path; every digest must be the same."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jellyfish_amd import capi
k, lsize = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 26
genome, n_reads = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000, int(sys.argv[4]) if len(sys.argv) > 4 else 6_000_000
res = {}
for mode in (1, 2, 2, 2, 2):
    with capi.Table(k, 1 << lsize, canonical=True) as t:
        t.set_mode(mode)
        buf = t.malloc(n_reads * 151 + 16)
        t.gen_genome_reads_dev(buf, 0, n_reads, 150, genome, 0.01, 42)
        t.sync()
        steps = 4
        for i in range(steps):
            a, b = n_reads * i // steps, n_reads * (i + 1) // steps
            t.count_ascii_dev(buf + a * 151, (b - a) * 151)
        t.sync()
        d = t.digest(); st = t.stats()
        print("mode", mode, d, st.distinct, st.total, "slot bytes", t.info.slot_bytes, flush=True)
        res.setdefault(mode, []).append(d)
        t.free(buf)
ok = len(set(res[2])) == 1 and res[2][0] == res[1][0]
print("DETERMINISTIC AND EQUAL TO THE DIRECT PATH" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
