"""Measurements on the GPU box (not part of the test suite): random-access roofline probes
(R_gups) at several table sizes and the count kernel's throughput on 1 Gbp."""
import json
import sys
import time

sys.path.insert(0, ".")
from jellyfish_amd import capi

res = {}
k, L = 21, 150
for lsize in (31, 34):
    with capi.Table(k, 1 << lsize) as t:
        n = 1 << 28
        for mode, name in ((0, "add_noret"), (1, "add_ret"), (2, "cas"), (3, "load+add")):
            ups = t.gups(n, mode)
            res[f"gups_{name}_2^{lsize}"] = round(ups / 1e9, 3)
            print(lsize, name, round(ups / 1e9, 3), "G updates/s", flush=True)
        t.clear()
        n_reads = 6_666_667
        nbytes = n_reads * (L + 1)
        d = t.malloc(nbytes + 16)
        t.gen_reads_dev(d, 0, n_reads, L, 42)
        t.sync()
        t.profile_enable(True)
        t.count_ascii_dev(d, nbytes)
        t.sync()
        ms, launches, units = t.profile_get(0)
        kmers = n_reads * (L - k + 1) * launches
        res[f"count_Gkmers_s_2^{lsize}"] = round(kmers / (ms * 1e-3) / 1e9, 3)
        print(lsize, "count", kmers, "kmers in", ms, "ms ->", kmers / ms / 1e6, "G kmers/s", flush=True)
        st = t.stats()
        print("stats", st.distinct, st.total, st.max_count, flush=True)
        t.free(d)
print(json.dumps(res))
