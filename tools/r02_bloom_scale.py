"""Config 3 at scale on the GPU box: bc pass (k=31, m = 14e10 cells = 28 GB, 10 hashes) over 10 Gbp of device-generated
reads with the direct kernel and with the partitioned path; the two byte arrays must be identical.  Then count --bc.
Prints one JSON object.  Not part of the test suite."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from jellyfish_amd import capi

L, k = 150, 31
gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
compare = len(sys.argv) > 2 and sys.argv[2] == "compare"
n_reads = int(round(gbp * 1e9 / L))
nbytes = n_reads * (L + 1)
kmers = n_reads * (L - k + 1)
res = {"gbp": gbp, "reads": n_reads, "kmers": kmers}
m = capi.opt_m(0.001, int(gbp * 1e9))
with capi.Table(k, 1 << 20) as helper:                   # only for device memory + the read generator
    d = helper.malloc(nbytes + 16)
    helper.gen_reads_dev(d, 0, n_reads, L, 42)
    helper.sync()
    bodies = {}
    for mode, name in ((2, "partitioned"), (1, "direct")):
        with capi.Bloom(k, m, capi.opt_k(0.001)) as b:
            b.set_mode(mode)
            if mode == 2:
                b.reserve(128 << 30)
            b.sync()
            b.profile_enable(True)
            steps = 10
            t0 = time.perf_counter()
            for i in range(steps):
                lo, hi = n_reads * i // steps, n_reads * (i + 1) // steps
                b.insert_ascii_dev(d + lo * (L + 1), (hi - lo) * (L + 1))
            fed = b.sync()
            dt = time.perf_counter() - t0
            assert fed == kmers, (fed, kmers)
            prof = {nm: b.profile_get(i) for i, nm in enumerate(("direct", "p1_route", "p2_partition", "segments"))}
            res["bc_" + name] = {"s": dt, "Gkmers_s": kmers / dt / 1e9, "stages_ms_launches_units": prof}
            print(name, dt, "s", kmers / dt / 1e9, "G k-mers/s", prof, flush=True)
            if compare:
                bodies[mode] = b.read()
            if mode == 2:                                  # count --bc on the partitioned filter
                with capi.Table(k, 1 << 33) as t:
                    t.attach_bloom(b)
                    t.sync()
                    t0 = time.perf_counter()
                    t.count_ascii_dev(d, nbytes)
                    t.sync()
                    dt = time.perf_counter() - t0
                    st = t.stats()
                    res["count_bc"] = {"s": dt, "Gkmers_s": kmers / dt / 1e9, "admitted": st.total, "distinct": st.distinct}
                    print("count --bc", dt, "s", kmers / dt / 1e9, "G k-mers/s admitted", st.total, flush=True)
                    t.attach_bloom(None)
    if compare:
        res["bodies_equal"] = bool(np.array_equal(bodies[1], bodies[2]))
        print("bodies equal:", res["bodies_equal"], flush=True)
    helper.free(d)
print(json.dumps(res))
