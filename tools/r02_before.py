"""Round-2 starting point for BASELINE configs 3 and 5 with the round-1 kernels (global atomics):
k=31 Bloom pass + filtered count, and k=63 count, on 10 Gbp of device-generated 150 bp reads.
Run on the GPU box; prints one JSON object.  Not part of the test suite."""
import json
import sys
import time

sys.path.insert(0, ".")
from jellyfish_amd import capi

L = 150
gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
n_reads = int(round(gbp * 1e9 / L))
nbytes = n_reads * (L + 1)
res = {"gbp": gbp, "reads": n_reads}


def timed(fn, sync):
    sync()
    t0 = time.perf_counter()
    fn()
    sync()
    return time.perf_counter() - t0


# ---- C3: bc -m 31 -s 10G -f 0.001 -C, then count -m 31 -C --bc ----
k = 31
with capi.Table(k, 1 << 33) as t:
    d = t.malloc(nbytes + 16)
    t.gen_reads_dev(d, 0, n_reads, L, 42)
    t.sync()
    m = capi.opt_m(0.001, int(gbp * 1e9))
    with capi.Bloom(k, m, capi.opt_k(0.001)) as b:
        kmers = n_reads * (L - k + 1)
        dt = timed(lambda: b.insert_ascii_dev(d, nbytes), b.sync)
        assert b.sync() == kmers
        res["c3_bc_pass_s"] = dt
        res["c3_bc_pass_Gkmers_s"] = kmers / dt / 1e9
        res["c3_m"] = m
        print("C3 bc pass", dt, "s", kmers / dt / 1e9, "G k-mers/s", flush=True)
        t.attach_bloom(b)
        dt = timed(lambda: t.count_ascii_dev(d, nbytes), t.sync)
        st = t.stats()
        res["c3_count_pass_s"] = dt
        res["c3_count_pass_Gkmers_s"] = kmers / dt / 1e9
        res["c3_admitted"] = st.total
        print("C3 count --bc pass", dt, "s", kmers / dt / 1e9, "G k-mers/s; admitted", st.total, "distinct", st.distinct, flush=True)
        t.attach_bloom(None)
    t.free(d)

# ---- C5: count -m 63 -C -s 8G ----
k = 63
with capi.Table(k, 1 << 33) as t:
    d = t.malloc(nbytes + 16)
    t.gen_reads_dev(d, 0, n_reads, L, 42)
    t.sync()
    kmers = n_reads * (L - k + 1)
    dt = timed(lambda: t.count_ascii_dev(d, nbytes), t.sync)
    st = t.stats()
    assert st.total == kmers, (st.total, kmers)
    res["c5_count_s"] = dt
    res["c5_Gkmers_s"] = kmers / dt / 1e9
    res["c5_distinct"] = st.distinct
    print("C5 count", dt, "s", kmers / dt / 1e9, "G k-mers/s; distinct", st.distinct, flush=True)
    t.free(d)
print(json.dumps(res))
