"""Driver for experimental builds of the engine (JFGPU_LIB=/path/to/lib.so): the 10 Gbp bench flow without the
bench harness, printing the per-stage HIP-event times.  Used with -DJFGPU_TILE_PROF builds (phase clocks of the
tile insert go to stderr at every sync)."""
import sys, os, json
sys.path.insert(0, ".")
from jellyfish_amd import capi
k, L = 21, 150
n_reads = 66_666_667
with capi.Table(k, 1 << 34) as t:
    nbytes = n_reads * (L + 1)
    d = t.malloc(nbytes + 16)
    t.gen_reads_dev(d, 0, n_reads, L, 42)
    t.reserve(nbytes)
    t.sync()
    for rep in range(2):
        t.clear()
        t.profile_enable(True); t.profile_reset()
        step = nbytes // 10 // 151 * 151
        for i in range(10):
            t.count_ascii_dev(d + i * step, step if i < 9 else nbytes - 9 * step)
        try:
            t.sync()
        except Exception as e:
            print("sync:", e)
    print(os.environ.get("JFGPU_ABLATE"), {i: round(t.profile_get(i)[0], 2) for i in (4, 5, 6)})
