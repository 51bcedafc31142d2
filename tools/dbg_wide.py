import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from jellyfish_amd import capi
what, k = sys.argv[1], int(sys.argv[2])
def show(t, tag):
    st = t.stats(); recs = t.dump_records(); keys, cnts = capi.decode_records(recs, k, 4)
    print(tag, "distinct", st.distinct, "total", st.total, "counts", sorted(cnts.tolist(), reverse=True)[:6], flush=True)
with capi.Table(k, 1 << 16, canonical=True) as t:
    print("lsize", t.info.lsize, "val_len", t.info.val_len, flush=True)
    if what == "keys":
        for key in ([0, 0], [0x123456789, 0x3FF]):
            for rep in (2, 64, 300):
                t.clear(); t0 = time.time()
                t.add_keys(np.repeat(np.array([key], dtype=np.uint64), rep, axis=0), val=1)
                show(t, f"add_keys key={key} x{rep} ({time.time()-t0:.2f}s)")
    else:
        for n in (k + 63, 200):
            t.clear(); t0 = time.time(); t.count_ascii(b"A" * n); t.sync(); show(t, f"polyA n={n} ({time.time()-t0:.2f}s)")
