#!/usr/bin/env python3
"""profiles/<tag>_{fetch,write}.csv (tools/rocpd_pmc.py output of the FETCH_SIZE / WRITE_SIZE passes of
tools/profile_bench.sh) -> profiles/r01_traffic.json, the HBM-traffic figures bench.py quotes.
Corrections per MI355X_MICROARCH.md: counters are KB; FETCH_SIZE is doubled for wide coalesced reads."""
import csv, json, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "profiles/r01_single"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01_traffic.json"


def load(path, col):
    d = {}
    for row in csv.DictReader(open(path)):
        d[row["Kernel"]] = (int(row["Dispatches"]), float(row[col]) * 1024.0)
    return d


f, w = load(tag + "_fetch.csv", "FETCH_SIZE"), load(tag + "_write.csv", "WRITE_SIZE")
kernels = {}
for k in f:
    if "jfgpu::p1_" in k or "jfgpu::p2_" in k or "tile_insert" in k or "granule" in k:
        kernels[k] = {"dispatches": f[k][0], "fetch_bytes": 2.0 * f[k][1], "write_bytes": w.get(k, (0, 0.0))[1]}


def tot(sub):
    return sum(v["fetch_bytes"] + v["write_bytes"] for k, v in kernels.items() if sub in k)


n_batches = max(v["dispatches"] for k, v in kernels.items() if "jfgpu::p1_" in k)      # 1 warm-up + 10 timed, equal size
res = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python bench.py --no-cpu-baseline` "
              "(tools/profile_bench.sh), sums over all dispatches of the run: 1 warm-up batch + 10 timed batches, flush kernels: "
              "1 Gbp warm-up flush + 10 Gbp timed flush; KB -> bytes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes; "
              "WRITE_SIZE as reported",
    "kernels": kernels,
    "per_timed_launch_bytes": {
        "p1_partition": tot("jfgpu::p1_") / n_batches,
        "p2_partition": tot("jfgpu::p2_") * 10.0 / 11.0,
        # tile insert: table writes are the same in both flushes, item reads scale 1:10
        "tile_insert": sum(v["write_bytes"] / 2.0 + v["fetch_bytes"] * 10.0 / 11.0 for k, v in kernels.items() if "tile_insert" in k),
    },
    "per_timed_launch_note": "p1: per batch (all P1 kernels / batches); p2: 10/11 of the two flushes; tile_insert: half of the table "
                             "writes + 10/11 of the item reads",
}
res["whole_run_timed_bytes"] = res["per_timed_launch_bytes"]["p1_partition"] * 10 + res["per_timed_launch_bytes"]["p2_partition"] + \
    res["per_timed_launch_bytes"]["tile_insert"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["per_timed_launch_bytes"]), res["whole_run_timed_bytes"] / 1e9, "GB")
