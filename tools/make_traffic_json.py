#!/usr/bin/env python3
"""<tag>_{fetch,write}.csv (tools/rocpd_pmc.py output of the FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh)
-> profiles/r05_traffic_<cfg>.json: HBM bytes per job of every stage bench.py names, the figures its `roofline.traffic`
quotes.  The profiled command runs exactly one job (--warmup 0 --repeats 1), so sums over the run are per-job sums.
Corrections per MI355X_MICROARCH.md: counters are KB; FETCH_SIZE is doubled for wide coalesced reads; WRITE_SIZE as is.
usage: make_traffic_json.py <tag path prefix> <C2|C3|C5> <gbp> <lsize> [out.json]"""
import csv
import json
import sys

tag, cfg, gbp, lsize = sys.argv[1], sys.argv[2], float(sys.argv[3]), int(sys.argv[4])
out = sys.argv[5] if len(sys.argv) > 5 else "profiles/r06_traffic_%s.json" % cfg


def load(path, col):
    d = {}
    for row in csv.DictReader(open(path)):
        d[row["Kernel"]] = (int(row["Dispatches"]), float(row[col] or 0) * 1024.0)
    return d


def stage_of(kernel):
    k = kernel
    if "p1_bloom" in k: return "bc_p1_route"
    if "bloom_segment" in k: return "bc_segments"
    if "bloom_insert" in k or "bloom_items" in k: return "bc_direct"
    if "tile_insert" in k or "tile_rank_insert" in k: return "tile_insert"
    if "items_direct" in k: return "items_direct"
    if "p1_stragglers" in k and ("P2RingDirect" in k or "BloomRingDirect" in k): return "bc_p2_partition" if cfg == "C3" else "p2_partition"   # P2's lists
    if "jfgpu::p2_" in k or "scan_matrix" in k: return "bc_p2_partition" if cfg == "C3" else "p2_partition"
    if "jfgpu::p1_" in k or "granule_finish" in k: return "p1_partition"
    if "count_ascii" in k: return "count_direct"
    return None


f, w = load(tag + "_fetch.csv", "FETCH_SIZE"), load(tag + "_write.csv", "WRITE_SIZE")
kernels, stages = {}, {}
for k in f:
    st = stage_of(k)
    if st is None:
        continue
    rec = {"stage": st, "dispatches": f[k][0], "fetch_bytes": 2.0 * f[k][1], "write_bytes": w.get(k, (0, 0.0))[1]}
    kernels[k] = rec
    stages[st] = stages.get(st, 0.0) + rec["fetch_bytes"] + rec["write_bytes"]
import os
import subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
try:
    commit = subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=bench.ROOT, text=True).strip()
except Exception:
    commit = None
res = {
    "config": cfg, "gbp": gbp, "lsize": lsize,
    # bench.py quotes these figures only while the engine's sources are the ones they were measured on
    "kernel_sources_sha256": bench.kernel_sources_sha(), "commit_when_made": commit,
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python bench.py --config %s --warmup 0 --repeats 1 "
              "--no-extras --no-cpu-baseline` (tools/profile_bench.sh): one job, sums over all its dispatches; KB -> bytes; FETCH_SIZE doubled as "
              "MI355X_MICROARCH.md prescribes for wide coalesced reads; WRITE_SIZE as reported" % cfg,
    "kernels": kernels,
    "per_job_bytes": stages,
    "whole_job_bytes": sum(stages.values()),
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(stages), res["whole_job_bytes"] / 1e9, "GB")
