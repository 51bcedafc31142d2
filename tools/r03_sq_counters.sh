#!/bin/bash
# SQ counters of the C2 bench command, per kernel (one rocprofv3 --pmc pass, kernel trace only): where the wave cycles of
# P1 / P2 / T go -- issuing (VALU, LDS, ...), parked on s_waitcnt / barriers, stalled on issue.
#   usage: tools/r03_sq_counters.sh <tag> [bench args]  -> gpurun_out/r03/<tag>_sq.csv
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r03; mkdir -p $O
rm -rf $O/${TAG}_sqp
timeout 280 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  -d $O/${TAG}_sqp -o p -- python $R/bench.py --no-cpu-baseline --no-extras --warmup 0 --repeats 1 "$@" > /dev/null 2> $O/${TAG}_sq.err
cd $R
python tools/rocpd_pmc.py $O/${TAG}_sqp > $O/${TAG}_sq.csv
rm -rf $O/${TAG}_sqp
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/${TAG}_sq.csv")))
for r in rows:
    if not any(s in r["Kernel"] for s in ("p1_ring", "p2_granule", "tile_rank")): continue
    wc = float(r.get("SQ_WAVE_CYCLES") or 0) or 1
    print(r["Kernel"][:60], "dispatches", r["Dispatches"])
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"):
        v = r.get(c)
        if v not in (None, ""): print("   %-22s %6.1f %% of wave cycles" % (c, 100 * float(v) / wc))
PY
