#!/bin/bash
# SQ counters of a bench.py command, per kernel (rocprofv3 --pmc passes with kernel trace only): where the wave cycles go
# (issuing VALU / LDS, parked on s_waitcnt and barriers) and what the LDS array does (instructions, index-active cycles,
# bank / address conflict cycles), normalised by the CU-busy cycles.
#   usage: tools/sq_counters.sh <tag> [ENV=.. --] [bench args]  -> gpurun_out/<tag>_sq{1,2}.csv + a readable summary
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-extras --warmup 0 --repeats 1 $*"
pass() {  # pass <n> <counters...>
  n=$1; shift
  rm -rf $O/${TAG}_sqp$n
  timeout 280 rocprofv3 --kernel-trace --pmc "$@" -d $O/${TAG}_sqp$n -o p -- $B > /dev/null 2> $O/${TAG}_sq$n.err
  (cd $R && python tools/rocpd_pmc.py $O/${TAG}_sqp$n > $O/${TAG}_sq$n.csv)
  rm -rf $O/${TAG}_sqp$n
}
pass 1 SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
pass 2 SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU
cd $R
python - "$O/${TAG}" <<'PY'
import csv, sys
tag = sys.argv[1]
for n in (1, 2):
    try:
        rows = list(csv.DictReader(open("%s_sq%d.csv" % (tag, n))))
    except Exception as e:
        print("pass", n, "failed:", e); continue
    for r in rows:
        if not any(s in r["Kernel"] for s in ("p1_", "p2_granule", "p2_ring", "tile_rank", "tile_insert", "bloom_seg", "wide")): continue
        base = "SQ_WAVE_CYCLES" if n == 1 else "SQ_BUSY_CU_CYCLES"
        wc = float(r.get(base) or 0) or 1
        print("[sq%d] %s  dispatches %s  (%% of %s)" % (n, r["Kernel"][:70], r["Dispatches"], base))
        for c, v in r.items():
            if c in ("Kernel", "Dispatches") or v in (None, ""): continue
            print("      %-24s %14.0f  %6.1f %%" % (c, float(v), 100 * float(v) / wc))
PY
