#!/bin/bash
# Collect the round's evidence for bench.py on the GPU box: per-kernel times (kernel trace) and HBM
# traffic (FETCH_SIZE / WRITE_SIZE in separate PMC passes, as the microarch guide prescribes).
# usage: tools/profile_bench.sh <tag> [bench args...]   -> gpurun_out/<tag>_{bench.json,stats.csv,fetch.csv,write.csv}
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --warmup 0 --repeats 1 $*"
O=$R/gpurun_out
rm -rf $O/${TAG}_p1 $O/${TAG}_p2 $O/${TAG}_p3
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_p1 -o p -- $B 2>/dev/null | grep '^{' | tail -1 > $O/${TAG}_bench.json
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_p2 -o p -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_p3 -o p -- $B > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py $O/${TAG}_p1 > $O/${TAG}_stats.csv
python tools/rocpd_pmc.py $O/${TAG}_p2 > $O/${TAG}_fetch.csv
python tools/rocpd_pmc.py $O/${TAG}_p3 > $O/${TAG}_write.csv
rm -rf $O/${TAG}_p1 $O/${TAG}_p2 $O/${TAG}_p3
cut -c1-160 $O/${TAG}_bench.json
