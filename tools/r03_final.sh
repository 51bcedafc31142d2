#!/bin/bash
# Round-3 measurement call (one gpurun): the whole -m gpu suite, the rocprofv3 kernel-stats and PMC passes of the C2 / C5 / C3
# bench commands (tools/profile_bench.sh), the traffic files bench.py quotes, then the bench lines (default = the contract
# line with its secondaries; distribution G; the sharded code path on one GPU).  Everything lands in gpurun_out/r03f/.
#   usage: tools/r03_final.sh [suite|profiles|bench ...]   (default: all three)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/r03f; mkdir -p $O
what=${*:-suite profiles bench}
for w in $what; do case $w in
suite)
  (time timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|note:" | tail -12) > $O/suite.txt 2>&1
  tail -8 $O/suite.txt ;;
profiles)
  for c in C2 C5 C3; do
    timeout 600 tools/profile_bench.sh r03f/$c --config $c > $O/${c}_profile.log 2>&1
    L=34; [ $c != C2 ] && L=33
    python tools/make_traffic_json.py $O/$c $c 10 $L $O/traffic_$c.json >> $O/${c}_profile.log 2>&1
    head -4 $O/${c}_stats.csv | cut -c1-200
  done ;;
bench)
  timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
  timeout 300 python bench.py --dist G --no-cpu-baseline --no-extras > $O/bench_distG.json 2> $O/bench_distG.err; echo "G rc=$?"
  JFGPU_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_forced_dist.json 2> $O/bench_forced_dist.err; echo "forced-dist rc=$?"
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --repeats 1 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "steps20 rc=$?"
  JFGPU_BENCH_RANK_TIMEOUT=200 timeout 260 python bench.py --gpus 2 --no-extras > $O/bench_gpus2_shared.json 2> $O/bench_gpus2_shared.err; echo "gpus2 (one device) rc=$?"
  for f in bench_default bench_distG bench_forced_dist bench_steps20 bench_gpus2_shared; do tail -1 $O/$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f', d['value'], {k:v['ms'] for k,v in d['kernels'].items()}, 'whole_path_frac', d['roofline']['whole_path_frac'])
for c,v in d.get('secondary',{}).items(): print('  ', c, v.get('value'), v.get('error'))
"; done ;;
esac; done
