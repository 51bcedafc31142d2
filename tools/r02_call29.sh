#!/bin/bash
# Round 2, final bench lines (after tools/r02_call28.sh's PMC passes became profiles/r02_traffic_<cfg>.json): C2 / C3 / C5 and
# the sharded code path on one GPU.
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
{
  for c in C2 C3 C5; do
    echo "== bench $c ($(date +%T))"
    timeout 1200 python bench.py --config $c 2> gpurun_out/r02f_bench_$c.err | grep '^{' > gpurun_out/r02f_bench_$c.json
    cut -c1-400 gpurun_out/r02f_bench_$c.json
  done
  echo "== bench C2 through the sharded code path ($(date +%T))"
  JFGPU_BENCH_FORCE_DIST=1 timeout 900 python bench.py --config C2 --no-cpu-baseline --no-extras --repeats 3 2> gpurun_out/r02f_fd.err | grep '^{' > gpurun_out/r02f_bench_C2_forced_dist.json
  cut -c1-700 gpurun_out/r02f_bench_C2_forced_dist.json
  echo "== done ($(date +%T))"
} > gpurun_out/r02_call29.log 2>&1
tail -20 gpurun_out/r02_call29.log | cut -c1-300
