#!/bin/bash
# Final bench lines of C3 and C5 (after profiles/r02_traffic_<cfg>.json were regenerated from call 45's PMC passes).
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
{
  for c in C3 C5; do
    echo "== bench $c ($(date +%T))"
    timeout 1200 python bench.py --config $c 2> gpurun_out/r02h_bench_$c.err | grep '^{' > gpurun_out/r02h_bench_$c.json
    cut -c1-300 gpurun_out/r02h_bench_$c.json
  done
} > gpurun_out/r02_call46.log 2>&1
cat gpurun_out/r02_call46.log | cut -c1-300
