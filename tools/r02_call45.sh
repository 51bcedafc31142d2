#!/bin/bash
# C3 and C5 after the single-pass P2 / shared-buffer flushes: rocprofv3 passes + bench lines; the whole GPU suite.
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
{
  echo "== GPU suite ($(date +%T))"
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
  for c in C3 C5; do
    echo "== rocprofv3 passes $c ($(date +%T))"
    bash tools/profile_bench.sh r02h_$c --config $c
    echo
  done
  echo "== done ($(date +%T))"
} > gpurun_out/r02_call45.log 2>&1
cat gpurun_out/r02_call45.log | cut -c1-300
