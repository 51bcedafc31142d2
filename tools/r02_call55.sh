#!/bin/bash
# C2 once more on the last engine of the round (kernel names in the rocprofv3 files = the sources'): PMC passes.
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
{
  echo "== rocprofv3 passes C2 ($(date +%T))"
  bash tools/profile_bench.sh r02i_C2 --config C2
  echo
} > gpurun_out/r02_call55.log 2>&1
cat gpurun_out/r02_call55.log | cut -c1-300
