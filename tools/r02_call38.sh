#!/bin/bash
# C5 after the shared-buffer flush: rocprofv3 passes + the bench line.
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
{
  echo "== rocprofv3 passes C5 ($(date +%T))"
  bash tools/profile_bench.sh r02g_C5 --config C5
  echo
} > gpurun_out/r02_call38.log 2>&1
cat gpurun_out/r02_call38.log | cut -c1-300
