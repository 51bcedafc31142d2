#!/bin/bash
# The round's measurement call: at-scale parity against the reference on its own generator's 10 Gbp file (references in
# the background on the host cores), rocprofv3 kernel-trace + PMC passes of the three bench configurations meanwhile,
# then the bench lines themselves on a quiet host.
set -u
R=$PWD; mkdir -p gpurun_out
{
  echo "== at-scale parity: start ($(date +%T))"
  AT_SCALE_PHASE=start bash tools/at_scale_parity.sh $R/gpurun_out/at_scale
  cat gpurun_out/at_scale/timing.txt
  for c in C2 C3 C5; do
    echo "== rocprofv3 passes $c ($(date +%T))"
    bash tools/profile_bench.sh r02_$c --config $c
  done
  echo "== at-scale parity: finish ($(date +%T))"
  AT_SCALE_PHASE=finish bash tools/at_scale_parity.sh $R/gpurun_out/at_scale
  for c in C2 C3 C5; do
    echo "== bench $c ($(date +%T))"
    timeout 900 python bench.py --config $c 2> gpurun_out/r02_final_$c.err | grep '^{' > gpurun_out/r02_final_$c.json; tail -2 gpurun_out/r02_final_$c.err | cut -c1-300
    cut -c1-400 gpurun_out/r02_final_$c.json
  done
  echo "== done ($(date +%T))"
} > gpurun_out/r02_call9.log 2>&1
tail -60 gpurun_out/r02_call9.log | cut -c1-600
