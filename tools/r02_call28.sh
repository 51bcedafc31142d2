#!/bin/bash
# Round 2, final measurement of the committed engine: the whole GPU suite, at-scale re-check against the reference's
# digests, rocprofv3 passes (kernel trace + FETCH_SIZE + WRITE_SIZE) and the bench lines for C2 / C3 / C5, the sharded
# code path on one GPU.
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
{
  echo "== GPU suite ($(date +%T))"
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
  echo "== at-scale re-check ($(date +%T))"
  AT_SCALE_PHASE=recheck timeout 1500 bash tools/at_scale_parity.sh $GRAFT_REPO_ROOT/gpurun_out/at_scale > /dev/null 2>&1
  cat gpurun_out/at_scale/summary_recheck.txt
  for c in C2 C3 C5; do
    echo "== rocprofv3 passes $c ($(date +%T))"
    bash tools/profile_bench.sh r02f_$c --config $c
    echo
  done
  echo "== done ($(date +%T))"
} > gpurun_out/r02_call28.log 2>&1
tail -40 gpurun_out/r02_call28.log | cut -c1-300
