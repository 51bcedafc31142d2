#!/bin/bash
# third GPU call: suite, then the three bench configurations with the round-2 kernels (no profiling yet)
mkdir -p gpurun_out
{
  echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
  for c in C5 C3 C2; do
    echo "== bench $c"; timeout 900 python bench.py --config $c --steps 10 --warmup 1 > gpurun_out/r02_bench_$c.json 2> gpurun_out/r02_bench_$c.err; echo "rc=$?"; tail -3 gpurun_out/r02_bench_$c.err; cut -c1-2500 gpurun_out/r02_bench_$c.json
  done
} > gpurun_out/r02_call3.log 2>&1
tail -60 gpurun_out/r02_call3.log | cut -c1-3000
