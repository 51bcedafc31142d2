#!/bin/bash
mkdir -p gpurun_out
{
  for v in 0 1; do
    echo "== bench C2 JFGPU_SLOT64=$v"
    JFGPU_SLOT64=$v timeout 600 python bench.py --no-cpu-baseline --no-extras --repeats 3 2> gpurun_out/r02_s$v.err | grep '^{' > gpurun_out/r02_bench_C2_slot64_$v.json; tail -1 gpurun_out/r02_s$v.err
    python - <<PY
import json
d = json.load(open("gpurun_out/r02_bench_C2_slot64_$v.json"))
print("value", d["value"], {k: (x["ms"], x["launches"]) for k, x in d["kernels"].items()}, d["config"]["workload"][-40:], d["repeats"]["kmers_per_s"])
PY
  done
} > gpurun_out/r02_call8.log 2>&1
cat gpurun_out/r02_call8.log | cut -c1-800
