#!/bin/bash
# Single-pass P2 (reservations inside fixed regions per pair of tiles): parity, A/B against the exact P2.
mkdir -p gpurun_out
{
  echo "== parity"
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
  for v in 1 0; do
    echo "== bench C2 JFGPU_P2_SINGLE=$v"
    JFGPU_FLUSH_TRACE=1 JFGPU_P2_SINGLE=$v timeout 900 python bench.py --config C2 --no-cpu-baseline --no-extras --repeats 3 2> gpurun_out/r02_c16_$v.err | grep '^{' > gpurun_out/r02_bench_C2_p2s$v.json; grep flush gpurun_out/r02_c16_$v.err | tail -1
    python - <<PY
import json
d = json.load(open("gpurun_out/r02_bench_C2_p2s$v.json"))
print("value", d["value"], {k: (x["ms"], x["launches"]) for k, x in d["kernels"].items()}, d["repeats"]["kmers_per_s"], d["content_digest"])
PY
  done
} > gpurun_out/r02_call16.log 2>&1
cat gpurun_out/r02_call16.log | cut -c1-1200
