#!/bin/bash
# P2 to pairs of tiles (runs of 128 bytes) + the pair tile kernel: A/B against single tiles, parity subset.
mkdir -p gpurun_out
{
  echo "== parity subset"
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL\|rccl" | grep -E "passed|failed|error" | tail -3
  for v in 1; do
    echo "== bench C2 JFGPU_TILE_PAIR=$v"
    JFGPU_TILE_PAIR=$v timeout 900 python bench.py --config C2 --no-cpu-baseline --no-extras --repeats 3 2> gpurun_out/r02_c13_$v.err | grep '^{' > gpurun_out/r02_bench_C2_pair$v.json; tail -1 gpurun_out/r02_c13_$v.err
    python - <<PY
import json
d = json.load(open("gpurun_out/r02_bench_C2_pair$v.json"))
print("value", d["value"], {k: (x["ms"], x["launches"]) for k, x in d["kernels"].items()}, d["repeats"]["kmers_per_s"], d["content_digest"])
PY
  done
} > gpurun_out/r02_call13.log 2>&1
cat gpurun_out/r02_call13.log | cut -c1-1200
