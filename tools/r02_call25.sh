#!/bin/bash
# The exchange's item path: parity (local transport world 1/2/4, RCCL world 1), forced-dist bench on one GPU.
mkdir -p gpurun_out
{
  echo "== parity"
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
  for v in 1; do
    echo "== bench C2 through the sharded code path (JFGPU_BENCH_FORCE_DIST=1), JFGPU_COMM_ITEMS=$v"
    JFGPU_FLUSH_TRACE=1 JFGPU_COMM_ITEMS=$v JFGPU_BENCH_FORCE_DIST=1 timeout 900 python bench.py --config C2 --no-cpu-baseline --no-extras --repeats 2 2> gpurun_out/r02_c25_$v.err | grep '^{' > gpurun_out/r02_bench_C2_fd_items$v.json; grep "comm\]" gpurun_out/r02_c25_$v.err | tail -2; grep -i "error\|Traceback" -A5 gpurun_out/r02_c25_$v.err | head
    python - <<PY
import json
d = json.load(open("gpurun_out/r02_bench_C2_fd_items$v.json"))
print("value", d["value"], {k: (x["ms"], x["launches"]) for k, x in d["kernels"].items()}, d["repeats"]["kmers_per_s"], d["content_digest"])
PY
  done
} > gpurun_out/r02_call25.log 2>&1
cat gpurun_out/r02_call25.log | cut -c1-1200
