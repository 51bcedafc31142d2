#!/bin/bash
# Longer P2 runs for Bloom cell updates and two-word keys, early items-per-byte estimate for 16-byte items.
mkdir -p gpurun_out
{
  echo "== parity (wide, bloom)"
  timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_bloom.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
  for c in C3; do
    echo "== bench $c"
    JFGPU_FLUSH_TRACE=1 timeout 900 python bench.py --config $c --no-cpu-baseline --no-extras --repeats 2 2> gpurun_out/r02_c19_$c.err | grep '^{' > gpurun_out/r02_bench_${c}_c19.json; grep flush gpurun_out/r02_c19_$c.err | tail -4
    python - <<PY
import json
d = json.load(open("gpurun_out/r02_bench_${c}_c19.json"))
print("value", d["value"], {k: (x["ms"], x["launches"]) for k, x in d["kernels"].items()}, d.get("passes"), d["repeats"]["kmers_per_s"], d["content_digest"])
PY
  done
} > gpurun_out/r02_call19.log 2>&1
cat gpurun_out/r02_call19.log | cut -c1-1200
