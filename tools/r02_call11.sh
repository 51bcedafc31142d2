#!/bin/bash
# P2 with the next chunk's loads in flight: phase clocks, parity subset, C2 and C5 lines.
mkdir -p gpurun_out
{
  echo "== phase clocks (C2 flow, tools/ablate.py)"
  JFGPU_LIB=$PWD/jellyfish_amd/lib/libjfgpu_phaseprof.so timeout 600 python tools/ablate.py 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== parity subset"
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py tests/test_gpu_bloom.py -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|RCCL\|rccl" | tail -4
  for c in C2 C5; do
    echo "== bench $c"
    timeout 900 python bench.py --config $c --no-cpu-baseline --no-extras --repeats 2 2> gpurun_out/r02_c11_$c.err | grep '^{' > gpurun_out/r02_bench_${c}_c11.json; tail -1 gpurun_out/r02_c11_$c.err
    python - <<PY
import json
d = json.load(open("gpurun_out/r02_bench_${c}_c11.json"))
print("value", d["value"], {k: (x["ms"], x["launches"]) for k, x in d["kernels"].items()}, d["repeats"]["kmers_per_s"])
PY
  done
} > gpurun_out/r02_call11.log 2>&1
cat gpurun_out/r02_call11.log | cut -c1-1200
