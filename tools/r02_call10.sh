#!/bin/bash
# Phase clocks of the two partition kernels (a -DJFGPU_PHASE_PROF build of the engine), and the C3 line after the
# filtered-pass fix.
mkdir -p gpurun_out
{
  echo "== phase clocks (C2 flow, tools/ablate.py)"
  JFGPU_LIB=$PWD/jellyfish_amd/lib/libjfgpu_phaseprof.so timeout 600 python tools/ablate.py 2>&1 | grep -v amdgpu.ids | tail -12
  echo "== bench C3"
  timeout 900 python bench.py --config C3 --no-cpu-baseline --no-extras --repeats 2 2> gpurun_out/r02_c3c.err | grep '^{' > gpurun_out/r02_bench_C3c.json; tail -2 gpurun_out/r02_c3c.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r02_bench_C3c.json"))
print("value", d["value"], {k: (x["ms"], x["launches"]) for k, x in d["kernels"].items()}, d.get("passes"), d["repeats"]["kmers_per_s"])
PY
} > gpurun_out/r02_call10.log 2>&1
cat gpurun_out/r02_call10.log | cut -c1-1200
