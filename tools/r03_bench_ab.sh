#!/bin/bash
# Round-3 measurement call: C2 bench line on the product library, then once more on the -DJFGPU_PHASE_PROF build
# (phase clocks of P1 / P2 / T as wave 0 of every block sees them).  usage: tools/r03_bench_ab.sh <tag> [extra bench args]
tag=$1; shift
mkdir -p gpurun_out/r03
timeout 280 python bench.py --no-extras --no-cpu-baseline --repeats 2 "$@" > gpurun_out/r03/${tag}_bench.json 2> gpurun_out/r03/${tag}_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/r03/${tag}_bench.json"))
print("value", d["value"], {k: v["ms"] for k, v in d["kernels"].items()}, d["repeats"]["kmers_per_s"], d["content_digest"])
PY
if [ -f jellyfish_amd/lib/libjfgpu_phaseprof.so ]; then
  JFGPU_LIB=$PWD/jellyfish_amd/lib/libjfgpu_phaseprof.so timeout 280 python bench.py --no-extras --no-cpu-baseline --repeats 1 "$@" > gpurun_out/r03/${tag}_prof.json 2> gpurun_out/r03/${tag}_prof.err
  grep "phase prof" gpurun_out/r03/${tag}_prof.err | tail -3
fi
