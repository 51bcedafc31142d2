#!/bin/bash
# usage: tools/r03_variants.sh <variant lib suffix>...   -- C2 stage times + phase clocks of experimental engine builds
mkdir -p gpurun_out/r03
for v in "$@"; do
  echo "== $v"
  JFGPU_LIB=$PWD/jellyfish_amd/lib/libjfgpu_$v.so timeout 280 python bench.py --no-extras --no-cpu-baseline --repeats 1 > gpurun_out/r03/var_$v.json 2> gpurun_out/r03/var_$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r03/var_$v.json"))
print("value", d["value"], {k: x["ms"] for k, x in d["kernels"].items()}, d["content_digest"])
PY
  grep "phase prof. T" gpurun_out/r03/var_$v.err | tail -1
done
