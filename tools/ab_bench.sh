#!/bin/bash
# A/B lines of bench.py on the GPU box: one compact line per variant (value, repeats, per-stage device ms).
#   tools/ab_bench.sh "<name>:<ENV=.. ENV=..>:<bench args>" ...      (results also in gpurun_out/ab_<name>.json)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; envs=${rest%%:*}; args=${rest#*:}
  [ "$args" = "$rest" ] && args=""
  env $envs timeout 600 python bench.py --no-extras --no-cpu-baseline --repeats 3 $args 2> gpurun_out/ab_${name}.err | grep '^{' | tail -1 > gpurun_out/ab_${name}.json
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.load(open("gpurun_out/ab_%s.json" % name))
    k = {n: round(v["ms"], 2) for n, v in d["kernels"].items()}
    print("[ab] %-22s %.2f G k-mers/s  repeats %s  %s  tile %s" % (name, d["value"] / 1e9, [round(x / 1e9, 2) for x in d["repeats"]["kmers_per_s"]], k, d.get("tile_kernel", {}).get("direct_inserts")))
except Exception as e:
    print("[ab] %-22s FAILED %r" % (name, e)); print(open("gpurun_out/ab_%s.err" % name).read()[-1500:])
PY
done
