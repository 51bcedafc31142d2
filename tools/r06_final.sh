#!/bin/bash
# Round 6's closing measurements on the GPU box: the lines profiles/r06_final_bench_*.json hold, then tools/at_scale_parity.sh.
cd "$(dirname "$0")/.."
O=gpurun_out
last_json() { grep '^{' | tail -1; }
python bench.py 2> $O/r06_final_default.err | last_json > $O/r06_final_bench_default.json
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras 2> /dev/null | last_json > $O/r06_final_bench_steps20_warmup5.json
JFGPU_BENCH_FORCE_DIST=1 python bench.py --no-extras --no-cpu-baseline 2> /dev/null | last_json > $O/r06_final_bench_C2_sharded_path_one_gpu.json
timeout 600 python bench.py --gpus 2 --no-extras --no-cpu-baseline 2> $O/r06_final_gpus2.err | last_json > $O/r06_final_bench_gpus2_on_one_device.json
timeout 600 python bench.py --gpus 4 --no-extras --no-cpu-baseline 2> $O/r06_final_gpus4.err | last_json > $O/r06_final_bench_gpus4_on_one_device.json
for f in default steps20_warmup5 C2_sharded_path_one_gpu gpus2_on_one_device gpus4_on_one_device; do
  python - $O/r06_final_bench_$f.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1], "%.2f" % (d["value"] / 1e9), {k: round(v["ms"], 2) for k, v in d.get("kernels", {}).items()}, {k: round(v / 1e9, 2) if isinstance(v, (int, float)) else v for k, v in d.get("roofline", {}).get("secondary_values", {}).items()})
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
done
true
