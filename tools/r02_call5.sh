#!/bin/bash
mkdir -p gpurun_out
{
  echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6
  echo "== bench C2 forced through the sharded path (RCCL world 1)"
  JFGPU_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --repeats 2 2> gpurun_out/r02_fd.err | grep '^{' > gpurun_out/r02_bench_C2_forced_dist.json; tail -2 gpurun_out/r02_fd.err
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_bench_C2_forced_dist.json"))
print("forced dist value", d["value"], {k: (v["ms"], v["launches"]) for k, v in d["kernels"].items()})
PY
  echo "== bench C2 (end to end)"
  JFGPU_FEED_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --repeats 2 2> gpurun_out/r02_c2b.err | grep '^{' > gpurun_out/r02_bench_C2b.json; grep -i "feed\|error" gpurun_out/r02_c2b.err | tail -5
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_bench_C2b.json"))
print("value", d["value"], "e2e", json.dumps(d.get("end_to_end")))
PY
} > gpurun_out/r02_call5.log 2>&1
tail -30 gpurun_out/r02_call5.log | cut -c1-1500
