#!/bin/bash
mkdir -p gpurun_out
{
  echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
  echo "== bench C2 forced through the sharded path (RCCL world 1)"
  JFGPU_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --repeats 2 > gpurun_out/r02_bench_C2_forced_dist.json 2> gpurun_out/r02_fd.err; echo "rc=$?"; tail -3 gpurun_out/r02_fd.err; cut -c1-1800 gpurun_out/r02_bench_C2_forced_dist.json
  echo "== bench C5"
  timeout 600 python bench.py --config C5 --no-cpu-baseline --repeats 2 > gpurun_out/r02_bench_C5b.json 2> gpurun_out/r02_c5b.err; echo "rc=$?"; tail -3 gpurun_out/r02_c5b.err; cut -c1-1500 gpurun_out/r02_bench_C5b.json
  echo "== bench C2 (end to end with the pipelined feed)"
  timeout 900 python bench.py --no-cpu-baseline --repeats 2 > gpurun_out/r02_bench_C2b.json 2> gpurun_out/r02_c2b.err; echo "rc=$?"; tail -3 gpurun_out/r02_c2b.err
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_bench_C2b.json"))
print("value", d["value"], "e2e", json.dumps(d.get("end_to_end")))
PY
} > gpurun_out/r02_call4.log 2>&1
tail -40 gpurun_out/r02_call4.log | cut -c1-2500
