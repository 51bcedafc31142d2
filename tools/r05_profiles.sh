#!/bin/bash
# Round 5's evidence on the GPU box: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the three single-GPU
# configurations (tools/profile_bench.sh), the traffic files bench.py quotes (tools/make_traffic_json.py), SQ counters.
cd "$(dirname "$0")/.."
R=$(pwd)
for c in C2 C3 C5; do
  tools/profile_bench.sh r05_$c --config $c
done
python tools/make_traffic_json.py gpurun_out/r05_C2 C2 10 34 gpurun_out/r05_traffic_C2.json
python tools/make_traffic_json.py gpurun_out/r05_C3 C3 10 33 gpurun_out/r05_traffic_C3.json
python tools/make_traffic_json.py gpurun_out/r05_C5 C5 10 33 gpurun_out/r05_traffic_C5.json
for c in C2 C3 C5; do tools/sq_counters.sh r05_${c}_sqc --config $c > gpurun_out/r05_${c}_sq_counters.txt 2>&1; done
ls -la gpurun_out/r05_*
