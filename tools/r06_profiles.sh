#!/bin/bash
# Round 6's evidence on the GPU box: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the three single-GPU
# configurations (tools/profile_bench.sh), the traffic files bench.py quotes (tools/make_traffic_json.py), SQ counters of
# config 2, and the request counters of config 3's filtered pass (how large are the random reads of the Bloom counter?).
cd "$(dirname "$0")/.."
R=$(pwd)
for c in C2 C3 C5; do
  tools/profile_bench.sh r06_$c --config $c
done
python tools/make_traffic_json.py gpurun_out/r06_C2 C2 10 34 gpurun_out/r06_traffic_C2.json
python tools/make_traffic_json.py gpurun_out/r06_C3 C3 10 33 gpurun_out/r06_traffic_C3.json
python tools/make_traffic_json.py gpurun_out/r06_C5 C5 10 33 gpurun_out/r06_traffic_C5.json
tools/sq_counters.sh r06_C2_sqc --config C2 > gpurun_out/r06_C2_sq_counters.txt 2>&1
tools/sq_counters.sh r06_C3_sqc --config C3 > gpurun_out/r06_C3_sq_counters.txt 2>&1
# config 3: read requests of the L2 to memory by size (one pass; gfx950 counter names as rocprofv3 lists them)
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --warmup 0 --repeats 1 --config C3"
rm -rf $R/gpurun_out/r06_C3_rd
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_MISS_sum -d $R/gpurun_out/r06_C3_rd -o p -- $B > /dev/null 2> $R/gpurun_out/r06_C3_rdreq.err
cd $R && python tools/rocpd_pmc.py gpurun_out/r06_C3_rd > gpurun_out/r06_C3_rdreq.csv 2>> gpurun_out/r06_C3_rdreq.err; rm -rf gpurun_out/r06_C3_rd
ls -la gpurun_out/r06_*
