# round 6, call 13: P1 with the rounds that skip the run logic (waves that see no chunk of eight equal bases) against the same
# kernel always taking the run logic (-DJFGPU_P1_ALWAYS_RUNS); then the GPU suite
O=gpurun_out
{
for rep in 1 2; do
echo "--- main (xs)"; JFGPU_MATRIX=xs python tools/c2_stage_times.py 2>&1 | grep "^k 21"
echo "--- runs (xs)"; JFGPU_MATRIX=xs JFGPU_LIB=jellyfish_amd/lib/libjfgpu_runs.so python tools/c2_stage_times.py 2>&1 | grep "^k 21"
done
echo "--- main (reference matrix)"; python tools/c2_stage_times.py 2>&1 | grep "^k 21"
bash tools/ab_bench.sh "C2G::--dist G --repeats 2" "C2Gruns:JFGPU_LIB=jellyfish_amd/lib/libjfgpu_runs.so:--dist G --repeats 2"
} > $O/r06_call13_p1.log 2>&1
cat $O/r06_call13_p1.log
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/r06_call13_suite.log
cat $O/r06_call13_suite.log
