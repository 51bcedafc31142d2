# round 6, call 12: ipc transport policies on the sharded-growth tests; device query; P1 ablations; the whole GPU suite
O=gpurun_out
{
for keep in 1 0; do for M in reference xs; do
echo "=== JFGPU_IPC_KEEP=$keep JFGPU_MATRIX=$M: too_small x 2"
for rep in 1 2; do JFGPU_IPC_KEEP=$keep JFGPU_MATRIX=$M timeout 600 python -m pytest tests/test_cli_gpu.py -q -k "too_small" 2>&1 | tail -4; done
done; done
echo "=== query + bloom"
timeout 900 python -m pytest tests/test_cli_gpu.py -x -q -k "query_sequence" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_bloom.py -x -q -k "partitioned_insert" 2>&1 | tail -5
} > $O/r06_call12_tests.log 2>&1
cat $O/r06_call12_tests.log | tail -60
{
echo "--- main (xs)"; JFGPU_MATRIX=xs python tools/c2_stage_times.py 2>&1 | grep "^k 21"
for v in noruns xs2; do echo "--- $v (xs)"; JFGPU_MATRIX=xs JFGPU_LIB=jellyfish_amd/lib/libjfgpu_$v.so python tools/c2_stage_times.py 2>&1 | grep "^k 21"; done
} > $O/r06_call12_p1.log 2>&1
cat $O/r06_call12_p1.log
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/r06_call12_suite.log
cat $O/r06_call12_suite.log
