# round 6, call 16: P1 with the round's codes / validity bits at compile-time offsets (-2 vector instructions per position)
O=gpurun_out
{
for rep in 1 2; do echo "--- main (xs)"; JFGPU_MATRIX=xs python tools/c2_stage_times.py 2>&1 | grep "^k 21"; done
echo "--- main (reference matrix)"; python tools/c2_stage_times.py 2>&1 | grep "^k 21"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_cli_gpu.py -q -x -k "half_gbp or golden or xorshift" 2>&1 | tail -3
} > $O/r06_call16.log 2>&1
cat $O/r06_call16.log
