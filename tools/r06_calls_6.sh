R=$(pwd); O=$R/gpurun_out
{
for v in "" p2sp p2lp; do
  echo "--- variant '$v' (xs)"
  if [ -z "$v" ]; then JFGPU_MATRIX=xs python tools/c2_stage_times.py 2>&1 | grep "^k 21"; else JFGPU_MATRIX=xs JFGPU_LIB=jellyfish_amd/lib/libjfgpu_$v.so python tools/c2_stage_times.py 2>&1 | grep "^k 21"; fi
done
echo "--- main, reference matrix"; python tools/c2_stage_times.py 2>&1 | grep "^k 21"
JFGPU_MATRIX=xs timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_parity.py 2>&1 | tail -3
timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_parity.py -k 'ring_p2_kernels or slot32_equals or single_pass_p1 or ragged or unaligned or calls_do_not or comm_item or high_coverage' 2>&1 | tail -3
} > $O/r06_call6.log 2>&1
cat $O/r06_call6.log
