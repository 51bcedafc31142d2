export JFGPU_MATRIX=xs
R=$(pwd); O=$R/gpurun_out
{
for v in st1prio st2prio st3prio prio; do
  echo "--- variant '$v'"
  JFGPU_LIB=jellyfish_amd/lib/libjfgpu_$v.so python tools/c2_stage_times.py 2>&1 | grep "^k 21"
done
} > $O/r06_p1stage4.log 2>&1
cat $O/r06_p1stage4.log
