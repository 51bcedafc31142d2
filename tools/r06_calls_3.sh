export JFGPU_MATRIX=xs
for v in "" st2 st3 prio; do
  echo "--- variant '$v'"
  if [ -z "$v" ]; then python tools/c2_stage_times.py 2>&1 | grep "^k 21"; else JFGPU_LIB=jellyfish_amd/lib/libjfgpu_$v.so python tools/c2_stage_times.py 2>&1 | grep "^k 21"; fi
done > gpurun_out/r06_p1stage2.log 2>&1
cat gpurun_out/r06_p1stage2.log
