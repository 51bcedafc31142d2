export JFGPU_MATRIX=xs
R=$(pwd); O=$R/gpurun_out
{
for v in bar2 bar2prio prio; do
  echo "--- variant '$v'"
  JFGPU_LIB=jellyfish_amd/lib/libjfgpu_$v.so python tools/c2_stage_times.py 2>&1 | grep "^k 21"
done
cd /tmp; export TMPDIR=/tmp
for v in "" st2 prio; do
  echo "--- kernel trace of variant '$v'"
  rm -rf $O/kt_$v
  if [ -z "$v" ]; then L=$R/jellyfish_amd/lib/libjfgpu.so; else L=$R/jellyfish_amd/lib/libjfgpu_$v.so; fi
  JFGPU_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$v -o p -- python $R/tools/c2_stage_times.py > /dev/null 2>&1
  python $R/tools/rocpd_stats.py $O/kt_$v | head -8 | cut -c1-200
  rm -rf $O/kt_$v
done
} > $O/r06_p1stage3.log 2>&1
cat $O/r06_p1stage3.log
