#!/bin/bash
# Round 6: the Bloom pass at BASELINE scale against the reference itself -- `jellyfish-amd bc` (P1b through rings of 256 bytes)
# and `ref_jf bc` (the reference's classes on the host cores) on the reference generator's 10 Gbp file; the 28 GB bodies compared byte by byte.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r06_bc_at_scale; mkdir -p $OUT
W=/dev/shm/jf_bc_scale; mkdir -p $W; cd $W
ms() { echo $(( ($(date +%s%N) - $1) / 1000000 )); }
t=$(date +%s%N); $R/oracle/_ref/ref_generate_sequence -s 42 -r 150 -o reads 10000000000; echo "generator_ms $(ms $t)" > $OUT/timing.txt
( t=$(date +%s%N); $R/oracle/_ref/ref_jf bc -m 31 -C -s 10000000000 -t ${1:-64} -o ref.bc reads.fa; echo "ref_bc_wall_ms $(ms $t)" >> $OUT/timing.txt ) &
export JFGPU_QUIET=1
t=$(date +%s%N); JFGPU_FLUSH_TRACE=1 $R/bin/jellyfish-amd bc -m 31 -C -s 10000000000 -o gpu.bc --timing $OUT/gpu_bc.timing reads.fa 2> $OUT/gpu_bc.err; echo "gpu_bc_wall_ms $(ms $t)" >> $OUT/timing.txt
wait
h1=$(( 9 + 10#$(head -c 9 ref.bc) )); h2=$(( 9 + 10#$(head -c 9 gpu.bc) ))
s1=$(stat -c %s ref.bc); s2=$(stat -c %s gpu.bc)
{
if [ $((s1 - h1)) -eq $((s2 - h2)) ] && cmp -s -i $h1:$h2 ref.bc gpu.bc; then echo "bloom counter bodies byte-identical: $((s1 - h1)) bytes (reference: ref_jf bc -t ${1:-64}; engine: jellyfish-amd bc)"; else echo "bloom counter bodies DIFFER (ref $((s1 - h1)) bytes, gpu $((s2 - h2)) bytes)"; fi
cat $OUT/timing.txt; cat $OUT/gpu_bc.timing; grep -c "Bloom" $OUT/gpu_bc.err
} | tee $OUT/summary.txt
rm -rf $W
