#!/bin/bash
# At-scale parity: the reference generator's own 10 Gbp file
#   generate_sequence -s 42 -r 150 -o reads10g 10000000000
# counted by jellyfish-amd on the GPU and by the reference's classes (oracle/_ref/ref_jf) on the host cores, for the
# three single-GPU configurations of BASELINE.json, compared through the content digest of the whole table
# (jfgpu_digest / ref_jf --digest: records, sum of counts, sum and xor of a per-record hash) and, for config 3, through
# the Bloom counter files byte by byte.  Nothing of /root/reference is needed at run time: the binaries travel prebuilt.
#   usage: tools/at_scale_parity.sh <out dir> [bases=10000000000] [ref threads=64]
# The reference runs are started in the background (one per configuration) so that the caller can use the GPU meanwhile:
#   source this with AT_SCALE_PHASE=start, do other work, then AT_SCALE_PHASE=finish -- or run it plainly for both.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/at_scale}; BASES=${2:-10000000000}; RT=${3:-64}
OUT=$(realpath -m "$OUT")                                   # (the script works in /dev/shm: a relative <out dir> would be lost)
PHASE=${AT_SCALE_PHASE:-both}
W=/dev/shm/jf_at_scale
REF=$R/oracle/_ref/ref_jf; GEN=$R/oracle/_ref/ref_generate_sequence; CLI=$R/bin/jellyfish-amd
mkdir -p $OUT $W; cd $W
ms() { echo $(( ($(date +%s%N) - $1) / 1000000 )); }
SIZE21=$(( BASES * 16 / 10 )); SIZE63=$(( BASES * 8 / 10 ))       # -s 16G / 8G at 10 Gbp: 2^34 and 2^33 slots

if [ $PHASE = start ] || [ $PHASE = both ]; then
  t0=$(date +%s%N)
  [ -f reads.fa ] || $GEN -s 42 -r 150 -o reads $BASES
  echo "generator_ms $(ms $t0)" > $OUT/timing.txt
  ls -l reads.fa | awk '{print "fasta_bytes", $5}' >> $OUT/timing.txt
  # reference, in the background: C2 | C5 | C3 (bc, then count --bc)
  ( t=$(date +%s%N); $REF count -m 21 -C -s $SIZE21 -t $RT --no-write --digest $OUT/ref_c2.digest --timing $OUT/ref_c2.timing reads.fa; echo "ref_c2_wall_ms $(ms $t)" >> $OUT/timing.txt ) &
  ( t=$(date +%s%N); $REF count -m 63 -C -s $SIZE63 -t $RT --no-write --digest $OUT/ref_c5.digest --timing $OUT/ref_c5.timing reads.fa; echo "ref_c5_wall_ms $(ms $t)" >> $OUT/timing.txt ) &
  ( t=$(date +%s%N); $REF bc -m 31 -C -s $BASES -t $RT -o ref.bc reads.fa; echo "ref_c3_bc_wall_ms $(ms $t)" >> $OUT/timing.txt
    t=$(date +%s%N); $REF count -m 31 -C -s $SIZE63 -t $RT --bc ref.bc --no-write --digest $OUT/ref_c3.digest --timing $OUT/ref_c3.timing reads.fa; echo "ref_c3_count_wall_ms $(ms $t)" >> $OUT/timing.txt ) &
  # the engine, now
  export JFGPU_QUIET=1
  t=$(date +%s%N); $CLI count -m 21 -C -s $SIZE21 --no-write --digest $OUT/gpu_c2.digest --timing $OUT/gpu_c2.timing reads.fa; echo "gpu_c2_wall_ms $(ms $t)" >> $OUT/timing.txt
  t=$(date +%s%N); $CLI count -m 63 -C -s $SIZE63 --no-write --digest $OUT/gpu_c5.digest --timing $OUT/gpu_c5.timing reads.fa; echo "gpu_c5_wall_ms $(ms $t)" >> $OUT/timing.txt
  t=$(date +%s%N); $CLI bc -m 31 -C -s $BASES -o gpu.bc --timing $OUT/gpu_c3_bc.timing reads.fa; echo "gpu_c3_bc_wall_ms $(ms $t)" >> $OUT/timing.txt
  t=$(date +%s%N); $CLI count -m 31 -C -s $SIZE63 --bc gpu.bc --no-write --digest $OUT/gpu_c3.digest --timing $OUT/gpu_c3.timing reads.fa; echo "gpu_c3_count_wall_ms $(ms $t)" >> $OUT/timing.txt
fi

# recheck: the engine alone against the reference digests of an earlier full run (tests/golden/at_scale/: the generator's
# file is a function of its arguments, so are the reference's digests) -- C2, C5, C3, and C2 again through `count --gpus 1`
# (rank process, RCCL communicator, item exchange, sharded writer's digest)
if [ $PHASE = recheck ]; then
  t0=$(date +%s%N)
  [ -f reads.fa ] || $GEN -s 42 -r 150 -o reads $BASES
  echo "generator_ms $(ms $t0)" > $OUT/timing.txt
  export JFGPU_QUIET=1
  G=$R/tests/golden/at_scale
  t=$(date +%s%N); $CLI count -m 21 -C -s $SIZE21 --no-write --digest $OUT/gpu_c2.digest --timing $OUT/gpu_c2.timing reads.fa; echo "gpu_c2_wall_ms $(ms $t)" >> $OUT/timing.txt
  t=$(date +%s%N); $CLI count -m 21 -C -s $SIZE21 --matrix xs --no-write --digest $OUT/gpu_c2_xs.digest --timing $OUT/gpu_c2_xs.timing reads.fa; echo "gpu_c2_xs_wall_ms $(ms $t)" >> $OUT/timing.txt
  t=$(date +%s%N); $CLI count -m 63 -C -s $SIZE63 --matrix xs --no-write --digest $OUT/gpu_c5_xs.digest --timing $OUT/gpu_c5_xs.timing reads.fa; echo "gpu_c5_xs_wall_ms $(ms $t)" >> $OUT/timing.txt
  t=$(date +%s%N); $CLI count -m 21 -C -s $SIZE21 --no-write --digest $OUT/gpu_c2_gpus1.digest --timing $OUT/gpu_c2_gpus1.timing --gpus 1 reads.fa 2> $OUT/gpus1.err; echo "gpu_c2_gpus1_wall_ms $(ms $t)" >> $OUT/timing.txt
  t=$(date +%s%N); $CLI count -m 63 -C -s $SIZE63 --no-write --digest $OUT/gpu_c5.digest --timing $OUT/gpu_c5.timing reads.fa; echo "gpu_c5_wall_ms $(ms $t)" >> $OUT/timing.txt
  t=$(date +%s%N); $CLI bc -m 31 -C -s $BASES -o gpu.bc --timing $OUT/gpu_c3_bc.timing reads.fa; echo "gpu_c3_bc_wall_ms $(ms $t)" >> $OUT/timing.txt
  t=$(date +%s%N); $CLI count -m 31 -C -s $SIZE63 --bc gpu.bc --no-write --digest $OUT/gpu_c3.digest --timing $OUT/gpu_c3.timing reads.fa; echo "gpu_c3_count_wall_ms $(ms $t)" >> $OUT/timing.txt
  # round 5: the same Bloom pass through `bc --gpus 1` with the rank's own share sent through RCCL (every ncclSend / ncclRecv an
  # N-rank merge makes, 28 GB in rounds of 1 GiB): the file body must be the single-process one's
  t=$(date +%s%N); JFGPU_COMM_SELF_RCCL=1 $CLI bc -m 31 -C -s $BASES -o gpu1.bc --gpus 1 reads.fa 2> $OUT/bc_gpus1.err; echo "gpu_c3_bc_gpus1_wall_ms $(ms $t)" >> $OUT/timing.txt
  o0=$(( 9 + 10#$(head -c 9 gpu.bc) )); o1=$(( 9 + 10#$(head -c 9 gpu1.bc 2>/dev/null || echo 0) ))
  {
    if [ -s gpu1.bc ] && cmp -s -i $o0:$o1 gpu.bc gpu1.bc; then echo "bc --gpus 1 (merge through RCCL): body byte-identical to the single-process file ($(( $(stat -c %s gpu.bc) - o0 )) bytes)"; else echo "bc --gpus 1: body DIFFERENT or missing"; grep -v "RCCL\|rccl" $OUT/bc_gpus1.err | tail -5; fi
    for c in c2 c5 c3; do
      if cmp -s $G/ref_$c.digest $OUT/gpu_$c.digest; then echo "$c digest EQUAL to the reference's (tests/golden/at_scale): $(tr '\n' ' ' < $OUT/gpu_$c.digest)"; else echo "$c digest DIFFERENT"; echo " ref: $(tr '\n' ' ' < $G/ref_$c.digest)"; echo " gpu: $(tr '\n' ' ' < $OUT/gpu_$c.digest)"; fi
    done
    for c in c2 c5; do
      if cmp -s $G/ref_$c.digest $OUT/gpu_${c}_xs.digest; then echo "$c under the xor-shift matrix (--matrix xs): digest EQUAL to the reference's"; else echo "$c under --matrix xs: digest DIFFERENT: $(tr '\n' ' ' < $OUT/gpu_${c}_xs.digest)"; fi
    done
    if cmp -s $G/ref_c2.digest $OUT/gpu_c2_gpus1.digest; then echo "c2 through count --gpus 1: digest EQUAL"; else echo "c2 through count --gpus 1: digest DIFFERENT: $(tr '\n' ' ' < $OUT/gpu_c2_gpus1.digest)"; grep -v "RCCL\|rccl" $OUT/gpus1.err | tail -5; fi
    cat $OUT/timing.txt
    for f in gpu_c2 gpu_c2_gpus1 gpu_c5 gpu_c3_bc gpu_c3; do echo "-- $f.timing"; cat $OUT/$f.timing 2>/dev/null; done
  } | tee $OUT/summary_recheck.txt
  rm -rf $W
  # (round-5 review: an empty summary was committed as evidence) the summary must exist, name every configuration and hold no DIFFERENT
  if [ ! -s $OUT/summary_recheck.txt ] || [ $(grep -c "EQUAL" $OUT/summary_recheck.txt) -lt 6 ] || grep -q "DIFFERENT" $OUT/summary_recheck.txt; then
    echo "at_scale_parity: recheck summary empty, incomplete or with a difference" >&2; exit 1
  fi
fi

if [ $PHASE = finish ] || [ $PHASE = both ]; then
  wait
  # when sourced in two phases the background jobs belong to the first shell: wait for their outputs instead
  for f in ref_c2.digest ref_c5.digest ref_c3.digest; do
    n=0; while [ ! -s $OUT/$f ] && [ $n -lt 1500 ]; do sleep 2; n=$((n+1)); done
  done
  {
    for c in c2 c5 c3; do
      if cmp -s $OUT/ref_$c.digest $OUT/gpu_$c.digest; then echo "$c digest EQUAL: $(tr '\n' ' ' < $OUT/gpu_$c.digest)"; else echo "$c digest DIFFERENT"; echo " ref: $(tr '\n' ' ' < $OUT/ref_$c.digest)"; echo " gpu: $(tr '\n' ' ' < $OUT/gpu_$c.digest)"; fi
    done
    # Bloom counter files: same length, bodies byte-identical (headers differ in provenance fields only)
    h1=$(( 9 + 10#$(head -c 9 ref.bc) )); h2=$(( 9 + 10#$(head -c 9 gpu.bc) ))
    s1=$(stat -c %s ref.bc); s2=$(stat -c %s gpu.bc)
    if [ $((s1 - h1)) -eq $((s2 - h2)) ] && cmp -s -i $h1:$h2 ref.bc gpu.bc; then echo "c3 bloom counter bodies byte-identical: $((s1 - h1)) bytes"; else echo "c3 bloom counter bodies DIFFER (ref $((s1 - h1)) bytes, gpu $((s2 - h2)) bytes)"; fi
    cat $OUT/timing.txt
    for f in ref_c2 gpu_c2 ref_c5 gpu_c5 ref_c3 gpu_c3 gpu_c3_bc; do echo "-- $f.timing"; cat $OUT/$f.timing 2>/dev/null; done
  } | tee $OUT/summary.txt
  rm -rf $W
fi
