#!/bin/bash
# first GPU call of round 2: box facts, sanity of the r01 suite, RMW-rate probe, "before" numbers for configs 3 and 5,
# cost of the reference generator and of the reference CPU count per Gbp (for sizing the at-scale parity run)
mkdir -p gpurun_out
{
  echo "== box"; nproc; free -g | head -2; df -h /dev/shm /tmp . | cat; lscpu | grep -E "Model name|Socket|Thread|Core" 
  echo "== pytest -m gpu"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
  echo "== atomic scope probe"; timeout 300 tools/probes/atomic_scope_probe
  echo "== before (C3, C5 with r01 kernels)"; timeout 600 python tools/r02_before.py 10
  echo "== reference generator + CPU count, 1 Gbp"
  cd /dev/shm
  /usr/bin/time -v $GRAFT_REPO_ROOT/oracle/_ref/ref_generate_sequence -s 42 -r 150 -o reads1g 1000000000 2>&1 | grep -E "Elapsed|Maximum resident"
  ls -la reads1g.fa
  for t in 64 128 256; do
    $GRAFT_REPO_ROOT/oracle/_ref/ref_jf count -m 21 -C -s 2G -t $t --no-write --timing tim$t reads1g.fa; echo "-t $t"; cat tim$t
  done
  rm -f reads1g.fa
} > gpurun_out/r02_call1.log 2>&1
tail -80 gpurun_out/r02_call1.log
