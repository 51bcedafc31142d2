#!/bin/bash
mkdir -p gpurun_out
{
  echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
  echo "== atomic scope probe"; timeout 300 tools/probes/atomic_scope_probe
  echo "== bloom at scale"; timeout 900 python tools/r02_bloom_scale.py 10 compare
  echo "== reference generator + CPU count, 1 Gbp"
  cd /dev/shm
  s=$(date +%s%N); $GRAFT_REPO_ROOT/oracle/_ref/ref_generate_sequence -s 42 -r 150 -o reads1g 1000000000; e=$(date +%s%N); echo "generator 1 Gbp: $(( (e - s) / 1000000 )) ms"
  ls -la reads1g*
  for t in 64 256; do
    $GRAFT_REPO_ROOT/oracle/_ref/ref_jf count -m 21 -C -s 2G -t $t --no-write --digest dg$t --timing tim$t reads1g.fa; echo "-t $t"; cat tim$t
  done
  cat dg64; cmp dg64 dg256 && echo "digests equal across thread counts"
  s=$(date +%s%N); $GRAFT_REPO_ROOT/bin/jellyfish-amd count -m 21 -C -s 2G --no-write --digest dgpu --timing timgpu reads1g.fa; e=$(date +%s%N); echo "jellyfish-amd count 1 Gbp wall: $(( (e - s) / 1000000 )) ms"; cat timgpu
  cmp dg64 dgpu && echo "GPU digest == reference digest (1 Gbp, reference generator file)"
  rm -f reads1g*
} > gpurun_out/r02_call2.log 2>&1
tail -100 gpurun_out/r02_call2.log
