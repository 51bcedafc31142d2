#!/bin/bash
# The reference's opt-in big test (tests/big.sh: 30 Gbp of seeded random sequence, k = 16 -C, -s 4G,
# --out-counter-len 2; golden md5 of `histo`) through jellyfish-amd.  Needs ~31 GB in /dev/shm and ~8 min,
# most of it the reference's single-threaded generator.  usage: tools/big_golden.sh [out.txt]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/big_golden.txt}
D=/dev/shm/jf_big; mkdir -p $D $(dirname $OUT); cd $D
GOLD=f52abd3e2a7cc5089cc8f32cb607c4c5
{
  echo "tests/big.sh golden through jellyfish-amd ($(date -u +%FT%TZ))"
  t0=$(date +%s.%N)
  [ -f seq30g.fa ] || $R/oracle/_ref/ref_generate_sequence -o seq30g -r 1000 -s 1602176487 30000000000
  t1=$(date +%s.%N)
  echo "generate_sequence: $(awk "BEGIN{print $t1 - $t0}") s, $(stat -c %s seq30g.fa) bytes"
  $R/bin/jellyfish-amd count -m 16 -s 4000000000 -o big_16.jf -c 4 -p 253 -C --out-counter-len 2 -t 16 --timing big.timing seq30g.fa
  rc=$?
  t2=$(date +%s.%N)
  echo "jellyfish-amd count rc=$rc wall $(awk "BEGIN{print $t2 - $t1}") s"; cat big.timing
  $R/bin/jellyfish-amd histo big_16.jf > big_16.histo
  t3=$(date +%s.%N)
  echo "histo wall $(awk "BEGIN{print $t3 - $t2}") s; output file $(stat -c %s big_16.jf) bytes"
  md5=$(md5sum big_16.histo | cut -d' ' -f1)
  echo "md5 $md5 expected $GOLD $( [ "$md5" = "$GOLD" ] && echo MATCH || echo MISMATCH )"
  head -5 big_16.histo
} > $OUT 2>&1
rm -rf $D
cat $OUT
