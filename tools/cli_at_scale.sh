#!/bin/bash
# `jellyfish-amd count` at the metric's scale from a fresh process, output enabled: Init / Counting / Writing of
# count_main.cc:375-382 with the records (86 GB at C2) written to /dev/shm.   usage: tools/cli_at_scale.sh [gbp] [k] [size]
set -u
GBP=${1:-10}; K=${2:-21}; SIZE=${3:-16G}
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$(mktemp -d /dev/shm/jf_scale_XXXX)
trap 'rm -rf $D' EXIT
python - "$D/reads.fa" "$GBP" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from jellyfish_amd import capi
path, gbp = sys.argv[1], float(sys.argv[2])
L = 150; n_reads = int(round(gbp * 1e9 / L)); step = 4_000_000
with capi.Table(21, 1 << 20) as t, open(path, "wb") as f:
    d = t.malloc(step * (L + 1) + 16)
    hdr = np.frombuffer(b">r\n", dtype=np.uint8)
    for r0 in range(0, n_reads, step):
        n = min(step, n_reads - r0)
        t.gen_reads_dev(d, r0, n, L, 42); t.wait()
        body = t.d2h(d, n * (L + 1)).reshape(n, L + 1); body[:, L] = ord("\n")
        np.concatenate([np.tile(hdr, (n, 1)), body], axis=1).tofile(f)
    t.free(d)
print("input", os.path.getsize(path) / 1e9, "GB")
PY
for rep in ${REPS:-1 2}; do
  T0=$(date +%s.%N)
  env JFGPU_QUIET=1 JFGPU_TIMING_DETAIL=1 $R/bin/jellyfish-amd count -m $K -C -s $SIZE -o $D/out.jf --timing $D/timing $D/reads.fa
  T1=$(date +%s.%N); echo "wall $(python3 -c "print(round($T1 - $T0, 3))") s"
  tr '\n' ' ' < $D/timing; echo; ls -l $D/out.jf | awk '{printf "output %.2f GB\n", $5/1e9}'
  rm -f $D/out.jf
done
