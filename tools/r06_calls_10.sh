# round 6, call 10: (a) the sharded growth under the xor-shift matrix, ipc transport, world 4, with the grow trace;
# (b) the at-scale recheck (C2 / C5 / C3 digests against the reference's, C2 and C5 also under --matrix xs)
O=gpurun_out
python - <<'PY'
import random
rng = random.Random(23 + 4)
with open("/tmp/reads.fa", "wb") as f:
    for r in range(3000):
        f.write((">r%d\n%s\n" % (r, "".join(rng.choice("ACGT") for _ in range(150)))).encode())
PY
{
for M in xs reference; do
echo "--- CLI --gpus 4 ipc, $M, grow trace"
JFGPU_MATRIX=$M JFGPU_COMM_TRANSPORT=ipc JFGPU_PARSE_CHUNK=100000 JFGPU_COMM_TRACE=1 timeout 300 bin/jellyfish-amd count -m 21 -C -s 1k -o /tmp/g4.jf --gpus 4 /tmp/reads.fa 2>&1 | grep "grow:\|does not own" | sort | tail -60
done
} > $O/r06_gpus4_grow.log 2>&1
tail -70 $O/r06_gpus4_grow.log
mkdir -p $O/r06_at_scale
AT_SCALE_PHASE=recheck bash tools/at_scale_parity.sh $PWD/$O/r06_at_scale > $O/r06_at_scale.log 2>&1; echo "recheck rc $?"
tail -40 $O/r06_at_scale/summary_recheck.txt
