O=gpurun_out
python - <<'PY'
import random
rng = random.Random(23 + 4)
with open("/tmp/reads.fa", "wb") as f:
    for r in range(3000):
        f.write((">r%d\n%s\n" % (r, "".join(rng.choice("ACGT") for _ in range(150)))).encode())
PY
{
echo "--- capi, local transport, world 4, xs"; JFGPU_MATRIX=xs python tools/probes/r06_sharded_growth_repro.py 2>&1 | tail -6
echo "--- CLI --gpus 4 ipc, xs, trace"
JFGPU_MATRIX=xs JFGPU_COMM_TRANSPORT=ipc JFGPU_PARSE_CHUNK=100000 JFGPU_COMM_TRACE=1 timeout 300 bin/jellyfish-amd count -m 21 -C -s 1k -o /tmp/g4.jf --gpus 4 /tmp/reads.fa 2>&1 | grep -v "waiting\|pull\|barrier" | tail -60; echo "rc $?"
echo "--- CLI --gpus 4 ipc, xs, -s 4M (no growth)"
JFGPU_MATRIX=xs JFGPU_COMM_TRANSPORT=ipc JFGPU_PARSE_CHUNK=100000 timeout 300 bin/jellyfish-amd count -m 21 -C -s 4M -o /tmp/g4b.jf --gpus 4 /tmp/reads.fa 2>&1 | tail -5; echo "rc $?"
} > $O/r06_gpus4_xs2.log 2>&1
tail -90 $O/r06_gpus4_xs2.log
