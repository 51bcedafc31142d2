O=gpurun_out
python - <<'PY'
import random
rng = random.Random(23 + 4)
with open("/tmp/reads.fa", "wb") as f:
    for r in range(3000):
        f.write((">r%d\n%s\n" % (r, "".join(rng.choice("ACGT") for _ in range(150)))).encode())
PY
export JFGPU_MATRIX=xs JFGPU_COMM_TRANSPORT=ipc JFGPU_PARSE_CHUNK=100000
{ timeout 300 bin/jellyfish-amd count -m 21 -C -s 1k -o /tmp/g4.jf --gpus 4 /tmp/reads.fa; echo "rc $?"; } > $O/r06_gpus4_xs.log 2>&1
tail -30 $O/r06_gpus4_xs.log
