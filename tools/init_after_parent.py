import os, sys, time, subprocess, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np
from jellyfish_amd import capi
pause = float(sys.argv[1]); reset = sys.argv[2] == "1"
L = 150; n = int(sys.argv[3]) if len(sys.argv) > 3 else 6_000_000
t = capi.Table(21, 1 << 34, canonical=True)
buf = t.malloc(66_700_000 * 151 + 16)      # the bench's 10 Gbp of reads
t.reserve(66_700_000 * 151)
fa = "/dev/shm/jf_init_probe.fa"
step = 4_000_000
d = t.malloc(step * (L + 1) + 16)
with open(fa, "wb") as f:
    for r0 in range(0, n, step):
        m = min(step, n - r0)
        t.gen_reads_dev(d, r0, m, L, 42); t.wait()
        body = t.d2h(d, m * (L + 1)).reshape(m, L + 1); body[:, L] = ord("\n")
        np.concatenate([np.tile(np.frombuffer(b">r\n", dtype=np.uint8), (m, 1)), body], axis=1).tofile(f)
t.close()
if reset:
    hip = ctypes.CDLL("libamdhip64.so"); hip.hipDeviceSynchronize(); hip.hipDeviceReset()
time.sleep(pause)
subprocess.check_call(["bin/jellyfish-amd", "count", "-m", "21", "-C", "-s", str(1 << 34), "--no-write", "--timing", "/dev/shm/jf_init_probe.t", fa], env=dict(os.environ, JFGPU_QUIET="1", JFGPU_INIT_TRACE="1"))
print("pause", pause, "reset", reset, open("/dev/shm/jf_init_probe.t").read().split())
os.unlink(fa)
