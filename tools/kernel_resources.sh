#!/bin/bash
# Per-kernel register / LDS / scratch use of the engine as hipcc compiles it for gfx950 (-Rpass-analysis=kernel-resource-usage).
#   tools/kernel_resources.sh [pattern]
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -c -Rpass-analysis=kernel-resource-usage \
      -o /dev/null jellyfish_amd/csrc/jfgpu.hip 2>&1 |
python3 -c '
import re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    for key, lab in (("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("SGPRs", "sgpr"), ("ScratchSize \[bytes/lane\]", "scratch"), ("Occupancy \[waves/SIMD\]", "occ"), ("LDS Size \[bytes/block\]", "lds")):
        m = re.search(key + r": (\d+)", line)
        if m and cur is not None: cur[lab] = int(m.group(1))
import subprocess
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("jfgpu::", "").replace("void ", "")
    if pat and pat not in name: continue
    print("%-70s vgpr %3d sgpr %3d scratch %4d lds %6d occ %d" % (name[:70], r.get("vgpr", -1), r.get("sgpr", -1), r.get("scratch", -1), r.get("lds", -1), r.get("occ", -1)))
' "$@"
