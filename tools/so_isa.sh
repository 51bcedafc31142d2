#!/bin/bash
# Disassembly + register / scratch / LDS metadata of the gfx950 code object inside a built engine library.
#   tools/so_isa.sh <lib.so> <outdir>      ->  <outdir>/dev.s (llvm-objdump -d) and <outdir>/meta.txt (one line per kernel)
set -e
lib=$1; out=$2; mkdir -p $out
LLVM=/opt/rocm/lib/llvm/bin
objcopy -O binary --only-section=.hip_fatbin $lib $out/fatbin.bin
$LLVM/clang-offload-bundler --unbundle --type=o --input=$out/fatbin.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$out/dev.co
$LLVM/llvm-objdump -d --no-show-raw-insn $out/dev.co > $out/dev.s
$LLVM/llvm-readelf --notes $out/dev.co | python3 -c '
import sys, re
cur = {}
for line in sys.stdin:
    m = re.match(r"\s+(-\s+)?\.(\w+):\s+(.*)", line)
    if not m: continue
    if m.group(2) in ("name", "vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size") :
        cur[m.group(2)] = m.group(3).strip()
    if m.group(2) == "vgpr_spill_count":
        cur["spill"] = m.group(3).strip()
    if m.group(2) == "symbol": cur["symbol"] = m.group(3).strip()
    if m.group(2) == "wavefront_size":          # (the keys of a kernel come in alphabetical order: this is the last one)
        print("%s vgpr %s agpr %s sgpr %s scratch %s lds %s spill %s" % (cur.get("symbol"), cur.get("vgpr_count"), cur.get("agpr_count"), cur.get("sgpr_count"), cur.get("private_segment_fixed_size"), cur.get("group_segment_fixed_size"), cur.get("spill")))
        cur = {}
' > $out/meta.txt
wc -l $out/meta.txt
