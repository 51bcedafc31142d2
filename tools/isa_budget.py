#!/usr/bin/env python3
"""Instruction budget of one kernel from a disassembly (llvm-objdump -d --no-show-raw-insn of the gfx950 code object).

  tools/isa_budget.py dev.s <mangled-name-substring> [--blocks]

Cuts the kernel at branch targets and branch instructions and prints, per basic block, how many vector-ALU, LDS, scalar,
vector-memory and wait instructions it holds; --top N prints the N largest blocks' opcode histograms (the unrolled round
of a partition kernel is the largest block by far)."""
import re, sys, collections

def load(path, pat):
    lines = open(path).read().split("\n")
    out, on = [], False
    for ln in lines:
        m = re.match(r"^([0-9a-f]+) <(.*)>:$", ln)
        if m:
            if on: break
            on = pat in m.group(2)
            if on: base = int(m.group(1), 16)
            continue
        if on and ln.strip(): out.append(ln)
    return base, out

def klass(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"

def main():
    path, pat = sys.argv[1], sys.argv[2]
    top = 3
    if "--top" in sys.argv: top = int(sys.argv[sys.argv.index("--top") + 1])
    base, body = load(path, pat)
    ins = []
    for ln in body:
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if not m: continue
        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    addr_idx = {a: i for i, (a, _, _) in enumerate(ins)}
    cuts = {0}
    for i, (a, op, args) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            cuts.add(i + 1)
            m = re.search(r"\+0x([0-9a-f]+)>", args)
            if m:
                tgt = base + int(m.group(1), 16)
                if tgt in addr_idx: cuts.add(addr_idx[tgt])
        if op.startswith("s_endpgm"): cuts.add(i + 1)
    cuts = sorted(c for c in cuts if c < len(ins)) + [len(ins)]
    blocks = []
    for b in range(len(cuts) - 1):
        seg = ins[cuts[b]:cuts[b + 1]]
        cnt = collections.Counter(klass(op) for _, op, _ in seg)
        blocks.append((cuts[b], len(seg), cnt, seg))
    tot = collections.Counter()
    for _, _, cnt, _ in blocks: tot.update(cnt)
    print("kernel: %d instructions in %d blocks; totals %s" % (len(ins), len(blocks), dict(tot)))
    for start, n, cnt, seg in sorted(blocks, key=lambda b: -b[1])[:top]:
        print("\nblock @%d (+0x%x): %d instructions %s" % (start, seg[0][0] - base, n, dict(cnt)))
        h = collections.Counter(op for _, op, _ in seg)
        for op, c in h.most_common(): print("   %4d  %s" % (c, op))

if __name__ == "__main__": main()
