import os, sys, random
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from jellyfish_amd import capi as gpu
rng = random.Random(27)
reads = [">r%d\n%s\n" % (r, "".join(rng.choice("ACGT") for _ in range(150))) for r in range(3000)]
seqs = ["".join(x.split("\n")[1] + "N" for x in reads[i::4]) for i in range(4)]
world, k, sb = 4, 21, 2
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shards = [gpu.Table(k, size, shard_bits=sb, shard_id=r) for r in range(world)]
print("lsize0", shards[0].info.lsize, "xs", shards[0].matrix_is_xorshift())
comm = gpu.Comm(world, local=True)
chunk = 100000
pos = 0
while pos < max(len(s) for s in seqs):
    ptrs, ns, keep = [], [], []
    for r in range(world):
        piece = seqs[r][pos:pos + chunk].encode()
        d = shards[r].malloc(len(piece) + 64)
        if piece: shards[r].h2d(d, np.frombuffer(piece, dtype=np.uint8))
        ptrs.append(d); ns.append(len(piece))
    comm.local_step(shards, ptrs, ns)
    pos += chunk
print(comm.finish())
for t in shards:
    try:
        t.sync(); print("lsize", t.info.lsize, t.stats().total, t.matrix_is_xorshift())
    except Exception as e:
        print("ERR", e)
