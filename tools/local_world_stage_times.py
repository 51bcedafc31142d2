#!/usr/bin/env python3
"""Stage times of the N-rank code path with all ranks in ONE process on one device (the in-process transport: device copies
instead of RCCL, the ranks' kernels run one after the other, so the per-stage device times are not inflated by processes
time-slicing the GPU as in `bench.py --gpus N` on a one-GPU box).  What it shows: the cost of the receive side's W-way split
(exchange_receive_split) and of the routing P1 at world W, per rank, for a total of `gbp` Gbp into a global table of 2^34 slots.
usage: [JFGPU_COMM_GBITS=..] python tools/local_world_stage_times.py [world=4] [gbp=5] [log2 of the global table size = 34] [k=21]
(JFGPU_COMM_GBITS=8 with k = 20 at world 4: the 8-way receive split of an 8-GPU run's full-size shards on 2^32-slot shards.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jellyfish_amd import capi
world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
gbp = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
lsize_g = int(sys.argv[3]) if len(sys.argv) > 3 else 34
L, K, steps = 150, (int(sys.argv[4]) if len(sys.argv) > 4 else 21), 5
n_reads = int(gbp * 1e9 / L) // world          # per rank
sb = world.bit_length() - 1
shards = [capi.Table(K, 1 << lsize_g, canonical=True, shard_bits=sb, shard_id=r) for r in range(world)]
comm = capi.Comm(world, local=True)
bufs = []
for r, t in enumerate(shards):
    d = t.malloc(n_reads * (L + 1) + 16)
    t.gen_reads_dev(d, r * n_reads, n_reads, L, 42)
    t.reserve(n_reads * (L + 1) * 2)
    t.sync()
    bufs.append(d)
for rep in range(2):
    for t in shards:
        t.clear(); t.profile_enable(True); t.profile_reset()
    t0 = time.time()
    for i in range(steps):
        a, b = n_reads * i // steps, n_reads * (i + 1) // steps
        comm.local_step(shards, [bufs[r] + a * (L + 1) for r in range(world)], [(b - a) * (L + 1)] * world)
    sent, received = comm.finish()
    for t in shards:
        t.sync()
    dt = time.time() - t0
names = ("count_direct", "exchange_receive_split", "exchange_route_p1", "lookup", "p1_partition", "p2_partition", "tile_insert", "items_direct")
tot = sum(t.stats().total for t in shards)
print("world", world, "total k-mers", tot, "sent == received", sent == received, "%.1f ms for all ranks one after the other" % (dt * 1e3), "= %.1f G k-mers/s of one device's time" % (tot / dt / 1e9))
for r, t in enumerate(shards):
    print("  rank", r, {nm: round(t.profile_get(i)[0], 1) for i, nm in enumerate(names) if t.profile_get(i)[1]}, "direct inserts", t.counters().get("direct"))
comm.close()
for t in shards:
    t.close()
