"""world_size-2 test of the hash-prefix exchange on CPU (gloo).  The device steps are played by
an oracle-backed backend (TEST ONLY); what is under test is the routing/exchange logic of
jellyfish_amd/dist.py: ownership by top hash bits, all-to-all-v bookkeeping, and that the union
of the shards equals the single-table result with shards being contiguous pos ranges."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import oracle_lib as O
from jellyfish_amd.dist import ShardedCounter, shard_bits_for
import jellyfish_amd.dist as jd
jd.MAX_KEYS_PER_MESSAGE = int(os.environ.get("JF_TEST_MAXMSG", str(1 << 27)))   # small => multi-round exchange

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
k, lsize_g = 21, 16
sb = shard_bits_for(world)
rng = np.random.default_rng(1234)           # same matrix on every rank (same seed)
cols = rng.integers(0, 1 << lsize_g, size=2 * k, dtype=np.uint64)

class OracleBackend:                        # CPU stand-in for the HIP steps
    def __init__(self): self.table = {}
    def partition(self, seq):
        kmers = O.extract(seq, k, True)[:, 0]
        pos = O.matrix_times(cols, lsize_g, 2 * k, kmers)
        owner = (pos >> np.uint64(lsize_g - sb)).astype(np.int64)
        order = np.argsort(owner, kind="stable")
        counts = np.bincount(owner, minlength=world).tolist()
        return torch.from_numpy(kmers[order].astype(np.int64)), counts
    def insert(self, recv, n):
        keys = recv.numpy().astype(np.uint64)[:n]
        pos = O.matrix_times(cols, lsize_g, 2 * k, keys)
        assert ((pos >> np.uint64(lsize_g - sb)) == rank).all(), "misrouted k-mer"
        for x in keys.tolist(): self.table[x] = self.table.get(x, 0) + 1

rs = np.random.default_rng(100 + rank)
batches = [bytes(rs.choice(list(b"ACGTN"), size=3000, p=[.24, .24, .24, .24, .04]).astype(np.uint8)) for _ in range(3)]
batches.append(b"")                          # a rank with nothing to send in one step
be = OracleBackend()
sc = ShardedCounter(be)
for b in batches: sc.step(b)
sc.finish()
tot = torch.tensor([sc.sent, sc.received], dtype=torch.int64)
dist.all_reduce(tot)
assert tot[0] == tot[1], "k-mers lost or duplicated in the exchange"
json.dump({"rank": rank, "batches": [b.decode() for b in batches], "table": {str(a): c for a, c in be.table.items()},
           "cols": cols.tolist()}, open(os.path.join(sys.argv[2], "rank%d.json" % rank), "w"))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,maxmsg", [(2, 1 << 27), (4, 1 << 27), (2, 300)])
def test_sharded_exchange_equals_single_table(tmp_path, world, maxmsg):
    import json
    import numpy as np
    import oracle_lib as O
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    port = 29600 + world + (maxmsg % 7)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", JF_TEST_MAXMSG=str(maxmsg))
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                    "--master-addr", "127.0.0.1", "--master-port", str(port), str(w), ROOT, str(tmp_path)],
                   check=True, env=env, timeout=300, capture_output=True)
    res = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(world)]
    whole = {}
    for r in res:
        for b in r["batches"]:
            keys, cnt = O.count(b.encode(), 21, True)
            for a, c in zip(keys[:, 0].tolist(), cnt.tolist()):
                whole[a] = whole.get(a, 0) + c
    merged = {}
    for r in res:
        for a, c in r["table"].items():
            assert int(a) not in merged, "a k-mer ended up on two shards"
            merged[int(a)] = c
    assert merged == whole
    # shards are contiguous, ordered pos ranges: concatenating per-shard (pos,key)-sorted lists is globally sorted
    cols = np.array(res[0]["cols"], dtype=np.uint64)
    glob = []
    for r in res:
        keys = np.array([int(a) for a in r["table"]], dtype=np.uint64)
        pos = O.matrix_times(cols, 16, 42, keys) if len(keys) else np.zeros(0, dtype=np.uint64)
        glob += sorted(zip(pos.tolist(), keys.tolist()))
    assert glob == sorted(glob)
