"""world_size 2 and 4 on CPU: real rank processes under torch.distributed (gloo), each holding one hash-prefix SHARD of a
table in the engine as built from the kernels' sources for the host (tests/host/hip_emu -- test infrastructure, see
tests/test_emu_kernels.py).  What is under test is the product's sharding across processes through the C ABI: the routing
kernels (jfgpu_partition_ascii_dev: owner = top bits of the global position), inserts into a shard that refuses k-mers
it does not own, and the shard dumps whose concatenation in rank order must be the single table's (pos, key)-sorted body.
The exchange itself is played by gloo's all_to_all here (the product's transports -- RCCL, and hipIpc* copies between rank
processes -- need a GPU: tests/test_cli_gpu.py::test_count_gpus_n_as_rank_processes_on_one_device)."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "host", "_build")

WORKER = r'''
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from jellyfish_amd import capi

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
sb = world.bit_length() - 1
k, size = 21, 1 << 20
rs = np.random.default_rng(100 + rank)
batches = [bytes(rs.choice(list(b"ACGTN"), size=30000, p=[.24, .24, .24, .24, .04]).astype(np.uint8)) for _ in range(3)]
batches.append(b"")                                         # a step in which this rank has nothing to send
sent = received = 0
with capi.Table(k, size, canonical=True, shard_bits=sb, shard_id=rank) as t:
    cap = 40000
    d_in, d_out = t.malloc(40000 + 64), t.malloc(cap * 8)
    for b in batches:
        counts = np.zeros(world, dtype=np.int64)
        keys = np.zeros(0, dtype=np.uint64)
        if b:
            t.h2d(d_in, np.frombuffer(b, dtype=np.uint8))
            counts = t.partition_ascii_dev(d_in, len(b), d_out, cap).astype(np.int64)     # the product's routing kernels
            keys = t.d2h(d_out, int(counts.sum()) * 8).view(np.uint64)
        sc = torch.from_numpy(counts.copy()); rc = torch.empty(world, dtype=torch.int64)
        dist.all_to_all_single(rc, sc)
        recv = torch.empty(int(rc.sum()), dtype=torch.int64)
        dist.all_to_all_single(recv, torch.from_numpy(keys.view(np.int64).copy()), output_split_sizes=rc.tolist(), input_split_sizes=counts.tolist())
        sent += int(counts.sum()); received += int(rc.sum())
        if len(recv):
            d_k = t.malloc(len(recv) * 8)
            t.h2d(d_k, recv.numpy().view(np.uint8))
            t.add_keys_dev(d_k, len(recv), 1)               # a misrouted key would be counted (CTR_MISROUTED -> error at sync)
            t.sync()
            t.free(d_k)
    t.sync()
    tot = torch.tensor([sent, received], dtype=torch.int64)
    dist.all_reduce(tot)
    assert tot[0] == tot[1], "k-mers lost or duplicated in the exchange"
    recs = t.dump_records()
    keys, cnts = capi.decode_records(recs, k, t.info.out_counter_len)
    st = t.stats()
    json.dump({"rank": rank, "batches": [b.decode() for b in batches], "keys": keys.tolist(), "counts": cnts.tolist(),
               "matrix": t.matrix().tolist(), "lsize": int(t.info.lsize), "distinct": int(st.distinct)},
              open(os.path.join(sys.argv[2], "rank%d.json" % rank), "w"))
dist.destroy_process_group()
'''


@pytest.fixture(scope="module")
def emu_lib():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    subprocess.check_call([os.path.join(ROOT, "tests", "host", "build_emu.sh")])
    return os.path.join(BUILD, "libjfgpu_emu.so")


@pytest.mark.parametrize("world", [2, 4])
def test_shards_in_rank_processes_equal_the_single_table(tmp_path, emu_lib, world):
    import oracle_lib as O
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", JFGPU_LIB=emu_lib, JFGPU_EMU_THREADS="2")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                    "--master-addr", "127.0.0.1", "--master-port", str(29700 + world), str(w), ROOT, str(tmp_path)],
                   check=True, env=env, timeout=600, capture_output=True)
    res = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(world)]
    whole = {}
    for r in res:
        for b in r["batches"]:
            keys, cnt = O.count(b.encode(), 21, True)
            for a, c in zip(keys[:, 0].tolist(), cnt.tolist()):
                whole[a] = whole.get(a, 0) + c
    merged = {}
    for r in res:
        assert len(r["keys"]) == r["distinct"]
        for a, c in zip(r["keys"], r["counts"]):
            assert a not in merged, "a k-mer ended up on two shards"
            merged[a] = c
    assert merged == whole and len(whole) > 50000
    # same matrix everywhere; shards are contiguous, ordered position ranges: their dumps concatenated in rank order are the
    # globally (pos, key)-sorted body, and every record sits on the rank its top position bits name
    cols = np.array(res[0]["matrix"], dtype=np.uint64)
    lsize, sb = res[0]["lsize"], world.bit_length() - 1
    glob = []
    for r in res:
        assert r["matrix"] == res[0]["matrix"]
        keys = np.array(r["keys"], dtype=np.uint64)
        pos = O.matrix_times(cols, lsize, 42, keys)
        assert ((pos >> np.uint64(lsize - sb)) == r["rank"]).all()
        glob += list(zip(pos.tolist(), keys.tolist()))
    assert glob == sorted(glob)
