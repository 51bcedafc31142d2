"""One process per GPU, hash-prefix sharded counting.

The reference has no multi-process data path at all; this is the new exchange step.
The global table has 2^lsize_g positions; with G = 2^shard_bits ranks, rank r owns
the positions whose top shard_bits bits equal r, so equal k-mers always meet on
one GPU and the concatenation of the shards' sorted dumps (rank order) is the
globally sorted file body.  Per batch:

    encode+hash+bucket by owner (HIP)  ->  counts all-to-all  ->  keys all-to-all-v
    (RCCL over xGMI, 8 B per routed k-mer)  ->  insert what arrived (HIP)

torch.distributed is plumbing only (backend "nccl" == RCCL on ROCm; "gloo" on CPU for
the world_size-2 tests, where the device steps are played by the test's oracle-backed
backend).  A backend provides:
    partition(batch) -> (int64 tensor of keys grouped by destination rank, list of counts)
    insert(int64 tensor of keys, n)
"""
from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_bits_for(world_size: int) -> int:
    sb = world_size.bit_length() - 1
    if (1 << sb) != world_size:
        raise ValueError("world size must be a power of two (hash-prefix sharding)")
    return sb


MAX_KEYS_PER_MESSAGE = 1 << 27   # 1 GiB per peer per round


def exchange_keys(send: torch.Tensor, send_counts: List[int], group=None):
    """All-to-all-v of routed k-mers.  `send` holds the keys for rank 0, then rank 1, ...
    Returns a list of (received keys, count) pieces (order is irrelevant to the table).

    Messages are capped at 1 GiB per peer: a single 6.9 GB self-message (world size 1, one
    1 Gbp batch) was silently not delivered by RCCL 2.26, and byte counts above 2^31 are a
    classic overflow spot, so large exchanges run in rounds."""
    world = dist.get_world_size(group)
    assert len(send_counts) == world
    dev = send.device
    sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    rc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = [int(x) for x in rc.tolist()]
    gmax = torch.tensor([max(max(send_counts), max(recv_counts))], dtype=torch.int64, device=dev)
    dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
    rounds = max(1, -(-int(gmax.item()) // MAX_KEYS_PER_MESSAGE))
    send_off = [0]
    for c in send_counts:
        send_off.append(send_off[-1] + int(c))
    pieces = []
    for r in range(rounds):
        lo = r * MAX_KEYS_PER_MESSAGE
        s_part = [max(0, min(int(c) - lo, MAX_KEYS_PER_MESSAGE)) for c in send_counts]
        r_part = [max(0, min(int(c) - lo, MAX_KEYS_PER_MESSAGE)) for c in recv_counts]
        recv = torch.empty(int(sum(r_part)), dtype=torch.int64, device=dev)
        if rounds == 1:
            dist.all_to_all_single(recv, send[:send_off[-1]], output_split_sizes=r_part, input_split_sizes=s_part, group=group)
        else:   # this round's slice of every peer's region, packed (all_to_all_single wants one contiguous input)
            ins = torch.cat([send[send_off[p] + lo: send_off[p] + lo + s_part[p]] for p in range(world)])
            dist.all_to_all_single(recv, ins, output_split_sizes=r_part, input_split_sizes=s_part, group=group)
        pieces.append((recv, int(sum(r_part))))
    return pieces


class ShardedCounter:
    """Drives one rank's share of a sharded count."""

    def __init__(self, backend, group=None):
        self.backend = backend
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.sent = 0
        self.received = 0

    def step(self, batch):
        """Route one batch of this rank's input and insert what this rank owns."""
        send, counts = self.backend.partition(batch)
        got = 0
        for recv, n in exchange_keys(send, counts, self.group):
            self.backend.insert(recv, n)
            got += n
        self.sent += int(sum(counts))
        self.received += got
        return got


class GpuBackend:
    """HIP steps of ShardedCounter on one MI355X through the C ABI."""

    def __init__(self, table, capacity_keys: int, device: torch.device):
        self.t = table
        self.device = device
        self.cap = capacity_keys
        self.send = torch.empty(capacity_keys, dtype=torch.int64, device=device)
        self._keep = None

    def partition(self, batch):
        d_ptr, nbytes = batch
        counts = self.t.partition_ascii_dev(d_ptr, nbytes, self.send.data_ptr(), self.cap)   # synchronous
        return self.send, [int(c) for c in counts]

    def insert(self, recv: torch.Tensor, n: int):
        # device-wide: the collective runs on RCCL's own stream; waiting only for torch's current
        # stream let the insert kernel read the receive buffer before the data had landed
        torch.cuda.synchronize(self.device)
        self.t.wait()                                             # previous insert/P1 has consumed its buffer
        self._keep = recv                                         # keep alive while the kernel reads it (the previous one is free now)
        if n:
            self.t.add_keys_dev(recv.data_ptr(), n, 1)           # direct insert, or P1-partitioned and applied at sync
