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


def exchange_keys(send: torch.Tensor, send_counts: List[int], group=None) -> Tuple[torch.Tensor, List[int]]:
    """All-to-all-v of routed k-mers.  `send` holds the keys for rank 0, then rank 1, ...
    Returns (received keys, per-source counts)."""
    world = dist.get_world_size(group)
    assert len(send_counts) == world
    dev = send.device
    sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    rc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = [int(x) for x in rc.tolist()]
    total_send = int(sum(send_counts))
    recv = torch.empty(int(sum(recv_counts)), dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv, send[:total_send], output_split_sizes=recv_counts, input_split_sizes=list(send_counts),
                           group=group)
    return recv, recv_counts


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
        recv, rcounts = exchange_keys(send, counts, self.group)
        n = int(sum(rcounts))
        self.backend.insert(recv, n)
        self.sent += int(sum(counts))
        self.received += n
        return n


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
        torch.cuda.current_stream(self.device).synchronize()     # the all-to-all has landed
        self.t.sync()                                             # previous insert retired -> its buffer may go
        self._keep = recv                                         # keep alive while the kernel reads it
        if n:
            self.t.add_keys_dev(recv.data_ptr(), n, 1)
