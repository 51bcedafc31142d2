import os, sys
sys.path.insert(0, ".")
import torch, torch.distributed as dist
from jellyfish_amd import capi, dist as jd
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
k, L, n_reads = 21, 150, 400_000
stride = L + 1
t = capi.Table(k, 1 << 28)
t.set_mode(1)
buf = torch.empty(n_reads * stride + 16, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
t.gen_reads_dev(buf.data_ptr(), 0, n_reads, L, 42); t.sync()
half = n_reads // 2
be = jd.GpuBackend(t, half * (L - k + 1) + 1024, dev)
sc = jd.ShardedCounter(be)
for i in range(2):
    p, n = buf.data_ptr() + i * half * stride, half * stride
    print("batch", i, "input checksum", int(buf[i * half * stride:(i + 1) * half * stride].to(torch.int64).sum()))
    send, counts = be.partition((p, n))
    print(" send checksum", int(send[:counts[0]].sum()), counts)
    for recv, nrecv in jd.exchange_keys(send, counts):
        torch.cuda.synchronize()
        print(" recv checksum", int(recv.sum()), nrecv, recv.data_ptr())
        be.insert(recv, nrecv)
t.sync()
s = t.stats(); print("distinct", s.distinct, "total", s.total)
dist.destroy_process_group()
