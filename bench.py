#!/usr/bin/env python3
"""bench.py -- k-mers/s of the `jellyfish count` hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): k=21 canonical, 10 Gbp of synthetic 150 bp reads per
GPU (uniform iid bases, the distribution of the reference's generate_sequence), table of
2^34 64-bit slots per GPU resident in HBM.  The reads are generated on the device before
the timed region, so `value` is throughput with the input already resident in HBM.

  step   = one batch (1/K of the 10 Gbp) through encode -> canonical -> GF(2) hash -> insert
  N = 1  : per step the batch is encoded, hashed and radix-partitioned on the device (P1); the
           last step's sync applies everything pending (P2 partition + LDS-resident tile insert),
           all inside the timed region.  JFGPU_MODE=direct selects the one-kernel atomic path.
  N > 1  : one process per GPU, table sharded by the top hash bits; per step
           partition (HIP) -> all-to-all-v of routed k-mers (RCCL over xGMI) -> insert (HIP).
           Weak scaling: every rank brings its own 10 Gbp.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (HBM; algorithmic bytes per k-mer
from SURVEY 8(d): 150/130 B of sequence + 16 B slot read-modify-write = 17.15 B) and
"cpu_baseline" (the reference's own CPU path, oracle/_ref, timed on this box's host cores on
a bounded sample of the same reads; falls back to the single-core C restatement when the
reference build is absent).  The oracle is only the baseline / checker here, never the
thing measured.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K = 21
READ_LEN = 150
B_ALG = READ_LEN / (READ_LEN - K + 1) + 16.0     # bytes per k-mer occurrence
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(sample_bytes, n_reads, k, tmpdir):
    """Reference CPU path on a bounded sample.  Returns (dict for the JSON line, stats text)."""
    import numpy as np
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_jf")
    kmers = n_reads * (READ_LEN - k + 1)
    arr = np.frombuffer(sample_bytes, dtype=np.uint8).reshape(n_reads, READ_LEN + 1).copy()
    if os.access(ref, os.X_OK):
        arr[:, READ_LEN] = ord("\n")
        hdr = np.tile(np.frombuffer(b">r\n", dtype=np.uint8), (n_reads, 1))
        fa = os.path.join(tmpdir, "sample.fa")
        np.concatenate([hdr, arr], axis=1).tofile(fa)
        cores = min(os.cpu_count() or 1, 64)
        out, timing = os.path.join(tmpdir, "ref.jf"), os.path.join(tmpdir, "timing")
        size = 1
        while size < 3 * kmers:
            size <<= 1
        subprocess.check_call([ref, "count", "-m", str(k), "-C", "-s", str(size), "-t", str(cores), "-o", out,
                               "--timing", timing, fa])
        t = dict(l.split() for l in open(timing).read().splitlines())
        assert int(t["Mers"]) == kmers
        stats = subprocess.check_output([ref, "stats", out]).decode()
        os.unlink(out); os.unlink(fa)
        return ({"value": kmers / float(t["Counting"]), "unit": "k-mers/s", "cores": cores, "kind": "reference",
                 "sample": "first %d reads (%.0f Mbp) of the same synthetic input, jellyfish 2.3.1 classes "
                           "(oracle/_ref), -t %d, table presized 2^%d, Counting phase only"
                           % (n_reads, n_reads * READ_LEN / 1e6, cores, size.bit_length() - 1)}, stats)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    n_small = min(n_reads, 100000)
    seq = arr[:n_small].tobytes()
    t0 = time.time()
    keys, cnt = O.count(seq, k, True)
    dt = time.time() - t0
    return ({"value": n_small * (READ_LEN - k + 1) / dt, "unit": "k-mers/s", "cores": 1, "kind": "port",
             "sample": "first %d reads through oracle/jf_oracle.c (sort+count restatement)" % n_small}, None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gbp", type=float, default=10.0, help="giga-bases of reads per GPU")
    ap.add_argument("--lsize", type=int, default=34, help="log2 slots per GPU")
    ap.add_argument("--cpu-sample-reads", type=int, default=666667)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist", choices=["U", "G"], default="U",
                    help="U: iid uniform reads (the metric's configuration); G: BASELINE.md's secondary distribution, reads sampled from a "
                         "100 Mbp random genome with 1 %% substitutions (about 100x coverage at 10 Gbp: most k-mers repeat)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from jellyfish_amd import capi
    from jellyfish_amd import dist as jd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU: there is no CPU path to measure"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    sb = jd.shard_bits_for(world)

    n_reads = int(round(args.gbp * 1e9 / READ_LEN))
    steps, warmup = args.steps, args.warmup
    stride = READ_LEN + 1
    kmers_per_read = READ_LEN - K + 1
    t = capi.Table(K, 1 << (args.lsize + sb), canonical=True, device=local_rank, shard_bits=sb, shard_id=rank)
    buf = torch.empty(n_reads * stride + 16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    if args.dist == "G":
        t.gen_genome_reads_dev(buf.data_ptr(), rank * n_reads, n_reads, READ_LEN, 100_000_000, 0.01, 42)
    else:
        t.gen_reads_dev(buf.data_ptr(), rank * n_reads, n_reads, READ_LEN, 42)
    t.sync()

    bounds = [n_reads * i // steps for i in range(steps + 1)]

    def batch(i):
        i %= steps
        return buf.data_ptr() + bounds[i] * stride, (bounds[i + 1] - bounds[i]) * stride

    # random-access roofline denominator on this very table (dirty afterwards -> cleared)
    gups = {}
    if rank == 0:
        for mode, name in ((0, "atomic_add"), (2, "atomic_cas")):
            gups[name] = t.gups(1 << 28, mode)
    t.clear()
    force_dist = os.environ.get("JFGPU_BENCH_FORCE_DIST") == "1"     # exercise the N>1 code path on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    # Init phase: workspace for one sync-to-sync span (like -s presizes the table)
    t.reserve(n_reads * stride if world == 1 and not force_dist else int(n_reads * kmers_per_read * 1.02) + (1 << 20))

    if world == 1 and not force_dist:
        def run_step(i):
            p, n = batch(i)
            t.count_ascii_dev(p, n)
    else:
        max_batch = max(bounds[i + 1] - bounds[i] for i in range(steps))
        be = jd.GpuBackend(t, max_batch * kmers_per_read, dev)
        sc = jd.ShardedCounter(be)

        def run_step(i):
            sc.step(batch(i))

    def fence():
        if world > 1 or force_dist:
            sc.finish()                  # last step's exchange + insert
        t.sync()
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()

    for i in range(warmup):
        run_step(i)
    fence()
    t.clear()
    t.profile_enable(True)
    t.profile_reset()
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        run_step(i)
    fence()
    elapsed = time.perf_counter() - t0
    t.profile_enable(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # every window of every read was counted exactly once (size-independent invariant)
    st = t.stats()
    tot = torch.tensor([st.total, st.distinct], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(tot)
    total_kmers = n_reads * kmers_per_read * world
    assert int(tot[0]) == total_kmers, "counted %d k-mers, expected %d" % (int(tot[0]), total_kmers)

    if rank == 0:
        slot_names = ["count_direct", "add_keys", "shard_partition", "lookup", "p1_partition", "p2_partition", "tile_insert", "items_direct"]
        kernels = {}
        for i, nm in enumerate(slot_names):
            kms, kl, ku = t.profile_get(i)
            if kl:
                kernels[nm] = {"ms": round(kms, 3), "launches": kl, "units": ku}
        which = max(range(len(slot_names)), key=lambda i: t.profile_get(i)[0])
        ms, launches, _ = t.profile_get(which)
        per_launch_kmers = int(st.total) / max(launches, 1)
        avg_ms = ms / max(launches, 1)
        achieved = per_launch_kmers * B_ALG / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        value = total_kmers / elapsed
        # HBM bytes per launch of the named kernel: PMC counters cannot be read from inside this process, so
        # the figure comes from the committed rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this same command
        # (profiles/r01_traffic.json); only quoted when the run matches that configuration.
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tj) and world == 1 and not force_dist and args.dist == "U" and abs(args.gbp - 10.0) < 1e-9 and args.lsize == 34 and steps == 10:
            traffic = json.load(open(tj)).get("per_timed_launch_bytes", {}).get(slot_names[which])
            traffic_src = None if traffic is None else "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, same command)"
        out = {
            "metric": "k-mers/sec at k=21 canonical, 150 bp synthetic reads, bit-exact counts",
            "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: k=21 -C, %.1f Gbp of 150 bp reads per GPU, 2^%d-slot "
                                    "64-bit table per GPU in HBM" % (args.gbp, args.lsize)) +
                                   ("" if args.dist == "U" else "; SECONDARY distribution G (reads from a 100 Mbp random genome, 1 % substitutions)"),
                       "k": K, "read_len": READ_LEN, "reads_per_gpu": n_reads, "table_slots_per_gpu": 1 << args.lsize,
                       "load_factor": float(tot[1]) / float(world << args.lsize),
                       "distinct": int(tot[1]), "total_kmers": total_kmers,
                       "parallelism": ("single GPU" if not force_dist else "single GPU through the sharded code path") if world == 1 else "hash-prefix shard x%d + all-to-all" % world},
            "kernels": kernels,
            "roofline": {"bound": "hbm", "kernel": slot_names[which],
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "bytes_per_kmer": B_ALG, "kmers_per_launch": per_launch_kmers,
                         "avg_launch_ms": avg_ms, "launches": launches,
                         "whole_path_achieved": value * B_ALG / 1e9, "whole_path_frac": value * B_ALG / 1e9 / HBM_PEAK_GBS,
                         "note": "achieved/frac follow the contract: the named (largest-total-time) kernel's k-mers per launch x 17.15 B / its "
                                 "average launch time; that kernel is one stage of a multi-kernel path, so whole_path_* (k-mers/s of the "
                                 "whole job x 17.15 B) is the number to compare with the 8 TB/s peak",
                         "gups_atomic_add": gups.get("atomic_add"), "gups_atomic_cas": gups.get("atomic_cas"),
                         "value_over_gups": value / world / gups["atomic_cas"] if gups.get("atomic_cas") else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            ns = min(args.cpu_sample_reads, n_reads)
            sample = bytes(buf[: ns * stride].cpu().numpy())
            with tempfile.TemporaryDirectory() as td:
                base, ref_stats = cpu_baseline(sample, ns, K, td)
            out["cpu_baseline"] = base
            if ref_stats is not None:       # bit-exactness spot check on the very sample the CPU counted
                with capi.Table(K, 1 << 28, canonical=True, device=local_rank) as t2:
                    t2.count_ascii_dev(buf.data_ptr(), ns * stride)
                    t2.sync()
                    s2 = t2.stats()
                mine = "Unique:    %d\nDistinct:  %d\nTotal:     %d\nMax_count: %d\n" % (s2.unique, s2.distinct, s2.total, s2.max_count)
                out["cpu_baseline"]["stats_equal_on_sample"] = (mine == ref_stats)
                assert mine == ref_stats, "GPU and reference disagree on the sample:\n%s\n%s" % (mine, ref_stats)
        print(json.dumps(out), flush=True)
    t.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
