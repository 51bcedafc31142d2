#!/usr/bin/env python3
"""bench.py -- k-mers/s of the `jellyfish count` hot path on MI355X (BASELINE.json metric).

  --config C2 (default; the metric's configuration, BASELINE configs[1]): k=21 canonical, 10 Gbp of synthetic 150 bp
           reads per GPU (iid uniform bases, the distribution of the reference's generate_sequence), 2^34-slot table.
  --config C3 (configs[2]): k=31, Bloom-counter pass (`jellyfish bc -s 10G -f 0.001`: m = 14e10 cells = 28 GB, 10
           hashes) followed by the filtered count (`count --bc`) over the same reads, 2^33-slot table.
  --config C5 (configs[4]): k=63 (two-word keys, 128-bit slots), 2^33-slot table.
The reads are generated on the device before the timed region: `value` is throughput with the input resident in HBM.

  step   = one batch (1/K of the reads) through encode -> canonical -> GF(2) hash -> route/insert
  N = 1  : per step the batch is encoded, hashed and radix-partitioned on the device (P1); the last step's sync applies
           everything pending (P2 partition + LDS-resident tile insert), all inside the timed region.  C3 runs K steps of
           the Bloom pass, its flush, then K steps of the filtered count and its flush.
  N > 1  : (C2) one process per GPU, table sharded by the top hash bits; per step route by owner (HIP) -> exchange of the
           routed k-mers (RCCL ncclSend/ncclRecv over xGMI, under the C ABI: jfgpu_comm_*) -> insert (HIP), pipelined by
           one step.  torch.distributed (gloo) only hands out the RCCL id and reduces the timings.  Weak scaling: every
           rank brings its own reads.

  --gpus N > 1 without WORLD_SIZE in the environment: this script starts its own N rank processes (one per GPU, rendezvous on
           127.0.0.1); under an external launcher (torch.distributed.run) it is one of the ranks.  Per GPU the work is 12.5 Gbp
           (BASELINE configs[3]: 100 Gbp over 8 GPUs), the same at every N > 1: weak scaling.

Prints ONE JSON line (rank 0).  `value` comes from the contract's timed region (K steps, once); `repeats` re-runs the
same job a few more times (median/min/max), `flush_sweep` forces 1/2/4/8 flushes per job, `end_to_end` is the Counting
phase of `jellyfish-amd count` from a FASTA file of the same reads (host read + H2D + device parse included), `roofline`
and `cpu_baseline` are the contract's extra objects (algorithmic bytes per k-mer from SURVEY 8(d); the reference's own
CPU path, oracle/_ref, timed on this box's host cores on a bounded sample, its table's content digest compared with the
engine's on the same sample), `secondary` carries BASELINE configs[2] and configs[4] (C3, C5: one job each, run after the
metric's own work in child processes of this script).  The oracle is only baseline / checker here.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READ_LEN = 150
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: 8 TB/s spec
CONFIGS = {
    #        k   lsize  slot B  algorithmic bytes per k-mer occurrence
    "C2": dict(k=21, lsize=34, slot=8, name="BASELINE configs[1]: k=21 -C, {gbp:.1f} Gbp of 150 bp reads per GPU, 2^{lsize}-slot table per GPU in HBM ({slot_bytes}-byte slots)"),
    "C3": dict(k=31, lsize=33, slot=8, name="BASELINE configs[2]: k=31 -C, Bloom-counter pass (m = 14 x {gbp:.0f}e9 cells, 10 hashes) then count --bc, {gbp:.1f} Gbp of 150 bp reads, 2^{lsize}-slot table"),
    "C5": dict(k=63, lsize=33, slot=16, name="BASELINE configs[4]: k=63 -C (two-word keys), {gbp:.1f} Gbp of 150 bp reads, 2^{lsize}-slot 128-bit table in HBM"),
    # not a BASELINE configuration: the plain count at the k most people use (8-byte items and slots), as a secondary of the default line
    "K31": dict(k=31, lsize=33, slot=8, gbp=5.0, name="k=31 -C plain count (no filter), {gbp:.1f} Gbp of 150 bp reads, 2^{lsize}-slot table ({slot_bytes}-byte slots)"),
}
C4_NAME = ("BASELINE configs[3]: k=21 -C, {total:.1f} Gbp of 150 bp reads hash-prefix partitioned across {world} GPUs ({gbp:.1f} Gbp and a 2^{lsize}-slot "
           "shard per GPU, {slot_bytes}-byte slots), routed k-mers exchanged by RCCL over xGMI")
GBP_PER_GPU_SHARDED = 12.5                       # configs[3]: 100 Gbp over 8 GPUs; kept at every N > 1 (weak scaling)
# What the three-stage design (P1 -> P2 -> T) cannot go below, in HBM bytes per k-mer occurrence of C2: the input (1.15), the
# 4-byte item written by P1, read and written by P2 and read by T (16), and the table written once -- 2^34 4-byte slots for
# 8.67 G occurrences, i.e. 7.9 B per occurrence at load 0.5.  (Round 3 quoted 9.15: it left out P2's read + write, T's read
# and the empty half of the table.)
DESIGN_MIN_BYTES = {"C2": 150.0 / 130.0 + 4 * 4.0 + 4.0 * 2 ** 34 / (10e9 / 150 * 130)}


def b_alg(cfg, k):
    seq = READ_LEN / (READ_LEN - k + 1)
    if cfg == "C3":      # Bloom pass: input + 10 cells x (1 B read + 1 B write); count pass: input + slot RMW
        return {"bc": seq + 20.0, "count": seq + 16.0}
    return {"count": seq + 2.0 * CONFIGS[cfg]["slot"]}


def kernel_sources_sha():
    """sha256 over the engine's device and host sources (jellyfish_amd/csrc): what a committed PMC traffic file must have
    been taken on to be quoted (tools/make_traffic_json.py records the same figure)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "jellyfish_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".inl")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def release_device(held_bytes=0):
    """Wait until the driver is done with what this process has just freed.  The amdgpu driver wipes VRAM when it is
    released, in the background, at about 25 GB/s; a process that allocates while that is going on waits for it -- round
    3's end-to-end line charged the CLI's Init 5.4 s for the 165 GB this script had just freed (tools/init_after_parent.py:
    Init 6.6 s right after the parent's free, 0.2 s after a pause; from a fresh shell 0.2 - 1.1 s,
    profiles/r04_cli_writing.log).  Everything this script holds on the device is plain hipMalloc memory behind the C ABI
    (table, workspace, reads), freed explicitly by the caller before this is called: no hipDeviceReset under live
    objects (round-4 advisor finding).  There is no call to ask the driver, so the wait is the freed bytes at 20 GB/s."""
    import ctypes
    try:
        ctypes.CDLL("libamdhip64.so").hipDeviceSynchronize()
    except OSError:
        pass
    wait = min(20.0, held_bytes / 20e9)
    time.sleep(wait)
    return {"waited_s": wait, "held_GB": held_bytes / 1e9}


def write_fasta(arr, path):
    """(n_reads, READ_LEN + 1) uint8 device-layout reads (bases + one 'N') -> FASTA with one-line records."""
    import numpy as np
    n = arr.shape[0]
    hdr = np.tile(np.frombuffer(b">r\n", dtype=np.uint8), (n, 1))
    body = arr.copy()
    body[:, READ_LEN] = ord("\n")
    np.concatenate([hdr, body], axis=1).tofile(path)


def _all_cpus():
    """preexec_fn for the child processes: numpy / torch may have narrowed this process's CPU affinity (their threading
    runtimes pin the main thread on some builds) and children inherit it -- the CLI's 32 read streams or the reference's 64
    worker threads would then share a few cores."""
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except OSError:
        pass


def ref_count(ref, fa_list, k, size, threads, tmpdir, extra=(), digest=None):
    out, timing = os.path.join(tmpdir, "ref.jf"), os.path.join(tmpdir, "timing")
    dg = ["--digest", digest, "--no-write"] if digest else []
    subprocess.check_call([ref, "count", "-m", str(k), "-C", "-s", str(size), "-t", str(threads), "-o", out, "--timing", timing] + dg + list(extra) + fa_list, preexec_fn=_all_cpus)
    t = dict(l.split() for l in open(timing).read().splitlines())
    return float(t["Counting"]), int(t["Mers"]), out


def read_digest(path):
    """(records, sum of counts, sum of h, xor of h) as `ref_jf count --digest` / `jellyfish-amd count --digest` write it."""
    return tuple(int(l.split()[1]) for l in open(path).read().splitlines())


def cpu_baseline(cfg, sample, k, tmpdir, job_load=0.5):
    """Reference CPU path (oracle/_ref: the reference's own classes, SSE2 hash like its configure enables) on a bounded
    sample of the same reads, its table presized to the timed job's load factor (SURVEY 8(d) / BASELINE.md 3.2).  Returns
    (dict for the JSON line, content digest of the reference's in-memory table after counting the sample -- walked by its
    own iterators, `ref_jf count --digest` -- or None)."""
    import numpy as np
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_jf")
    n_reads = sample.shape[0]
    kmers = n_reads * (READ_LEN - k + 1)
    if not os.access(ref, os.X_OK):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        n_small = min(n_reads, 100000)
        t0 = time.time()
        O.count(sample[:n_small].tobytes(), k, True)
        dt = time.time() - t0
        return ({"value": n_small * (READ_LEN - k + 1) / dt, "unit": "k-mers/s", "cores": 1, "kind": "port",
                 "sample": "first %d reads through oracle/jf_oracle.c (sort+count restatement)" % n_small}, None)
    fa = os.path.join(tmpdir, "sample.fa")
    write_fasta(sample, fa)
    ncpu = os.cpu_count() or 1
    best_t = min(ncpu, 64)                       # measured on 2 x EPYC 9575F: -t 64 beats -t 256 (profiles/r02_call2_*.log)

    def presize(n_kmers):                        # smallest power of two that keeps the load at or below the job's (1 % tolerance)
        size = 1
        while n_kmers > size * job_load * 1.01:
            size <<= 1
        return size
    size = presize(kmers)
    res = {"unit": "k-mers/s", "kind": "reference"}
    if cfg == "C3":
        bc = os.path.join(tmpdir, "ref.bc")
        t0 = time.time()
        subprocess.check_call([ref, "bc", "-m", str(k), "-C", "-s", str(n_reads * READ_LEN), "-t", str(best_t), "-o", bc, fa], preexec_fn=_all_cpus)
        t_bc = time.time() - t0
        dgp = os.path.join(tmpdir, "ref.digest")
        t_cnt, mers, out = ref_count(ref, [fa], k, size, best_t, tmpdir, ["--bc", bc], digest=dgp)
        assert mers == kmers
        stats = read_digest(dgp)
        res.update({"value": kmers / (t_bc + t_cnt), "cores": best_t, "bc_pass_kmers_per_s": kmers / t_bc, "count_bc_pass_kmers_per_s": kmers / t_cnt,
                    "sample": "first %d reads (%.0f Mbp) of the same input: jellyfish 2.3.1 classes (oracle/_ref, SSE2 hash), bc (whole command, "
                              "-s %d -f 0.001) then count --bc (Counting phase), -t %d" % (n_reads, n_reads * READ_LEN / 1e6, n_reads * READ_LEN, best_t)})
        return res, stats
    dgp = os.path.join(tmpdir, "ref.digest")
    t_best, mers, out = ref_count(ref, [fa], k, size, best_t, tmpdir, digest=dgp)
    assert mers == kmers
    stats = read_digest(dgp)
    res.update({"value": kmers / t_best, "cores": best_t, "sample_gbp": n_reads * READ_LEN / 1e9, "table_log2": size.bit_length() - 1, "load": kmers / size, "job_load": job_load,
                "sample": "first %d reads (%.2f Gbp) of the same synthetic input, jellyfish 2.3.1 classes (oracle/_ref, -DHAVE_SSE -msse2), -t %d, table presized "
                          "2^%d = load %.2f (the timed job's: %.2f), Counting phase only (SURVEY 8(d): >= 1 Gbp at the job's load factor; the reference's table "
                          "for it is %.1f GB as it packs entries)" % (n_reads, n_reads * READ_LEN / 1e9, best_t, size.bit_length() - 1, kmers / size, job_load, size * 25.0 / 8 / 1e9)})
    if cfg == "C2":
        variants = {"one_file": {"kmers_per_s": kmers / t_best, "threads": best_t, "sample_reads": n_reads, "load": kmers / size}}
        # -F 4 on the same sample, same table: the single serial parser is the reference's bottleneck at high thread counts
        parts = []
        for i in range(4):
            p = os.path.join(tmpdir, "q%d.fa" % i)
            write_fasta(sample[n_reads * i // 4: n_reads * (i + 1) // 4], p)
            parts.append(p)
        tf, mf, of = ref_count(ref, parts, k, size, best_t, tmpdir, ["-F", "4", "--no-write"])
        variants["F4"] = {"kmers_per_s": mf / tf, "threads": best_t, "files": 4, "sample_reads": n_reads, "load": mf / size}
        # one thread and every hardware thread: on a 165 Mbp cut of the sample (a table of its own at the same load) -- minutes otherwise
        n_small = min(n_reads, 1100000)
        fas = os.path.join(tmpdir, "small.fa")
        write_fasta(sample[:n_small], fas)
        ks = n_small * (READ_LEN - k + 1)
        n1 = max(1, n_small // 10)
        fa1 = os.path.join(tmpdir, "s1.fa")
        write_fasta(sample[:n1], fa1)
        t1, m1, o1 = ref_count(ref, [fa1], k, presize(n1 * (READ_LEN - k + 1)), 1, tmpdir, ["--no-write"])
        variants["t1"] = {"kmers_per_s": m1 / t1, "threads": 1, "sample_reads": n1, "load": m1 / presize(m1)}
        if ncpu > best_t:
            ta, ma, oa = ref_count(ref, [fas], k, presize(ks), ncpu, tmpdir, ["--no-write"])
            variants["t_nproc"] = {"kmers_per_s": ma / ta, "threads": ncpu, "sample_reads": n_small, "load": ma / presize(ks)}
        res["variants"] = variants
        full = {nm: v for nm, v in variants.items() if v["sample_reads"] == n_reads}
        res["value"] = max(v["kmers_per_s"] for v in full.values())     # the reference at its best on the 8(d) sample
        res["value_is"] = "best of the runs on the whole sample: %s (t1 / t_nproc, on smaller cuts, are in `variants` only)" % ", ".join(sorted(full))
    return res, stats


def spawn_ranks(n):
    """python bench.py --gpus N from a bare shell: N copies of this command, one rank per GPU, rendezvous on 127.0.0.1 (what
    torch.distributed.run would set up).  Rank 0's stdout (the JSON line) is this process's; a failing rank ends the others,
    and so does SIGTERM / SIGINT to this process.  A box with fewer than N GPUs gets the inter-process test transport (ranks
    share devices, hipIpc* copies between them) at a size that fits: a functional run of the N-rank path, not a scaling
    number -- the JSON line says so.  JFGPU_BENCH_RANK_TIMEOUT (seconds, default 1500): ranks still running then are asked
    where they are (SIGUSR1 -> faulthandler) and killed."""
    import signal
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    extra = {}
    if os.environ.get("JFGPU_COMM_TRANSPORT") != "ipc":
        try:
            import torch
            if torch.cuda.device_count() < n:
                extra = {"JFGPU_COMM_TRANSPORT": "ipc", "JFGPU_BENCH_SHARED_DEVICES": "1"}
        except Exception:
            pass
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **extra)
        procs.append(subprocess.Popen([sys.executable] + sys.argv, env=env, stdout=None if r == 0 else subprocess.DEVNULL))

    def stop_all(*_):
        for q in procs:
            if q.poll() is None:
                q.kill()
        sys.exit(143)
    signal.signal(signal.SIGTERM, stop_all)
    signal.signal(signal.SIGINT, stop_all)
    deadline = time.time() + float(os.environ.get("JFGPU_BENCH_RANK_TIMEOUT", "1500"))
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is None:
                continue
            live.discard(r)
            if c != 0 and rc == 0:
                rc = c
                for q in live:
                    procs[q].terminate()
        if live and time.time() > deadline:
            sys.stderr.write("bench.py: ranks %s still running at the time limit\n" % sorted(live))
            for q in live:
                procs[q].send_signal(signal.SIGUSR1)
            time.sleep(3)
            for q in live:
                procs[q].kill()
            return 124
        time.sleep(0.2)
    return rc


def plan(args):
    """The workload of `bench.py --gpus N --config C` without running it: which BASELINE configuration, how much sequence,
    which table and which device memory per rank.  The arithmetic main() does on the GPU box, in one place that a CPU test
    can call (tests/test_bench_launcher.py)."""
    world = args.gpus
    sb = (world - 1).bit_length()
    if 1 << sb != world:
        raise SystemExit("the number of GPUs must be a power of two (shards = top hash bits)")
    cfg = args.config
    if world > 1 and cfg != "C2":
        raise SystemExit("--config %s is a single-GPU configuration (BASELINE.json)" % cfg)
    gbp = args.gbp or (GBP_PER_GPU_SHARDED if world > 1 else CONFIGS[cfg].get("gbp", 10.0))
    K = CONFIGS[cfg]["k"]
    lsize = args.lsize or CONFIGS[cfg]["lsize"]
    n_reads = int(round(gbp * 1e9 / READ_LEN))
    stride = READ_LEN + 1
    slot_bytes = 4 if (cfg == "C2" and 2 * K - (lsize + sb) <= 10) else CONFIGS[cfg]["slot"]      # kmer_core.hpp: 32-bit slots when at most ten key bits are left to store
    table = (1 << lsize) * slot_bytes
    reads = n_reads * stride
    kmers = n_reads * (READ_LEN - K + 1)
    item = 4 if cfg == "C2" else (8 if cfg in ("C3", "K31") else 16)
    workspace = int(kmers * item * (2.3 if world == 1 else 3.4))       # P1 regions + P2 regions (+ routed regions and what arrived, sharded)
    exchange = None
    if world > 1:
        # What a step puts on the wires (abi_comm.inl: item path).  A rank routes its step's k-mers into 1024 regions of `cap`
        # 4-byte items -- (owner, the owner's coarse bucket) -- and sends every other owner its 1024 / W regions WHOLE (fixed
        # capacity: mean x 1.03 for the per-byte estimate x 1.03 head-room + two stranded granules per workgroup and bucket),
        # one message per peer and step over that peer's own xGMI link; its own share does not move.
        items_step = kmers / args.steps
        cap = int(items_step / 1024 * 1.03 * 1.03 + 2 * 256 * 64)      # (JFGPU_COMM_SLACK = 0.03 twice: on the k-mers-per-byte estimate and on the mean; round 5: 1.10 x 1.10)
        send = (world - 1) / world * 1024 * cap * 4
        per_link = send / (world - 1)
        link = 76.5e9        # one direction of one xGMI link: the guide's ~153 GB/s per link, halved (an assumption until a run measures it)
        # one GPU's stage times per Gbp of input through the sharded code path (profiles/r06_final_bench_C2_sharded_path_one_gpu.json)
        per_gbp_ms = {"route_p1": 2.94, "receive_split": 1.41, "p2_partition": 1.96, "tile_insert": 2.57}
        exchange = {"items_per_step_per_rank": int(items_step), "region_capacity_items": cap, "item_bytes_per_step_per_rank": int(items_step * 4),
                    "send_bytes_per_step_per_rank": int(send), "bytes_per_link_per_step": int(per_link), "link_GB_per_s_assumed": link / 1e9,
                    "exchange_ms_per_step": per_link / link * 1e3,
                    "compute_ms_per_step_measured_on_one_gpu": {k: v * gbp / args.steps for k, v in per_gbp_ms.items()},
                    "note": "the exchange of step i runs beside the routing of step i + 1 and the split of step i - 1 (events, own stream): "
                            "it is hidden while exchange_ms_per_step stays below route_p1 + receive_split of a step"}
    return {"id": cfg if world == 1 else "C4", "world": world, "shard_bits": sb, "k": K, "gbp_per_gpu": gbp, "total_gbp": gbp * world,
            "reads_per_gpu": n_reads, "kmers_per_gpu": kmers, "table_slots_per_gpu": 1 << lsize, "global_table_slots": 1 << (lsize + sb), "slot_bytes": slot_bytes,
            "scaling": "weak", "steps": args.steps, "warmup": args.warmup,
            "launch": "one process per GPU (torch.distributed.run or bench.py's own launcher), RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 from the environment",
            "exchange": None if world == 1 else "jfgpu_comm_count_ascii_dev per step: route by hash prefix, ncclSend / ncclRecv of 4-byte items grouped for the receiver, insert",
            "exchange_estimate": exchange,
            "hbm_bytes_per_gpu": {"table": table, "reads": reads, "workspace_estimate": workspace, "total_estimate": table + reads + workspace},
            "workload": (CONFIGS[cfg]["name"].format(gbp=gbp, lsize=lsize, slot_bytes=slot_bytes) if world == 1 else
                         C4_NAME.format(total=gbp * world, world=world, gbp=gbp, lsize=lsize, slot_bytes=slot_bytes))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2")
    ap.add_argument("--gbp", type=float, default=0.0, help="giga-bases of reads per GPU (default: 10 at N = 1, 12.5 at N > 1)")
    ap.add_argument("--lsize", type=int, default=0, help="log2 slots per GPU (default: the configuration's)")
    ap.add_argument("--cpu-sample-reads", type=int, default=0,
                    help="reads of the CPU-baseline sample; 0 = SURVEY 8(d)'s protocol: at least 1 Gbp of the same reads, as many as put the reference's "
                         "power-of-two table at the timed job's own load factor (C2: 8.26 M reads = 1.24 Gbp into 2^31 slots, load 0.50)")
    ap.add_argument("--matrix", choices=["xs", "reference"], default="xs",
                    help="hash matrix family of the table (include/jfgpu.h: JFGPU_MATRIX_*): xs = the xor-shift matrix the partition kernel evaluates in "
                         "registers; reference = the matrix `jellyfish count` itself draws (file bodies byte-identical to the reference's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeats", type=int, default=5, help="how many times the whole job is run in all (first = the contract's timed region)")
    ap.add_argument("--no-extras", action="store_true", help="skip flush sweep, end-to-end and the secondary configurations (quick runs, profiling)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C5 / C3 jobs attached to the default C2 line")
    ap.add_argument("--as-secondary", action="store_true", help=argparse.SUPPRESS)      # child of the default run: one job, compact line
    ap.add_argument("--dist", choices=["U", "G"], default="U",
                    help="U: iid uniform reads (the metric's configuration); G: BASELINE.md's secondary distribution, reads sampled from a "
                         "100 Mbp random genome with 1 %% substitutions (about 100x coverage at 10 Gbp: most k-mers repeat)")
    ap.add_argument("--plan", action="store_true", help="print the run's workload as JSON (what every rank would be given) and exit: no GPU needed")
    args = ap.parse_args()
    if args.plan:
        print(json.dumps(plan(args)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if args.as_secondary:
        args.no_extras = True; args.no_cpu_baseline = True
        if "--repeats" not in sys.argv:
            args.repeats = 1
    # ranks started by an external launcher (torch.distributed.run) on a box with fewer GPUs than ranks: what spawn_ranks
    # arranges for its own children -- the inter-process test transport, devices shared, a size that fits (the line says so)
    if args.gpus > 1 and "WORLD_SIZE" in os.environ and os.environ.get("JFGPU_COMM_TRANSPORT") != "ipc":
        import torch
        if torch.cuda.device_count() < args.gpus:
            os.environ["JFGPU_COMM_TRANSPORT"] = "ipc"
            os.environ["JFGPU_BENCH_SHARED_DEVICES"] = "1"
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    shared = 1                                       # ranks per device (> 1: fewer GPUs than ranks, see spawn_ranks)
    if os.environ.get("JFGPU_BENCH_SHARED_DEVICES") == "1":
        import torch
        shared = -(-args.gpus // max(torch.cuda.device_count(), 1))
    if not args.gbp:
        args.gbp = GBP_PER_GPU_SHARDED if args.gpus > 1 else CONFIGS[args.config].get("gbp", 10.0)
        if shared > 1:                               # what fits `shared` ranks' tables, inputs and workspaces in one HBM
            args.gbp = 10.0 / shared
            if not args.lsize:
                args.lsize = CONFIGS[args.config]["lsize"] - (shared - 1).bit_length()

    import faulthandler
    import signal
    faulthandler.register(signal.SIGUSR1, all_threads=True)     # kill -USR1 <pid>: where a stuck run is waiting
    import numpy as np
    from jellyfish_amd import capi
    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU: there is no CPU path to measure"
    emu = False

    cfg = args.config
    K = CONFIGS[cfg]["k"]
    lsize = args.lsize or CONFIGS[cfg]["lsize"]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1 and cfg != "C2":
        raise SystemExit("--config %s is a single-GPU configuration (BASELINE.json)" % cfg)
    if os.environ.get("JFGPU_COMM_TRANSPORT") == "ipc":     # the inter-process test transport: all ranks on the devices there are
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")          # control plane only; the data path is the engine's own RCCL communicator
    sb = (world - 1).bit_length()
    assert 1 << sb == world, "the number of GPUs must be a power of two (shards = top hash bits)"

    def device_sync():
        torch.cuda.synchronize()

    n_reads = int(round(args.gbp * 1e9 / READ_LEN))
    steps, warmup = args.steps, args.warmup
    stride = READ_LEN + 1
    kmers_per_read = READ_LEN - K + 1
    t = capi.Table(K, 1 << (lsize + sb), canonical=True, device=local_rank, shard_bits=sb, shard_id=rank, matrix_kind=args.matrix)
    lsize = t.info.lsize - sb                        # the engine may raise a size below the slot format's minimum
    t_info_xs = bool(t.matrix_is_xorshift())
    slot_bytes = t.info.slot_bytes
    buf = t.malloc(n_reads * stride + 16)            # plain device memory through the C ABI
    device_sync()
    if args.dist == "G":
        t.gen_genome_reads_dev(buf, rank * n_reads, n_reads, READ_LEN, 100_000_000, 0.01, 42)
    else:
        t.gen_reads_dev(buf, rank * n_reads, n_reads, READ_LEN, 42)
    t.sync()

    def reads_to_host(a, b):                         # reads [a, b) as a (b - a, stride) uint8 array
        return t.d2h(buf + a * stride, (b - a) * stride).reshape(b - a, stride)
    ns = args.cpu_sample_reads
    if not ns and cfg != "C2":
        ns = 1100000                                 # (C3 / C5 run as secondaries without a CPU leg; asked for directly: 165 Mbp)
    if not ns:      # "the first 1 Gbp at identical table load factor" -- the smallest power-of-two table that holds >= 1 Gbp at the job's load
        job_load = n_reads * kmers_per_read / float(1 << lsize)
        sz = 1
        while sz * job_load < 1e9 / READ_LEN * kmers_per_read:
            sz <<= 1
        ns = int(sz * job_load / kmers_per_read)
    ns = min(ns, n_reads)
    sample = reads_to_host(0, ns) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    bounds = [n_reads * i // steps for i in range(steps + 1)]

    def batch(i):
        i %= steps
        return buf + bounds[i] * stride, (bounds[i + 1] - bounds[i]) * stride

    # random-access roofline denominator on this very table (dirty afterwards -> cleared)
    gups = {}
    if rank == 0 and cfg != "C5":
        for mode, name in ((0, "atomic_add"), (2, "atomic_cas")):
            gups[name] = t.gups(1 << 28, mode)
    t.clear()
    force_dist = os.environ.get("JFGPU_BENCH_FORCE_DIST") == "1"     # exercise the N>1 code path on one GPU
    sharded = world > 1 or force_dist
    comm = None
    if sharded:
        ids = [capi.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        comm = capi.Comm(world, rank, ids[0], device=local_rank)
        try:        # RCCL prints a version banner through C stdio: push it out now, so that the JSON line is the last thing on stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    bloom = None
    if cfg == "C3":
        bloom = capi.Bloom(K, capi.opt_m(0.001, int(args.gbp * 1e9)), capi.opt_k(0.001), canonical=True, device=local_rank)
        # routing workspace of the Bloom pass: the more, the fewer flushes stream the 28 GB array (150 GB: three).  On
        # distribution G the filtered count admits most k-mers and its own flush needs ~64 GB beside table, filter and reads
        bloom.reserve(int(os.environ.get("JFGPU_BENCH_BC_WS_GB", "150" if args.dist == "U" else "100")) << 30)
    # Init phase: workspace for one sync-to-sync span (like -s presizes the table).  C3: the filtered pass admits next to
    # nothing of a uniform input, and the Bloom pass needs the memory: two batches' worth, flushed as it fills
    if cfg == "C3":
        t.reserve(2 * ((n_reads + steps - 1) // steps) * stride)
    else:
        t.reserve(n_reads * stride if not sharded else int(n_reads * kmers_per_read * 1.02) + (1 << 20))

    exchanged = [0, 0]

    def fence():
        if sharded:
            exchanged[0], exchanged[1] = comm.finish()     # last step's exchange + insert; running totals sent / received
        t.sync()
        device_sync()
        if world > 1:
            dist.barrier()

    def job(n_steps, flushes=1, wall=None):
        """n_steps batches through the whole path of this configuration (both passes for C3), applied and synchronised."""
        if cfg == "C3":
            t.attach_bloom(None)
            for i in range(n_steps):
                p, n = batch(i)
                bloom.insert_ascii_dev(p, n)
            bloom.sync()                 # flush of the Bloom pass: end of `jellyfish bc`
            t.attach_bloom(bloom)
        for i in range(n_steps):
            if sharded:
                p, n = batch(i)
                comm.step(t, p, n)
                if wall is not None:
                    wall.append(time.perf_counter())
            else:
                p, n = batch(i)
                t.count_ascii_dev(p, n)
            if flushes > 1 and i + 1 < n_steps and (i + 1) * flushes // n_steps > i * flushes // n_steps:
                t.sync()
        fence()

    def reset():
        t.clear()
        if bloom is not None:
            t.attach_bloom(None)
            bloom.clear()

    # N ranks: the exchange alternates between two sets of buffers (turns); a warm-up of one step would leave the second set's
    # first use -- allocations, the peers' mappings or registrations of them -- inside the timed region.  The warm-up is W
    # steps as asked, and at least two when the path is sharded (untimed either way).
    job(max(warmup, 2) if sharded else warmup)
    reset()
    t.profile_enable(True); t.profile_reset()
    if bloom is not None:
        bloom.profile_enable(True); bloom.profile_reset()
    fence()
    if comm is not None:
        comm.exchange_times()                        # (drops what the warm-up logged)
    step_wall = []                                   # host time at which every step's call returned (the calls only enqueue: a step that blocks shows here)
    t0 = time.perf_counter()
    job(steps, wall=step_wall)
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    t.profile_enable(False)
    if bloom is not None:
        bloom.profile_enable(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- N-rank runs: what every rank did in every step, so that one record says where a slow run loses its time ----
    per_rank = None
    if sharded:
        spans = t.profile_spans()
        xt = comm.exchange_times()
        names = {1: "split", 2: "route", 4: "p1", 5: "p2", 6: "tile", 7: "direct"}
        sums = {}
        for w_, ms_ in spans:
            nm = names.get(w_, "slot%d" % w_)
            sums[nm] = sums.get(nm, 0.0) + ms_
        cw, cr = comm.world_rank()
        mine = {"rank": rank, "comm_world": cw, "comm_rank": cr, "device": local_rank, "elapsed_s": round(my_elapsed, 4),
                "device_ms": {k_: round(v_, 2) for k_, v_ in sorted(sums.items())},
                "route_ms_per_launch": [round(ms_, 2) for w_, ms_ in spans if w_ == 2],
                "split_ms_per_launch": [round(ms_, 2) for w_, ms_ in spans if w_ == 1],
                "exchange_ms_per_step": [round(a, 2) for a, b_ in xt], "exchange_wire_MB_per_step": [round(b_ / 1e6, 1) for a, b_ in xt],
                "exchange_GB_per_s": [round(b_ / 1e9 / (a * 1e-3), 1) if a > 0 else None for a, b_ in xt],
                "step_call_returned_at_ms": [round((w_ - t0) * 1e3, 1) for w_ in step_wall],
                "transport": os.environ.get("JFGPU_COMM_TRANSPORT", "rccl")}
        if world > 1:
            gathered = [None] * world if rank == 0 else None
            dist.gather_object(mine, gathered, dst=0)
            per_rank = gathered
        else:
            per_rank = [mine]

    # size-independent invariants of the timed job: every window was seen exactly once; unfiltered runs counted them all
    st = t.stats()
    tot = [st.total, st.distinct, st.mers_fed]
    if world > 1:
        tt = torch.tensor(tot + exchanged, dtype=torch.int64)
        dist.all_reduce(tt)
        tot = [int(x) for x in tt.tolist()]
        assert tot[3] == tot[4], "k-mers lost or duplicated in the exchange: sent %d, received %d" % (tot[3], tot[4])
    total_kmers = n_reads * kmers_per_read * world
    if cfg == "C3":
        assert bloom.sync() == total_kmers and int(tot[2]) == total_kmers, "k-mers fed: bc %d, count %d, expected %d" % (bloom.sync(), int(tot[2]), total_kmers)
    else:
        assert int(tot[0]) == total_kmers, "counted %d k-mers, expected %d" % (int(tot[0]), total_kmers)
    digest = t.digest() if world == 1 else None
    ctrs = t.counters()

    out = None
    if rank == 0:
        slot_names = ["count_direct", "add_keys", "shard_partition", "lookup", "p1_partition", "p2_partition", "tile_insert", "items_direct"]
        if sharded:     # the exchange's item path: slot 2 = P1 over the global table (sender), slot 1 = split of what arrived (receiver)
            slot_names[1], slot_names[2] = "exchange_receive_split", "exchange_route_p1"
        B = b_alg(cfg, K)
        kernels = {}
        for i, nm in enumerate(slot_names):
            kms, kl, ku = t.profile_get(i)
            if kl:
                kernels[nm] = {"ms": round(kms, 3), "launches": kl, "units": ku, "bytes_per_kmer": B["count"]}
        if bloom is not None:
            for i, nm in enumerate(("bc_direct", "bc_p1_route", "bc_p2_partition", "bc_segments")):
                kms, kl, ku = bloom.profile_get(i)
                if kl:
                    kernels[nm] = {"ms": round(kms, 3), "launches": kl, "units": ku, "bytes_per_kmer": B["bc"]}
        dom = max(kernels, key=lambda nm: kernels[nm]["ms"])
        ms, launches = kernels[dom]["ms"], kernels[dom]["launches"]
        per_launch_kmers = total_kmers / world / max(launches, 1)
        avg_ms = ms / max(launches, 1)
        bpk = kernels[dom]["bytes_per_kmer"]
        achieved = per_launch_kmers * bpk / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        value = total_kmers / elapsed
        whole_bpk = sum(B.values())          # C3: both passes touch every k-mer
        # HBM bytes per launch of the named kernel: PMC counters cannot be read from inside this process, so the figure
        # comes from the committed rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this same command (profiles/r03_traffic_<cfg>.json,
        # made by tools/profile_bench.sh); quoted whenever workload and kernel match, whatever --steps is (traffic per
        # job does not depend on how the input is cut into batches -- it is scaled to this run's launch count)
        traffic, traffic_src = None, None
        tj = os.path.join(ROOT, "profiles", "r06_traffic_%s%s.json" % (cfg, "" if args.dist == "U" else "_G"))
        if os.path.exists(tj) and world == 1 and not force_dist:
            rec = json.load(open(tj))
            if rec.get("kernel_sources_sha256") != kernel_sources_sha():
                # counters taken on other kernels say nothing about these: refuse them rather than quote a stale figure
                traffic_src = "%s was taken on different kernel sources (sha %s..., now %s...): not quoted" % (
                    os.path.relpath(tj, ROOT), str(rec.get("kernel_sources_sha256"))[:12], kernel_sources_sha()[:12])
            elif abs(rec.get("gbp", 0) - args.gbp) < 1e-9 and rec.get("lsize") == lsize and dom in rec.get("per_job_bytes", {}):
                traffic = rec["per_job_bytes"][dom] / max(launches, 1)
                traffic_src = "%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, same command, same kernel sources): bytes per job / %d launches" % (os.path.relpath(tj, ROOT), launches)
                # every stage's measured HBM traffic next to its live time: the rate each stage really runs at
                for nm, kv in kernels.items():
                    if nm in rec["per_job_bytes"] and kv.get("ms"):
                        kv["hbm_traffic_bytes_per_job"] = rec["per_job_bytes"][nm]
                        kv["hbm_GB_per_s"] = rec["per_job_bytes"][nm] / (kv["ms"] * 1e-3) / 1e9
        out = {
            "metric": "k-mers/sec at k=%d canonical, 150 bp synthetic reads, bit-exact counts" % K,
            "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64" if CONFIGS[cfg]["slot"] == 8 else "u128", "data": "synthetic",
            "config": {"workload": (CONFIGS[cfg]["name"].format(gbp=args.gbp, lsize=lsize, slot_bytes=slot_bytes) if world == 1 else
                                    C4_NAME.format(total=args.gbp * world, world=world, gbp=args.gbp, lsize=lsize, slot_bytes=slot_bytes)) +
                                   ("" if args.dist == "U" else "; SECONDARY distribution G (reads from a 100 Mbp random genome, 1 % substitutions)"),
                       "matrix": {"xs": "xor-shift family (jfgpu.h JFGPU_MATRIX_XORSHIFT): evaluated in registers by the partition kernel; any matrix with an invertible low block is "
                                        "legal for the file format, readers take it from the header", "reference": "the reference's own draw (JFGPU_MATRIX_REFERENCE): byte-identical file bodies"}[args.matrix]
                                  if t_info_xs == (args.matrix == "xs") else "reference family (the xor-shift one is defined for one-word keys)",
                       "id": cfg if world == 1 else "C4", "k": K, "read_len": READ_LEN, "reads_per_gpu": n_reads, "table_slots_per_gpu": 1 << lsize, "slot_bytes": slot_bytes,
                       "load_factor": float(tot[1]) / float(world << lsize),
                       "distinct": int(tot[1]), "total_kmers": total_kmers,
                       "parallelism": ("single GPU" if not force_dist else "single GPU through the sharded code path") if world == 1 else
                                      "hash-prefix shard x%d + all-to-all" % world + ("" if shared == 1 else
                                      "; %d RANKS PER DEVICE over the inter-process test transport (hipIpc* copies): a functional run of the N-rank path on a box with fewer GPUs, not a scaling number" % shared)},
            "kernels": kernels,
            "content_digest": digest,
            "tile_kernel": {"flushes_plain": ctrs["flushes_plain"], "flushes_heavy": ctrs["flushes_heavy"], "direct_inserts": ctrs["direct"],
                            "what": "instantiation of the tile stage chosen per flush from its own sample (plain / HEAVY for high-coverage input), and items "
                                    "the partition kernels inserted with global atomics (region or ring overflow, runs of one k-mer)"},
            "roofline": {"bound": "hbm", "kernel": dom,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "bytes_per_kmer": bpk, "kmers_per_launch": per_launch_kmers,
                         "avg_launch_ms": avg_ms, "launches": launches,
                         "whole_path_achieved": value * whole_bpk / 1e9, "whole_path_frac": value * whole_bpk / 1e9 / HBM_PEAK_GBS,
                         "note": "achieved/frac follow the contract: the named (largest-total-time) kernel's k-mers per launch x its algorithmic "
                                 "bytes per k-mer / its average launch time; that kernel is one stage of a multi-kernel path, so whole_path_* "
                                 "(k-mers/s of the whole job x the path's algorithmic bytes) is the number to compare with the 8 TB/s peak.  "
                                 "bytes_per_kmer is SURVEY 8(d)'s contract figure (one read-modify-write of an 8-byte slot for one-word keys) "
                                 "whatever the slot width in use (config.slot_bytes); design_min_bytes_per_kmer is what this three-stage design "
                                 "cannot go below (input + the 4-byte item written by P1, read and written by P2, read by T + every slot of the table written once)",
                         "design_min_bytes_per_kmer": DESIGN_MIN_BYTES.get(cfg),
                         "per_rank": per_rank,
                         "gups_atomic_add": gups.get("atomic_add"), "gups_atomic_cas": gups.get("atomic_cas"),
                         "value_over_gups": value / world / gups["atomic_cas"] if gups.get("atomic_cas") else None},
        }
        if cfg == "C3":
            bc_ms = sum(v["ms"] for nm, v in kernels.items() if nm.startswith("bc_"))
            cnt_ms = sum(v["ms"] for nm, v in kernels.items() if not nm.startswith("bc_"))
            out["passes"] = {"bc_device_ms": bc_ms, "count_bc_device_ms": cnt_ms, "admitted_kmers": int(tot[0]),
                             "bc_kmers_per_s_device": total_kmers / (bc_ms * 1e-3) if bc_ms else None,
                             "count_bc_kmers_per_s_device": total_kmers / (cnt_ms * 1e-3) if cnt_ms else None}

    # ---- repeats of the whole job (outside the contract's region; same work, same checks) ----
    vals = [total_kmers / elapsed]
    for r in range(max(0, args.repeats - 1)):
        reset(); fence()
        t1 = time.perf_counter()
        job(steps)
        dt = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        vals.append(total_kmers / dt)
        if world == 1 and digest is not None:
            assert t.digest() == digest, "repeat %d produced a different table" % (r + 1)
    if rank == 0:
        out["repeats"] = {"n": len(vals), "kmers_per_s": vals, "median": statistics.median(vals), "min": min(vals), "max": max(vals),
                          "note": "first entry = the contract's timed region (value); the others re-run the identical job after a clear, table digest equal every time"}

    # ---- the CPU-baseline sample counted on the GPU, IN THE TIMED TABLE (after a clear): same geometry, same slot width, so
    # the kernels that produced `value` -- ring P1, loader / storer P2, the 4-byte-slot tile kernel for C2 -- are the ones whose
    # result is compared with the reference's digest further down (round-4 review: the sample used to go into a 2^28 side
    # table with 8-byte slots).  Which kernels ran is read from the engine's counters and reported.  Taken here, while the
    # reads are still in device memory.  (Through the sharded code path the sample goes into a side table as before.) ----
    sample_digest, sample_kernels = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        b2 = None
        if cfg == "C3":      # the reference's `bc -s <sample bases>`: a filter of the sample's own size
            b2 = capi.Bloom(K, capi.opt_m(0.001, ns * READ_LEN), capi.opt_k(0.001), canonical=True, device=local_rank)
            b2.insert_ascii_dev(buf, ns * stride)
            b2.sync()
        if not force_dist:
            reset()
            if b2 is not None:
                t.attach_bloom(b2)
            t.set_mode(2)                    # (a flush of so few items into so large a table would otherwise be inserted item by item with global atomics)
            t.count_ascii_dev(buf, ns * stride)
            t.sync()
            t.set_mode(0)
            sample_digest = tuple(t.digest())
            c2 = t.counters()
            sample_kernels = {nm: c2[nm] for nm in ("p1_ring", "p1_other", "p2_roles", "p2_ring", "p2_sort", "p2_exact", "flushes_plain", "flushes_heavy", "direct")}
            sample_kernels["table"] = "the timed table: 2^%d slots of %d bytes" % (lsize, slot_bytes)
            if cfg == "C2" and lsize >= 33 and slot_bytes == 4:       # did the timed job's own kernels take the sample?  (reported, not asserted: the contract line must come out)
                sample_kernels["timed_kernels_took_it"] = bool(c2["p1_ring"] >= 1 and c2["p1_other"] == 0 and c2["p2_roles"] >= 1 and c2["p2_sort"] + c2["p2_exact"] == 0
                                                               and c2["flushes_plain"] + c2["flushes_heavy"] >= 1)
            t.attach_bloom(None)
        else:
            with capi.Table(K, 1 << 28, canonical=True, device=local_rank) as t2:
                if b2 is not None:
                    t2.attach_bloom(b2)
                t2.count_ascii_dev(buf, ns * stride)
                t2.sync()
                sample_digest = tuple(t2.digest())
                t2.attach_bloom(None)
        if b2 is not None:
            b2.close()

    if rank == 0 and world == 1 and not force_dist and not args.no_extras and cfg == "C2":
        # ---- how much of the rate depends on flushing once: forced flushes inside the job ----
        sweep = {}
        for f in (1, 2, 4, 8):
            if f > steps:
                break
            reset(); fence()
            t1 = time.perf_counter()
            job(steps, flushes=f)
            sweep[str(f)] = total_kmers / (time.perf_counter() - t1)
            assert t.digest() == digest
        out["flush_sweep"] = {"kmers_per_s_by_flushes_per_job": sweep,
                              "note": "a flush streams every dirty tile of the table once, so its cost is O(table), not O(batch)"}
        # ---- end to end: `jellyfish-amd count` from a FASTA file of the same reads (page cache warm, /dev/shm) ----
        cli = os.path.join(ROOT, "bin", "jellyfish-amd")
        shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
        if os.access(cli, os.X_OK):
            reset()
            with tempfile.TemporaryDirectory(dir=shm) as td:
                fa = os.path.join(td, "reads.fa")
                chunk = 8_000_000
                with open(fa, "wb") as fh:                       # the same reads, one-line FASTA records (">r\n" + 150 bases + "\n")
                    for a in range(0, n_reads, chunk):
                        b = min(n_reads, a + chunk)
                        arr = reads_to_host(a, b)
                        hdr = np.tile(np.frombuffer(b">r\n", dtype=np.uint8), (b - a, 1))
                        arr[:, READ_LEN] = ord("\n")
                        np.concatenate([hdr, arr], axis=1).tofile(fh)
                fbytes = os.path.getsize(fa)
                t.free(buf); buf = None
                t.close()                                         # the CLI needs the device memory ...
                if bloom is not None:
                    bloom.close(); bloom = None
                released = release_device((1 << lsize) * slot_bytes + n_reads * stride * (1 + 2 * 4.3))      # table + reads + workspace
                tim, dg, outp = os.path.join(td, "timing"), os.path.join(td, "digest"), os.path.join(td, "out.jf")
                env = dict(os.environ, JFGPU_QUIET="1")
                t1 = time.perf_counter()
                subprocess.check_call([cli, "count", "-m", str(K), "-C", "-s", str(1 << lsize), "-o", outp, "--timing", tim, "--digest", dg,
                                       "--device", str(local_rank), fa], env=env, preexec_fn=_all_cpus)
                wall = time.perf_counter() - t1
                tm = dict(l.split() for l in open(tim).read().splitlines())
                dgl = tuple(int(l.split()[1]) for l in open(dg).read().splitlines())
                assert dgl == digest, "CLI run and HBM-resident run disagree: %r vs %r" % (dgl, digest)
                obytes = os.path.getsize(outp)
                rec_bytes = (2 * K + 7) // 8 + 4
                hdr_len = 9 + int(open(outp, "rb").read(9))
                assert obytes == hdr_len + digest[0] * rec_bytes, "output file: %d bytes, expected header %d + %d records of %d bytes" % (obytes, hdr_len, digest[0], rec_bytes)
                os.unlink(outp)
                cs, ws = float(tm["Counting"]), float(tm["Writing"])
                out["end_to_end"] = {"parent_affinity_cpus": len(os.sched_getaffinity(0)), "init_s": float(tm["Init"]), "counting_s": cs, "writing_s": ws, "process_wall_s": wall,
                                     "file_bytes": fbytes, "file_GB_per_s": fbytes / cs / 1e9, "kmers_per_s": total_kmers / cs,
                                     "output_bytes": obytes, "output_GB_per_s": obytes / ws / 1e9, "kmers_per_s_wall": total_kmers / wall,
                                     "digest_equal_to_resident_run": True, "device_released_before": released,
                                     "what": "`jellyfish-amd count -o <file>` (output ENABLED) from a fresh process on a %.1f GB FASTA file of the same reads in /dev/shm: "
                                             "Init / Counting / Writing as count_main.cc:375-382 reports them; Counting = host read, host->device copy, device parse, count "
                                             "(PCIe-inclusive, never `value`); Writing = device sort + copy + %.1f GB of records into one file on /dev/shm (the file system's "
                                             "single-file write rate is the limit: profiles/r04_cli_writing.log)" % (fbytes / 1e9, obytes / 1e9)}
                out["roofline"]["end_to_end"] = {"counting_s": cs, "kmers_per_s": total_kmers / cs, "writing_s": ws, "init_s": float(tm["Init"]),
                                                 "file_GB_per_s": fbytes / cs / 1e9, "output_GB_per_s": obytes / ws / 1e9,
                                                 "what": "`jellyfish-amd count` from a FASTA file in /dev/shm, Counting phase PCIe- and parse-inclusive (SURVEY 8(d)'s first number; never `value`)"}
            t = None

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        with tempfile.TemporaryDirectory() as td:
            base, ref_stats = cpu_baseline(cfg, sample, K, td, job_load=n_reads * kmers_per_read / float(1 << lsize))
        out["cpu_baseline"] = base
        if ref_stats is not None:       # bit-exactness on the very sample the CPU counted: per-k-mer content, not aggregates
            mine = sample_digest
            out["cpu_baseline"]["digest_equal_on_sample"] = (mine == tuple(ref_stats))
            out["cpu_baseline"]["sample_counted_by"] = sample_kernels      # launches per partition kernel: the timed job's own, in the timed table
            out["cpu_baseline"]["sample_digest"] = {"records": mine[0], "total": mine[1], "sum_h": mine[2], "xor_h": mine[3],
                                                    "what": "content digest of the whole table (records, sum of counts, sum and xor of a per-record hash of key words and "
                                                            "count): jfgpu_digest on the device table vs `ref_jf count --digest` on the reference's in-memory table"}
            assert mine == tuple(ref_stats), "GPU and reference disagree on the sample: %r vs %r" % (mine, ref_stats)
    # ---- BASELINE configs[4] and configs[2]: one job each, in children of this script (they need the device memory) ----
    if rank == 0 and world == 1 and cfg == "C2" and not args.no_extras and not args.no_secondary and not args.as_secondary and args.dist == "U":
        if t is not None:
            if buf is not None:
                t.free(buf); buf = None
            t.close(); t = None
        if bloom is not None:
            bloom.close(); bloom = None
        release_device()                                          # the children need the device memory (they are not timed on their Init)
        sec = {}
        # C5, C3: BASELINE configs[4] and [2] on the metric's uniform reads; C2_G, C3_G: the same engine on BASELINE.md's secondary
        # distribution (reads from a 100 Mbp genome, 1 % substitutions, ~100 x coverage), each job run twice with the table
        # digest asserted equal (the check that found the tile stage's race in round 3)
        for name, c, extra in (("C5", "C5", []), ("C3", "C3", []), ("C2_G", "C2", ["--dist", "G", "--repeats", "2"]), ("C3_G", "C3", ["--dist", "G", "--repeats", "2"]),
                               ("K31", "K31", []), ("C2_reference_matrix", "C2", ["--matrix", "reference"])):
            try:
                if "--matrix" not in extra:
                    extra = extra + ["--matrix", args.matrix]
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", c, "--as-secondary", "--steps", str(steps), "--warmup", str(warmup)] + extra,
                                   capture_output=True, text=True, timeout=900, preexec_fn=_all_cpus)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode == 0 and line:
                    d = json.loads(line[-1])
                    sec[name] = {k: d[k] for k in ("value", "unit", "ms_per_step", "dtype", "config", "kernels", "roofline", "content_digest", "repeats", "tile_kernel") if k in d}
                    if "passes" in d:
                        sec[name]["passes"] = d["passes"]
                else:
                    sec[name] = {"error": (r.stderr or r.stdout)[-400:]}
            except Exception as e:       # the contract line must come out whatever the extras do
                sec[name] = {"error": repr(e)}
        out["secondary"] = sec
        # the driver's record keeps the `roofline` dict whole but only the names of other top-level keys: the secondaries'
        # numbers, compact, where they survive (round-4 review, item 9)
        out["roofline"]["secondary_values"] = {
            nm: ({"value": v["value"], "whole_path_frac": v.get("roofline", {}).get("whole_path_frac"),
                  "kernel": v.get("roofline", {}).get("kernel"), "frac": v.get("roofline", {}).get("frac")} if "value" in v else {"error": v.get("error", "")[:120]})
            for nm, v in sec.items()}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if bloom is not None:
        bloom.close()
    if t is not None:
        t.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
