#!/usr/bin/env python3
"""End-to-end `jellyfish-amd count` on a generated FASTA / FASTQ file: device parser vs --host-parse.
Usage: tools/cli_feed_bench.py [n_reads] [out_json].  Files go to /dev/shm (page cache speed)."""
import json, os, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bin", "jellyfish-amd")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
out_json = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "cli_feed.json")
ln = 150
rng = np.random.default_rng(1)
tmp = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
res = {"n_reads": n_reads, "read_len": ln}


def write(path, fastq):
    with open(path, "wb") as f:
        step = 500_000
        for a in range(0, n_reads, step):
            m = min(step, n_reads - a)
            bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(m, ln), dtype=np.uint8)]
            hdr = np.frombuffer(b"".join((b"@" if fastq else b">") + b"r%09d\n" % (a + i) for i in range(m)), dtype=np.uint8).reshape(m, -1)
            nl = np.full((m, 1), 10, dtype=np.uint8)
            if fastq:
                plus = np.tile(np.frombuffer(b"+\n", dtype=np.uint8), (m, 1))
                qual = np.full((m, ln), ord("I"), dtype=np.uint8)
                rec = np.concatenate([hdr, bases, nl, plus, qual, nl], axis=1)
            else:
                rec = np.concatenate([hdr, bases, nl], axis=1)
            f.write(rec.tobytes())


for fmt in ("fa", "fq"):
    path = os.path.join(tmp, "feed_bench." + fmt)
    write(path, fmt == "fq")
    size = os.path.getsize(path)
    for mode in ("device", "device_pinned", "host"):
        tm = os.path.join(tmp, "feed_tm")
        cmd = [CLI, "count", "-m", "21", "-C", "-s", "8G" if n_reads > 20_000_000 else "2G", "--no-write", "--timing", tm, path] + (["--host-parse"] if mode == "host" else [])
        t0 = time.time()
        subprocess.check_call(cmd, env=dict(os.environ, JFGPU_TIMING_DETAIL="1", JFGPU_FEED_PINNED="1" if mode == "device_pinned" else "0"))
        wall = time.time() - t0
        t = dict(l.split() for l in open(tm))
        res["%s_%s" % (fmt, mode)] = {"file_bytes": size, "wall_s": round(wall, 3), "counting_s": float(t["Counting"]), "init_s": float(t["Init"]),
                                      "device_parse_s": float(t.get("DeviceParse", 0)),
                                      "file_GBps_counting": round(size / float(t["Counting"]) / 1e9, 3)}
    os.unlink(path)
os.makedirs(os.path.dirname(out_json), exist_ok=True)
json.dump(res, open(out_json, "w"), indent=1)
print(json.dumps(res))
