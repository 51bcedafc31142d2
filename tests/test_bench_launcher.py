"""bench.py's own rank launcher (python bench.py --gpus N from a bare shell) on a machine without a GPU: the ranks must
fail loudly -- there is no CPU path to measure -- and the launcher must come back with their exit code instead of waiting
for ranks that are gone."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_a_gpu_fails_loudly_and_returns():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the launcher is covered by the -m gpu runs")
    env = dict(os.environ, JFGPU_BENCH_RANK_TIMEOUT="240")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=400)
    assert r.returncode not in (0, 124), r.stderr[-2000:]
    assert "needs a GPU" in r.stderr
    assert r.stdout.strip() == ""                       # no JSON line from a run that measured nothing


def _plan(*argv):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plan"] + list(argv), cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_gpus_8_plan_is_baseline_config_3():
    """`bench.py --gpus 8` is BASELINE configs[3] (k = 21, 100 Gbp hash-prefix partitioned across 8 GPUs): 12.5 Gbp and a
    2^34-slot shard per GPU, weak scaling at every N > 1, and a per-GPU memory budget that fits 288 GB -- the arithmetic
    of the run, checked where no GPU is (the run itself needs eight)."""
    p8 = _plan("--gpus", "8", "--steps", "20", "--warmup", "5")
    assert p8["id"] == "C4" and p8["world"] == 8 and p8["shard_bits"] == 3 and p8["k"] == 21
    assert p8["gbp_per_gpu"] == 12.5 and p8["total_gbp"] == 100.0 and p8["scaling"] == "weak"
    assert p8["table_slots_per_gpu"] == 1 << 34 and p8["global_table_slots"] == 1 << 37 and p8["slot_bytes"] == 4
    assert p8["reads_per_gpu"] == 83333333 and p8["kmers_per_gpu"] == 83333333 * 130
    assert 0.5 < p8["kmers_per_gpu"] / p8["table_slots_per_gpu"] < 0.8          # load factor of a shard
    assert p8["hbm_bytes_per_gpu"]["total_estimate"] < 288e9
    assert "100.0 Gbp" in p8["workload"] and "8 GPUs" in p8["workload"]
    # what a step puts on every link, and what that costs next to the step's own work (round-4 review, item 5)
    x = p8["exchange_estimate"]
    assert x["items_per_step_per_rank"] == p8["kmers_per_gpu"] // 20
    assert x["send_bytes_per_step_per_rank"] == 7 * x["bytes_per_link_per_step"]
    assert 1.0 < x["send_bytes_per_step_per_rank"] / (7 / 8 * x["item_bytes_per_step_per_rank"]) < 1.35       # whole regions travel: head-room on the wire
    assert 0 < x["exchange_ms_per_step"] < sum(x["compute_ms_per_step_measured_on_one_gpu"].values())
    for n in (2, 4):
        pn = _plan("--gpus", str(n))
        assert pn["id"] == "C4" and pn["gbp_per_gpu"] == 12.5 and pn["table_slots_per_gpu"] == 1 << 34 and pn["global_table_slots"] == 1 << (34 + pn["shard_bits"])
    p1 = _plan()
    assert p1["id"] == "C2" and p1["world"] == 1 and p1["gbp_per_gpu"] == 10.0 and p1["slot_bytes"] == 4 and p1["exchange"] is None and p1["exchange_estimate"] is None
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plan", "--gpus", "3"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and "power of two" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plan", "--gpus", "2", "--config", "C5"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and "single-GPU configuration" in r.stderr
