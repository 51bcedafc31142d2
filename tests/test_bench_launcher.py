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
