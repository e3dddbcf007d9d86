"""bench.py --gpus N typed directly: the PARENT (which never touches a GPU) runs the ranks as child jobs — MPPI_EXCHANGE=rccl first,
then auto — each with a time limit, and relays the better line (a fault of the direct exchange must cost a field of the
line, not the line). On a box without a GPU both jobs fail: the parent has to say so, with both outcomes, and exit non-zero promptly."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_parent_reports_both_jobs_and_never_hangs():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU box: the real two-job run is what tests/test_bench_gpu.py and the driver exercise")
    env = dict(os.environ, MPPI_BENCH_BUDGET_S="200")
    env.pop("MPPI_EXCHANGE", None)
    env.pop("MPPI_BENCH_ONE_GPU", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and r.stdout.strip() == ""
    tail = [l for l in r.stderr.splitlines() if "no job produced a line" in l]
    assert tail, r.stderr[-2000:]
    outcome = json.loads(tail[-1].split("no job produced a line: ", 1)[1])
    assert set(outcome) == {"rccl", "auto"}
    assert all(v.startswith("failed: rc=") for v in outcome.values()), outcome
    assert "MPPI_EXCHANGE=rccl" in r.stderr and "MPPI_EXCHANGE=auto" in r.stderr
