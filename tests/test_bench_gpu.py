"""bench.py end to end, every --workload, at small sizes: one JSON line with the contract's fields (the driver depends on it)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def run_bench(*argv):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--min-time", "0", *argv],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # ONE line on stdout
    assert len(lines[0]) < 4096
    return json.loads(lines[0])


@pytest.mark.parametrize("workload,samples,horizon", [("pm1d", 128, 32), ("pm2d", 4096, 64), ("pm3d", 8192, 16), ("mlp", 2048, 8),
                                                      ("mlp32", 4096, 8), ("auv", 4096, 8), ("nnauv", 4096, 8), ("nnspeed", 4096, 8)])
def test_every_workload_prints_the_contract_line(workload, samples, horizon):
    out = run_bench("--workload", workload, "--samples", str(samples), "--horizon", str(horizon), "--no-subrecords", "--no-cpu-baseline")
    for key in CONTRACT:
        assert key in out, key
    assert "rollouts/s" in out["metric"] and out["unit"] == "rollouts/s"
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 2
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert abs(out["value"] - samples / (out["ms_per_step"] * 1e-3)) <= 0.02 * out["value"]
    assert out["config"]["workload"]
    assert out["ms_per_control_step_sync"]["median"] > 0
    if workload.startswith("pm"):  # the opt-in pre-launched pipeline rides beside the headline (never as it): a figure, or why there is none
        pre = out["prelaunched"]
        assert pre.get("ms_per_step", 0) > 0 or "unavailable" in pre, pre
        assert "prelaunched" not in out["config"]["workload"]
    else:
        assert "prelaunched" not in out


@pytest.mark.parametrize("workload,extra", [("pm2d", []), ("auv", ["--samples", "4096", "--horizon", "8"]), ("nnspeed", ["--samples", "4096", "--horizon", "8"])])
def test_cpu_baseline_rides_along(workload, extra):
    out = run_bench("--workload", workload, "--no-subrecords", *extra)
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]


def test_rank_mode_measures_the_allgather_path_first_and_the_direct_exchange_in_children():
    """How the driver runs N > 1: `python -m torch.distributed.run ... bench.py --gpus N` — bench.py IS a rank. The ranks measure the
    all-gather path in-process (that line is safe), then try the direct exchange in CHILD processes with a time limit, and rank 0 prints
    ONE line with both outcomes (first contact with the direct exchange must not cost the curve). Rehearsed on the one GPU
    (MPPI_BENCH_ONE_GPU=1: gloo rendezvous, both ranks and their children on cuda:0 = 4 GPU processes)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MPPI_BENCH_ONE_GPU="1")
    env.pop("MPPI_EXCHANGE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--min-time", "0",
                        "--samples", "8192", "--no-subrecords"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    ex = out["exchange"]
    assert ex["rccl"]["used"] == "rccl" and ex["rccl"]["value"] > 0
    assert isinstance(ex["p2p"], dict) and ex["p2p"]["used"] == "p2p", ex
    assert ex["printed"] in ("rccl", "p2p") and out["n_gpus"] == 2
    assert out["value"] == max(ex["rccl"]["value"], ex["p2p"]["value"]) or abs(out["value"] - max(ex["rccl"]["value"], ex["p2p"]["value"])) < 1e-3 * out["value"]
