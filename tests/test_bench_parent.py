"""bench.py --gpus N typed directly: the PARENT (which never touches a GPU) runs the ranks as child jobs — the all-gather through
torch.distributed first (the path every earlier round measured), then the direct exchange, then the one-call RCCL path — each with a
time limit, and relays the best line (VERDICT r03 / ADVICE r04: a fault of a path that has never run between two devices must cost a
field of the line, not the line). On a box without a GPU every job fails: the parent has to say so, with every outcome, and exit
non-zero promptly. The merge of the jobs' lines is a pure function, tested here on synthetic lines."""
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
    assert set(outcome) == {"rccl_torch", "p2p", "rccl_c"}
    assert all(v.startswith("failed: rc=") for v in outcome.values()), outcome
    assert "MPPI_EXCHANGE=rccl MPPI_RCCL_CALL=torch" in r.stderr and "MPPI_EXCHANGE=auto" in r.stderr and "MPPI_EXCHANGE=rccl MPPI_RCCL_CALL=c" in r.stderr


def fake_line(value, used, call=None, parity=None, subs=None):
    line = {"metric": "rollouts/s", "value": value, "ms_per_step": 65536 * 8 / value * 1e3, "config": {"exchange": used},
            "exchange": {"used": used, "direct_exchange_bring_up": "self-test passed on 8 rank(s)" if used == "p2p" else "not requested", "rccl_call": call}}
    if parity:
        line["parity"] = parity
    if subs:
        line["sub_records"] = subs
    return line


def test_merge_keeps_the_parity_verdict_and_the_subrecords():
    """whichever job's headline is printed, the line carries a parity verdict, every job's own verdict stays with
    its entry, and the sub-records of the first job survive a later job that ran without them."""
    sys.path.insert(0, ROOT)
    import bench
    par = {"steps": 3, "ranks_bit_identical": True, "sharded_vs_unsharded_max_abs": 1.2e-7}
    got = {"rccl_torch": fake_line(1.9e10, "rccl", "torch.distributed all_gather", par, [{"config": "configs[4]"}]),
           "p2p": fake_line(2.4e10, "p2p", None, dict(par, sharded_vs_unsharded_max_abs=2.4e-7))}
    order = ["rccl_torch", "p2p", "rccl_c"]
    line = bench.merge_job_lines(got, {"rccl_c": "failed: no line within 120 s (ranks killed)"}, order)
    assert line["value"] == 2.4e10 and line["exchange"]["printed"] == "p2p"
    assert line["parity"]["sharded_vs_unsharded_max_abs"] == 2.4e-7 and line["parity"]["ranks_bit_identical"] is True
    assert line["exchange"]["rccl_torch"]["parity"] == par and line["exchange"]["rccl_torch"]["call"] == "torch.distributed all_gather"
    assert line["exchange"]["rccl_c"].startswith("failed") and line["sub_records"] == [{"config": "configs[4]"}]
    # the direct exchange fell back inside its job: said so; a printed job without a verdict of its own borrows the first one that has it
    got2 = {"rccl_torch": fake_line(1.9e10, "rccl", "torch", par), "p2p": fake_line(2.0e10, "rccl", "one C call per step")}
    line2 = bench.merge_job_lines(got2, {}, order)
    assert "direct exchange not used" in line2["exchange"]["p2p"]["note"]
    assert line2["parity"]["measured_by"] == "rccl_torch" and line2["parity"]["ranks_bit_identical"] is True
    json.dumps(line); json.dumps(line2)
