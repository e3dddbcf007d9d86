#!/usr/bin/env python3
"""What the all-gather fallback costs per control step on ONE GPU (BASELINE configs[2] per-GPU shape, K=65536 H=64), with one rank
of an RCCL group of one:
  unsharded          mppi_next_device
  rccl_torch         mppi_shard_partial -> torch.distributed all_gather_into_tensor -> mppi_shard_finish (three ctypes calls + one
                     collective call from Python per step; what r03 measured at 29.9 us)
  rccl_c_call        ONE call per step: mppi_shard_step -> ncclAllGather on the controller's own communicator (r04)
  no_comm            mppi_shard_step with coll = NULL: the sharded sequence of kernels (rollout, record, finish) without any collective
each with the pipelined time per step (2000 steps, one synchronisation) and the HOST time of one step's enqueue (steps issued into an
empty queue, timed before the synchronisation), and the same two numbers from the native host (examples/host_loop_sharded rccl | step).
(r02: replaying the three calls as captured graphs measured 34.7 us against 28.2 us enqueued one by one and was dropped.)
Usage: python tools/time_sharded.py [K H steps]"""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
K, H, steps = (int(v) for v in (sys.argv[1:4] + ["65536", "64", "2000"][len(sys.argv) - 1:]))
dist.init_process_group("nccl", rank=0, world_size=1)
from mppi_tf_amd.distributed import ShardedController

a = 3
cfg = dict(k=K, tau=H, s_dim=6, a_dim=a, dt=0.1, lam=1.0, sigma=0.25 * np.eye(a), goal=[1, 0, .5, 0, .75, 0])
x = torch.zeros(6, device="cuda")
out = {"K": K, "H": H, "steps": steps}


def run(ctrl, n, step=None):
    step = step or (lambda: ctrl.next(x))
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n * 1e6
    hs = []
    for _ in range(5):  # host side alone: 16 steps into an empty queue
        t0 = time.perf_counter()
        for _ in range(16):
            step()
        hs.append((time.perf_counter() - t0) / 16 * 1e6)
        torch.cuda.synchronize()
    return {"us_per_step": round(el, 2), "host_enqueue_us": round(float(np.median(hs)), 2)}


def env(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


for norm, tag in ((False, ""), (True, "normalize_")):
    kw = dict(cfg, normalize_cost=True) if norm else cfg
    env(MPPI_FORCE_EXCHANGE=None, MPPI_RCCL_CALL=None)
    out[tag + "unsharded"] = run(ShardedController(**kw), steps)
    env(MPPI_FORCE_EXCHANGE="1", MPPI_RCCL_CALL="torch")
    c = ShardedController(exchange="rccl", **kw)
    assert c.rccl is None
    out[tag + "rccl_torch"] = run(c, steps)
    env(MPPI_RCCL_CALL=None)
    c = ShardedController(exchange="rccl", **kw)
    assert c.rccl is not None, c.rccl_note
    out[tag + "rccl_c_call"] = run(c, steps)
    st = torch.cuda.current_stream().cuda_stream
    h, u = c.backend.h, c.u
    out[tag + "no_comm"] = run(c, steps, lambda: h.shard_step(x.data_ptr(), u.data_ptr(), None, st))
    del c
dist.destroy_process_group()

# the native host: same step, RCCL called from C++ (three calls) and through mppi_shard_step (one call)
exe = os.path.join(ROOT, "examples", "host_loop_sharded")
if os.path.exists(exe):
    for mode in ("rccl", "step", "p2p"):
        r = subprocess.run([exe, str(K), str(H), "3", "50", mode], capture_output=True, text=True, timeout=300)
        mt = re.search(r"([\d.]+) us per sharded control step.*host enqueue ([\d.]+) us", r.stdout)
        out["native_" + mode] = {"us_per_step": float(mt.group(1)), "host_enqueue_us": float(mt.group(2))} if mt else "failed: " + (r.stderr or r.stdout)[-200:]
print(json.dumps(out))
