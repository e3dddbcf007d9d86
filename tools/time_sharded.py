#!/usr/bin/env python3
"""What the all-gather fallback costs per control step on ONE GPU (BASELINE configs[2] per-GPU shape, K=65536 H=64):
unsharded mppi_next_device against the sharded path with one rank (partial -> all_gather -> finish), as one rank of an
RCCL group of one. (r02: replaying the three as captured graphs measured 34.7 us against 28.2 us enqueued one by one —
hipGraphLaunch costs more than the three enqueues it saves — and was dropped.)
Usage: python tools/time_sharded.py [K H steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def run(ctrl, n):
    for _ in range(50):
        ctrl.next(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ctrl.next(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


out["unsharded_us"] = run(ShardedController(**cfg), steps)
os.environ["MPPI_FORCE_EXCHANGE"] = "1"
c = ShardedController(exchange="rccl", **cfg)
out["rccl_one_rank_us"] = run(c, steps)
# normalizeCost: unsharded (two passes of the rollout kernel) against the sharded form with its second, 2-float collective
os.environ.pop("MPPI_FORCE_EXCHANGE")
out["normalize_unsharded_us"] = run(ShardedController(normalize_cost=True, **cfg), steps)
os.environ["MPPI_FORCE_EXCHANGE"] = "1"
out["normalize_rccl_one_rank_us"] = run(ShardedController(normalize_cost=True, **cfg), steps)
print(json.dumps(out))
dist.destroy_process_group()
