#!/usr/bin/env python3
"""r05: one shape, every way of running its control step — pipelined (device-resident x) and host-synchronous (mppi_next),
two launches / one fused launch / pre-launched / armed. Usage: python tools/time_modes.py [K H a] [--steps N]
Prints one JSON line per mode (wall per step; kernel time by the launch's own timestamps where profiled)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mppi_tf_amd as m

args = [v for v in sys.argv[1:] if not v.startswith("--")]
K, H, a = (int(v) for v in (args[:3] + ["4096", "64", "2"][len(args):]))
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 400
cfg = dict(k=K, tau=H, s_dim=2 * a, a_dim=a, dt=0.1, lam=1.0, sigma=0.25 * np.eye(a), goal=([1, 0, .5, 0, .75, 0, .25, 0])[:2 * a])


def pipelined(tuning):
    h = m.Handle(tuning=tuning, **cfg)
    x, u = torch.zeros(2 * a, device="cuda"), torch.zeros(a, device="cuda")
    for _ in range(50):
        h.next_device(x.data_ptr(), u.data_ptr())
    h.synchronize()
    walls = []
    for _ in range(15):
        t0 = time.perf_counter()
        for _ in range(steps):
            h.next_device(x.data_ptr(), u.data_ptr())
        h.synchronize()
        walls.append((time.perf_counter() - t0) / steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        h.next_device(x.data_ptr(), u.data_ptr())
    enq = (time.perf_counter() - t0) / steps  # the host's enqueue time per step (the queue absorbs it)
    h.synchronize()
    h.profile_begin(steps)
    for _ in range(steps):
        h.next_device(x.data_ptr(), u.data_ptr())
    h.synchronize()
    r, f, n = h.profile_end()
    name = h.rollout_kernel_name()
    h.close()
    return dict(kernel=name.replace("mppi::", ""), us_per_step=round(float(np.median(walls)) * 1e6, 2), min_us=round(min(walls) * 1e6, 2),
                host_enqueue_us=round(enq * 1e6, 2), kernel_us=round(r * 1e3, 2), finish_us=round(f * 1e3, 2), rollouts_per_s=float("%.4g" % (K / np.median(walls))))


def sync(tuning, n=600):
    h = m.Handle(tuning=tuning, **cfg)
    x = np.zeros(2 * a, np.float32)
    ts = []
    for i in range(n + 40):
        t0 = time.perf_counter()
        u = h.next(x)
        t1 = time.perf_counter()
        if i >= 40:
            ts.append(t1 - t0)
        for j in range(a):
            x[2 * j] += 0.1 * x[2 * j + 1] + 0.005 * u[j]
            x[2 * j + 1] += 0.1 * u[j]
    ts = np.sort(ts) * 1e6
    h.close()
    return dict(median_us=round(float(np.median(ts)), 2), p95_us=round(float(ts[int(.95 * len(ts))]), 2), min_us=round(float(ts[0]), 2))


nb = (K + 63) // 64
out = {"K": K, "H": H, "a": a, "tiles": nb}
print(json.dumps(out))
modes = [("two_launches", {"fused_step": 0})] + ([("fused", {"fused_step": 1})] if nb <= 128 else [])
for name, t in modes:
    print(json.dumps({"pipelined": name, **pipelined(t)}), flush=True)
try:  # the opt-in pre-launched pipeline (MPPI_TUNE_PRELAUNCH): its kernel_us INCLUDES the wait for U' of the step before
    print(json.dumps({"pipelined": "pre-launched", **pipelined({"prelaunch": 1})}), flush=True)
except Exception as e:
    print(json.dumps({"pipelined": "pre-launched", "error": str(e)[:120]}), flush=True)
for name, t in modes:
    print(json.dumps({"sync": name, **sync(t)}), flush=True)
    try:
        print(json.dumps({"sync": name + "+armed", **sync(dict(t, armed_us=500))}), flush=True)
    except Exception as e:
        print(json.dumps({"sync": name + "+armed", "error": str(e)}), flush=True)
