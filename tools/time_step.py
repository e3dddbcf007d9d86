#!/usr/bin/env python3
"""Time the kernels of one control step with HIP events (mppi_profile_*). Usage:
   [MPPI_SO_PATH=build/variants/libmppi_hip_X.so] python tools/time_step.py [K H a steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mppi_tf_amd as m

K, H, a, steps = (int(v) for v in (sys.argv[1:5] + ["65536", "64", "3", "200"][len(sys.argv) - 1:]))
h = m.Handle(k=K, tau=H, s_dim=2 * a, a_dim=a, dt=0.1, lam=1.0, sigma=0.25 * np.eye(a), goal=([1, 0, .5, 0, .75, 0, .25, 0])[:2 * a])
x = torch.zeros(2 * a, device="cuda")
u = torch.zeros(a, device="cuda")
for _ in range(20):
    h.next_device(x.data_ptr(), u.data_ptr())
h.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    h.next_device(x.data_ptr(), u.data_ptr())
h.synchronize()
wall = (time.perf_counter() - t0) / steps
h.profile_begin(steps)
for _ in range(steps):
    h.next_device(x.data_ptr(), u.data_ptr())
h.synchronize()
r, f, n = h.profile_end()
print("%-40s K=%d H=%d a=%d  wall/step %.1f us | rollout kernel %.1f us | combine+finish %.1f us | %.3g rollouts/s" % (
    os.path.basename(os.environ.get("MPPI_SO_PATH", "libmppi_hip.so")), K, H, a, wall * 1e6, r * 1e3, f * 1e3, K / wall))
