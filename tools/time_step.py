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

args = [v for v in sys.argv[1:] if v not in ("mlp", "bx3", "v1")]
K, H, a, steps = (int(v) for v in (args[:4] + ["65536", "64", "3", "200"][len(args):]))
mlp = None
if "mlp" in sys.argv:  # SURVEY §8d synthetic 2x256 MLP
    rng = np.random.default_rng(0)
    dims = [3 * a, 256, 256, 2 * a]
    W = [(rng.uniform(-1, 1, (dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    b = [(rng.uniform(-1, 1, dims[i + 1]) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    W[2] *= 0.1
    b[2] *= 0.1
    mlp = dict(W=W, b=b)
h = m.Handle(k=K, tau=H, s_dim=2 * a, a_dim=a, dt=0.1, lam=1.0, sigma=0.25 * np.eye(a), goal=([1, 0, .5, 0, .75, 0, .25, 0])[:2 * a], mlp=mlp,
             mlp_bf16x3="bx3" in sys.argv, tuning=({"mlp_v1": 1} if "v1" in sys.argv else None))
print(h.rollout_kernel_name())
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
if mlp is not None:
    fl = 2.0 * (3 * a * 256 + 256 * 256 + 256 * 2 * a) * K * H
    if "bx3" in sys.argv:
        print("MLP bf16x3: %.1f algorithmic TFLOP/s = %.1f executed bf16 TFLOP/s in the rollout kernel (bf16 MFMA peak ~2500)" % (
            fl / (r * 1e-3) / 1e12, 3 * fl / (r * 1e-3) / 1e12))
    else:
        print("MLP: %.1f TFLOP/s in the rollout kernel (fp32 MFMA peak 157.3)" % (fl / (r * 1e-3) / 1e12))
print("%-40s K=%d H=%d a=%d  wall/step %.1f us | rollout kernel %.1f us | combine+finish %.1f us | %.3g rollouts/s" % (
    os.path.basename(os.environ.get("MPPI_SO_PATH", "libmppi_hip.so")), K, H, a, wall * 1e6, r * 1e3, f * 1e3, K / wall))
