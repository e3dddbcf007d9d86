#!/usr/bin/env python3
"""Pipelined time per control step against K on one GPU (H = 64), for the kernels of the bench line: the point mass, the Fossen AUVModel, NNAUVModel, NNAUVModelSpeed
(r04's two-wave pipelines; grids of more than one round take wave-index roles). Run on the GPU box:   python tools/k_scaling.py > profiles/r04_k_scaling.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mppi_tf_amd as m
from mppi_tf_amd.auv import auv_task


def mlp(dims, seed=0):
    r = np.random.default_rng(seed); n = len(dims) - 1
    return dict(W=[(r.uniform(-1, 1, (dims[i], dims[i+1])) / np.sqrt(dims[i]) * (0.1 if i == n-1 else 1)).astype(np.float32) for i in range(n)],
                b=[(r.uniform(-1, 1, dims[i+1]) / np.sqrt(dims[i]) * (0.1 if i == n-1 else 1)).astype(np.float32) for i in range(n)])


pm = dict(tau=64, s_dim=6, a_dim=3, dt=0.1, lam=1.0, sigma=0.25 * np.eye(3), goal=[1, 0, .5, 0, .75, 0])
at = auv_task(64, learned=True); x13 = np.asarray(at.pop("x0"), np.float32)
av = auv_task(64); av.pop("x0")
cases = [("point_mass3d", pm, np.zeros(6, np.float32)), ("Fossen AUVModel rk2", av, x13), ("NNAUVModel Dense(32)x3", dict(nnauv=mlp([16, 32, 32, 32, 13]), **at), x13),
         ("NNAUVModelSpeed Dense(16)x3", dict(nnauv_speed=mlp([15, 16, 16, 16, 6]), **at), x13)]
for name, kw, x0 in cases:
    for K in (4096, 16384, 65536, 131072, 262144, 524288):
        h = m.Handle(k=K, **kw)
        x = torch.tensor(x0, device="cuda"); u = torch.zeros(kw["a_dim"], device="cuda")
        n = 200 if K <= 65536 else 60
        for _ in range(20): h.next_device(x.data_ptr(), u.data_ptr())
        h.synchronize(); t0 = time.perf_counter()
        for _ in range(n): h.next_device(x.data_ptr(), u.data_ptr())
        h.synchronize(); el = (time.perf_counter() - t0) / n
        print("%-28s %-40s K=%7d  %9.1f us per step  %.3g rollouts/s" % (name, h.rollout_kernel_name().replace("mppi::", ""), K, 1e6 * el, K / el), flush=True)
        h.close()
