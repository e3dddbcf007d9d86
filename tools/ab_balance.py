#!/usr/bin/env python3
"""A/B of the two-wave pipeline kernels' role placement (run on the GPU box): MPPI_TUNE_PC_BALANCE = 1 (a wave's role from the SIMD it runs on, flipped
for every second workgroup of a CU: one wave of each kind per SIMD) against 0 (roles by wave index), K = 65536, H = 64, alternating.
r04 measured NO difference (auv 0.1532-0.1545 / 0.1532, nnspeed 0.2182-0.2184 / 0.2193-0.2194, nnauv 0.3050-0.3052 / 0.3054-0.3058 ms per step):
the hardware's own placement of 4-wave workgroups already spreads the two kinds over the SIMDs. The placement code stays (it costs one barrier at kernel
start) because it is what guarantees the property; this script is how to check it on another part.
   python tools/ab_balance.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import mppi_tf_amd as m
from mppi_tf_amd.auv import auv_task
def mlp(dims, seed=0):
    r = np.random.default_rng(seed); n = len(dims) - 1
    return dict(W=[(r.uniform(-1, 1, (dims[i], dims[i+1])) / np.sqrt(dims[i]) * (0.1 if i == n-1 else 1)).astype(np.float32) for i in range(n)],
                b=[(r.uniform(-1, 1, dims[i+1]) / np.sqrt(dims[i]) * (0.1 if i == n-1 else 1)).astype(np.float32) for i in range(n)])
at = auv_task(64, learned=True); x13 = np.asarray(at.pop("x0"), np.float32)
av = auv_task(64); av.pop("x0")
cases = [("auv", dict(k=65536, **av)), ("nnspeed", dict(k=65536, nnauv_speed=mlp([15,16,16,16,6]), **at)), ("nnauv", dict(k=65536, nnauv=mlp([16,32,32,32,13]), **at))]
for name, kw in cases:
    for bal in (1, 0, 1, 0):
        h = m.Handle(tuning={"pc_balance": bal}, **kw)
        x = torch.tensor(x13, device="cuda"); u = torch.zeros(6, device="cuda")
        for _ in range(30): h.next_device(x.data_ptr(), u.data_ptr())
        h.synchronize(); t0 = time.perf_counter()
        for _ in range(300): h.next_device(x.data_ptr(), u.data_ptr())
        h.synchronize(); el = (time.perf_counter() - t0) / 300
        print("%-8s balance=%d  %.4f ms/step" % (name, bal, 1e3 * el), flush=True)
        h.close()
