#!/usr/bin/env python3
"""Host-synchronous closed-loop latency of mppi_next(x)->u (the reference's host-loop shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mppi_tf_amd as m
K, H, a = (int(v) for v in (sys.argv[1:4] + ["65536", "64", "3"][len(sys.argv) - 1:]))
h = m.Handle(k=K, tau=H, s_dim=2 * a, a_dim=a, sigma=0.25 * np.eye(a), goal=([1, 0, .5, 0, .75, 0, .25, 0])[:2 * a])
x = np.zeros(2 * a, np.float32)
ts = []
for i in range(320):
    t0 = time.perf_counter(); u = h.next(x); t1 = time.perf_counter()
    if i >= 20: ts.append(t1 - t0)
    for j in range(a):
        x[2 * j] += 0.1 * x[2 * j + 1] + 0.005 * u[j]; x[2 * j + 1] += 0.1 * u[j]
ts = np.sort(ts) * 1e6
print("K=%d H=%d a=%d sync mppi_next: median %.1f us  p95 %.1f us  min %.1f us | |x-goal| %.3f" % (K, H, a, np.median(ts), ts[int(.95 * len(ts))], ts[0], np.linalg.norm(x - np.array(([1, 0, .5, 0, .75, 0, .25, 0])[:2 * a]))))
