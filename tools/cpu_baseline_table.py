#!/usr/bin/env python3
"""CPU baseline table of SURVEY §8d: the CPU restatement (oracle/, a faithful fp32 restatement of the reference's
dataflow; the reference itself needs TensorFlow and is not runnable here) on THIS box's host cores, single-threaded
and with all cores, for BASELINE configs C1-C3 (whole control steps: Philox noise + rollouts + update) and C4 at
reduced K. Measurement aid like bench.py's `cpu_baseline` leg — never part of the product path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc

GOALS = {1: [1, 0], 2: [1, 0, 0, 0], 3: [1, 0, 0.5, 0, 0.75, 0]}


def mlp3():
    rng = np.random.default_rng(0)
    dims = [9, 256, 256, 6]
    W = [(rng.uniform(-1, 1, (dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    b = [(rng.uniform(-1, 1, dims[i + 1]) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    W[2] *= 0.1
    b[2] *= 0.1
    return dict(W=W, b=b)


ALL = min(orc.num_threads(), orc.usable_cpus())


def run(name, K, H, a, threads, budget, mlp=None):
    orc.set_num_threads(1 if threads == 1 else ALL)
    s = 2 * a
    sigma = 0.25 * np.eye(a)
    p = orc.Problem(tau=H, s=s, a=a, dt=0.1, mass=1.0, lam=1.0, sigma=sigma, goal=GOALS[a], threads=threads, mlp=mlp)
    x, U = np.zeros(s, np.float32), np.zeros((H, a), np.float32)
    p.next_with_noise(x, U, orc.noise(1, 0, 0, K, H, a, sigma))
    n, t0 = 0, time.perf_counter()
    while True:
        eps = orc.noise(1, n + 1, 0, K, H, a, sigma)
        _, U, _ = p.next_with_noise(x, U, eps)
        n += 1
        el = time.perf_counter() - t0
        if el > budget or n >= 500:
            break
    print("| %s K=%d H=%d | %s | %.3g ms/step | %.3g rollouts/s | %d steps |" % (
        name, K, H, "1 thread" if threads == 1 else "%d threads" % ALL, 1e3 * el / n, K * n / el, n), flush=True)


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    print("host cores: %d visible, %d usable by this job (affinity / cgroup quota) -> OpenMP threads %d" % (os.cpu_count(), orc.usable_cpus(), ALL))
    for name, K, H, a, mlp in [("C1 point_mass1d", 128, 32, 1, None), ("C2 point_mass2d", 4096, 64, 2, None),
                               ("C3 point_mass3d", 65536, 64, 3, None), ("C4 point_mass3d + MLP (reduced K)", 2048, 64, 3, mlp3())]:
        for threads in (1, 0):
            run(name, K, H, a, threads, budget, mlp)
