#!/usr/bin/env python3
"""Phase timeline of the two waves of a k_rollout_nnspeed_pc tile (timing study). Needs the MPPI_PC_GEN_TIMELINE variant:
   python -c "import mppi_tf_amd.build as b; print(b.build_variant('pc_timeline', ['MPPI_PC_GEN_TIMELINE']))"
   MPPI_SO_PATH=build/variants/libmppi_hip_pc_timeline.so python tools/timeline_pc.py
Both waves stamp s_memtime at their phase boundaries in two consecutive steps; the stamps come back where the tile's costs go."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mppi_tf_amd as m
from mppi_tf_amd.auv import auv_task


def mlp(dims, seed=0):
    r = np.random.default_rng(seed); n = len(dims) - 1
    return dict(W=[(r.uniform(-1, 1, (dims[i], dims[i+1])) / np.sqrt(dims[i]) * (0.1 if i == n-1 else 1)).astype(np.float32) for i in range(n)],
                b=[(r.uniform(-1, 1, dims[i+1]) / np.sqrt(dims[i]) * (0.1 if i == n-1 else 1)).astype(np.float32) for i in range(n)])


at = auv_task(64, learned=True); x13 = np.asarray(at.pop("x0"), np.float32)
h = m.Handle(k=65536, nnauv_speed=mlp([15, 16, 16, 16, 6]), **at)
x = torch.tensor(x13, device="cuda"); u = torch.zeros(6, device="cuda")
for _ in range(5):
    h.next_device(x.data_ptr(), u.data_ptr())
h.synchronize()
c = h.debug_get(m.DBG_COSTS).reshape(-1, 64).astype(np.float64)
NAMES = {0: "N step start", 1: "N noise + action cost done", 2: "N inputs read", 3: "N three layers issued", 4: "N output layer, velocities done", 5: "N step handed over (r04: at the barrier)",
         6: "N after the hand-over (r04: through the barrier)", 8: "P step start", 9: "P cost done", 10: "P pose done", 13: "P Euler angles done, handed over (r04: at the barrier)", 14: "P after the hand-over"}
for tile in (0, 1, 512, 1023):
    row = c[tile][:32]
    t0 = min(v for v in row if v > 0)
    print("---- tile %d (cycles since its first stamp; two steps)" % tile)
    for v, n in sorted((v, "step+%d %s" % (i // 16, NAMES.get(i % 16, "slot %d" % (i % 16)))) for i, v in enumerate(row) if v > 0):
        print("  %7.0f  %s" % ((v - t0) % (1 << 24), n))
