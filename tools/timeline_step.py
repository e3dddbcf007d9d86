#!/usr/bin/env python3
"""Phase timeline of k_step_pc (the fused step; timing study). Needs the MPPI_PC_TIMELINE variant:
   python -c "import mppi_tf_amd.build as b; b.build_variant('timeline', ['MPPI_PC_TIMELINE'])"
   MPPI_SO_PATH=build/variants/libmppi_hip_timeline.so python tools/timeline_step.py [K H a]
Every wave of a tile stamps s_memrealtime (100 MHz: 10 ns) at its phase boundaries, the consumer writes the tile's 64 stamps where its
costs go; column wave 0 leaves four stamps in the handle's debug words. All stamps share one clock: tiles can be laid side by side."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mppi_tf_amd as m

K, H, a = (int(v) for v in (sys.argv[1:4] + ["4096", "64", "2"][len(sys.argv) - 1:]))
h = m.Handle(k=K, tau=H, s_dim=2 * a, a_dim=a, dt=0.1, lam=1.0, sigma=0.25 * np.eye(a), goal=([1, 0, .5, 0, .75, 0, .25, 0])[:2 * a])
print(h.rollout_kernel_name())
x = torch.zeros(2 * a, device="cuda")
u = torch.zeros(a, device="cuda")
for _ in range(30):
    h.next_device(x.data_ptr(), u.data_ptr())
h.synchronize()
c = h.debug_get(m.DBG_COSTS).reshape(-1, 64).astype(np.float64)
aux = h.debug_get(m.DBG_AUX).astype(np.float64)
names = {0: "consumer start", 1: "consumer has x, constants", 9: "tile soft-min stored"}
for ch in range(4):
    names[2 + ch] = "consumed chunk %d" % ch
for p in range(5):
    names[10 + 8 * p] = "P%d start" % p
    for i in range(4):
        names[10 + 8 * p + 1 + i] = "P%d published chunk %d" % (p, i)
    names[10 + 8 * p + 5] = "P%d got weights" % p
    names[10 + 8 * p + 6] = "P%d record stored" % p
t_first = min(v for row in c for v in row[:50] if v > 0)
for b in sorted({0, 1, c.shape[0] // 2, c.shape[0] - 1}):
    row = c[b][:50]
    print("---- tile %d (us since the first stamp of any tile)" % b)
    for v, n in sorted((v, names.get(i, "slot %d" % i)) for i, v in enumerate(row) if v > 0):
        print("  %7.2f  %s" % ((v - t_first) / 100.0, n))
clk = [100.0 * ((row[61] - row[60]) % (1 << 24)) / max((row[9] - row[0]) % (1 << 24), 1.0) for row in c]
print("shader clock during the tiles (s_memtime against s_memrealtime): %.0f .. %.0f MHz" % (min(clk), max(clk)))
ends = [max(row[:50]) for row in c]
print("tiles: last record stored between %.2f and %.2f us after the first stamp" % ((min(ends) - t_first) / 100.0, (max(ends) - t_first) / 100.0))
print("column wave 0: start %.2f, sentinel seen %.2f, sweep complete %.2f, U' stored %.2f us" % tuple((v - t_first) / 100.0 for v in aux[2:6]))
