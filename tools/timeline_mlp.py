#!/usr/bin/env python3
"""Per-phase cycle totals of k_rollout_mlp_bx3 (timing study): build the variant with
   python -c "import mppi_tf_amd.build as b; b.build_variant('mlp_tl', ['MPPI_MLP_TIMELINE'])"
   MPPI_SO_PATH=build/variants/libmppi_hip_mlp_tl.so python tools/timeline_mlp.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mppi_tf_amd as m

K, H, a = 65536, 64, 3
rng = np.random.default_rng(0)
dims = [3 * a, 256, 256, 2 * a]
W = [(rng.uniform(-1, 1, (dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
b = [(rng.uniform(-1, 1, dims[i + 1]) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
W[2] *= 0.1
b[2] *= 0.1
h = m.Handle(k=K, tau=H, s_dim=2 * a, a_dim=a, dt=0.1, lam=1.0, sigma=0.25 * np.eye(a), goal=[1, 0, .5, 0, .75, 0],
             mlp=dict(W=W, b=b), mlp_bf16x3=True)
x = torch.zeros(2 * a, device="cuda")
u = torch.zeros(a, device="cuda")
for _ in range(2):
    h.next_device(x.data_ptr(), u.data_ptr())
h.synchronize()
pipe = False
c = h.debug_get(m.DBG_COSTS)
if pipe:  # 128 rollouts per workgroup: the first 64 cost slots of every block hold [wave][phase]; units = half-iterations
    c = c.reshape(-1, 128)[:, :64]
c = c.reshape(-1, 8, 8).astype(np.float64)  # [block][wave][phase] cycles summed over H steps
names = ["fragments -> L1 -> relu/split -> image", "barrier A", "L2 MFMA issue (+ chain of the other set)", "L3 + y write", "barrier B",
         "-", "-", "-"] if pipe else ["noise/action cost/input split", "exchange + L1 MFMA issue", "relu + split + image write", "barrier 1",
         "(noise gen) + L2 MFMA issue", "L3 (waits for MFMA) + y write", "barrier 2", "y reduce + state + cost"]
per = c.mean(axis=(0, 1)) / (2 * H if pipe else H)
print("cycles per %s, mean over blocks and waves (total %.0f = %.2f us at 2.4 GHz):" % ("half-iteration (64 rollouts x 1 step)" if pipe else "horizon step", per.sum(), per.sum() / 2400))
for n, v in zip(names, per):
    print("  %7.0f  %s" % (v, n))
print("per wave (block 0):")
for w in range(8):
    print("  wave %d: " % w + " ".join("%6.0f" % (v / (2 * H if pipe else H)) for v in c[0, w]))
