#!/usr/bin/env python3
"""Phase timeline of one k_rollout_pc workgroup (timing study). Needs the MPPI_PC_TIMELINE variant:
   python -c "import mppi_tf_amd.build as b; b.build_variant("timeline", ["MPPI_PC_TIMELINE"])"
   MPPI_SO_PATH=build/variants/libmppi_hip_timeline.so python tools/timeline.py [K H a]
Every wave stamps s_memtime at its phase boundaries; the consumer writes the 64 stamps where the tile's costs go."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mppi_tf_amd as m

K, H, a = (int(v) for v in (sys.argv[1:4] + ["65536", "64", "3"][len(sys.argv) - 1:]))
# (the stamps live in k_rollout_pc: the two-launch step; its slot map holds 3 producers — a small K, whose default is 5, needs pc_producers 3)
tuning = {"fused_step": 0}
if (K + 63) // 64 <= 512:
    tuning["pc_producers"] = 3
h = m.Handle(k=K, tau=H, s_dim=2 * a, a_dim=a, dt=0.1, lam=1.0, sigma=0.25 * np.eye(a), goal=([1, 0, .5, 0, .75, 0, .25, 0])[:2 * a], tuning=tuning)
x = torch.zeros(2 * a, device="cuda")
u = torch.zeros(a, device="cuda")
for _ in range(5):
    h.next_device(x.data_ptr(), u.data_ptr())
h.synchronize()
c = h.debug_get(m.DBG_COSTS).reshape(-1, 64).astype(np.float64)
GHZ = float(os.environ.get("SCLK_GHZ", "2.4"))
names = {0: "consumer start", 1: "chunk0 published"}
for ch in range(7):
    names[2 + ch] = "consumed chunk %d" % ch
names[9] = "epilogue done"
for p in range(3):
    names[16 + 10 * p + 9] = "P%d start" % p
    for i in range(7):
        names[16 + 10 * p + i] = "P%d produced chunk %d" % (p, i)
    names[16 + 10 * p + 7] = "P%d got weights" % p
    names[16 + 10 * p + 8] = "P%d wsum stored" % p
starts = []
for b in [0, 1, 8, 255, 256, 511, 512, 1023]:
    if b >= c.shape[0]:
        continue
    row = c[b][:62]
    t0 = min(v for v in row if v > 0)
    starts.append((b, t0))
    if b in (0, 512):
        print("---- workgroup %d (us since its first stamp, SCLK %.1f GHz assumed)" % (b, GHZ))
        ev = sorted((v, names.get(i, "slot %d" % i)) for i, v in enumerate(row) if v > 0)
        for v, n in ev:
            print("  %7.2f  %s" % ((v - t0) / (GHZ * 1e3), n))
# placement: slot 48+role (role 0 = consumer, 1.. = producers) holds HW_ID[15:0] | XCC_ID << 16
hw = c[:, 48:52].astype(np.int64)
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
xcc = (hw >> 16) & 15
cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
from collections import defaultdict
percu = defaultdict(list)
for b in range(c.shape[0]):
    assert len(set(cukey[b])) == 1
    percu[int(cukey[b, 0])].append(b)
print("CUs used: %d ; workgroups per CU: %s" % (len(percu), sorted(set(len(v) for v in percu.values()))))
print("block -> (xcc, se, sh, cu) of blocks 0..9, 256: " + " ".join("%d:(%d,%d,%d,%d)" % (b, xcc[b, 0], se[b, 0], sh[b, 0], cu[b, 0]) for b in list(range(10)) + [256]))
print("SIMD of roles (consumer, P0, P1, P2) for blocks 0, 1, 256, 512, 768: " + " ".join(str(tuple(simd[b])) for b in (0, 1, 256, 512, 768) if b < c.shape[0]))
load = defaultdict(lambda: np.zeros((4, 2), int))
for b in range(c.shape[0]):
    for r in range(4):
        load[int(cukey[b, 0])][simd[b, r], 0 if r == 0 else 1] += 1
pat = defaultdict(int)
for k, v in load.items():
    pat[tuple(sorted((int(a), int(bb)) for a, bb in v))] += 1
print("per-CU (consumers, producers) on its 4 SIMDs, sorted -> number of CUs:")
for k, v in sorted(pat.items(), key=lambda kv: -kv[1])[:8]:
    print("   ", k, v)
c[:, 48:52] = 0
rt0, rt1 = c[:, 62] * 0.01, c[:, 63] * 0.01  # s_memrealtime, 10 ns ticks -> us (comparable across CUs)
c = c[:, :62]
base = rt0.min()
print("workgroup START after the first one (us): " + " ".join("p%d=%.2f" % (q, np.percentile(rt0 - base, q)) for q in (0, 10, 50, 90, 99, 100)))
print("consumer epilogue END after first start  : " + " ".join("p%d=%.2f" % (q, np.percentile(rt1 - base, q)) for q in (0, 10, 50, 90, 99, 100)))
print("start->epilogue per workgroup            : " + " ".join("p%d=%.2f" % (q, np.percentile(rt1 - rt0, q)) for q in (0, 10, 50, 90, 99, 100)))
worst = np.argsort(rt1)[-5:]
for b in worst:
    k = int(cukey[b, 0])
    print("  late block %d ends %.2f: CU shares blocks %s, SIMD loads (cons, prod) %s" % (b, rt1[b] - base, percu[k], load[k].tolist()))
nb = c.shape[0]
for lo in range(0, nb, max(1, nb // 8)):
    sl = slice(lo, lo + max(1, nb // 8))
    print("  blocks %5d..%5d: start %.2f..%.2f  end %.2f..%.2f" % (lo, lo + max(1, nb // 8) - 1, (rt0[sl] - base).min(), (rt0[sl] - base).max(), (rt1[sl] - base).min(), (rt1[sl] - base).max()))
ends = c.max(axis=1)
first = np.array([min(v for v in r if v > 0) for r in c])
print("workgroup duration (first to last stamp): median %.2f us, min %.2f, max %.2f" % tuple(
    x / (GHZ * 1e3) for x in (np.median(ends - first), (ends - first).min(), (ends - first).max())))
