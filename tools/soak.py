#!/usr/bin/env python3
"""Soak run on the GPU box: thousands of consecutive control steps of every rollout-kernel family (point mass with its option paths,
the 2x256 and Dense(32) networks in both precisions, the 13-state family); checks the step counter and finite action sequences.
   python tools/soak.py [multiplier of the step counts]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mppi_tf_amd as m
from mppi_tf_amd.auv import auv_task
rng = np.random.default_rng(0)
def mlp(dims, seed=0):
    r = np.random.default_rng(seed)
    n = len(dims) - 1
    return dict(W=[(r.uniform(-1, 1, (dims[i], dims[i+1])) / np.sqrt(dims[i]) * (0.1 if i == n-1 else 1)).astype(np.float32) for i in range(n)],
                b=[(r.uniform(-1, 1, dims[i+1]) / np.sqrt(dims[i]) * (0.1 if i == n-1 else 1)).astype(np.float32) for i in range(n)])
pm = dict(tau=64, s_dim=6, a_dim=3, dt=0.1, lam=1.0, sigma=0.25*np.eye(3), goal=[1,0,.5,0,.75,0])
pm2 = dict(tau=64, s_dim=4, a_dim=2, dt=0.1, lam=1.0, sigma=0.25*np.eye(2), goal=[1,0,.5,0])
at = auv_task(64, learned=True); x13 = np.asarray(at.pop("x0"), np.float32)
cases = [
 ("pm3d", dict(k=65536, **pm), np.zeros(6, np.float32), 4000),
 ("pm3d normalize", dict(k=65536, normalize_cost=True, **pm), np.zeros(6, np.float32), 2000),
 ("pm3d ellipse", dict(k=65536, ellipse=dict(a=1.5,b=.8,cx=.2,cy=-.1,speed=.7,m_state=2.,m_vel=.5), **pm), np.zeros(6, np.float32), 2000),
 ("mlp 2x256 bx3", dict(k=65536, mlp=mlp([9,256,256,6]), mlp_bf16x3=True, **pm), np.zeros(6, np.float32), 1500),
 ("mlp 2x256", dict(k=65536, mlp=mlp([9,256,256,6]), **pm), np.zeros(6, np.float32), 300),
 ("mlp32 bx3", dict(k=65536, mlp=mlp([9,32,32,32,6]), mlp_bf16x3=True, **pm), np.zeros(6, np.float32), 3000),
 ("mlp32 pc", dict(k=65536, mlp=mlp([9,32,32,32,6]), **pm), np.zeros(6, np.float32), 3000),       # r04: k_rollout_mlp32_pc (network wave + cost wave)
 ("mlp32 pc ragged", dict(k=200001, mlp=mlp([9,32,32,6]), **pm), np.zeros(6, np.float32), 500),
 ("mlp32 one wave", dict(k=65536, mlp=mlp([9,32,32,32,6]), tuning={"mlp32_valu": 2}, **pm), np.zeros(6, np.float32), 1000),
 ("nnauv32 bx3", dict(k=65536, nnauv=mlp([16,32,32,32,13]), mlp_bf16x3=True, **at), x13, 2000),
 ("nnauv pc", dict(k=65536, nnauv=mlp([16,32,32,32,13]), **at), x13, 2500),                      # r04: k_rollout_nnauv_pc (network wave + cost wave)
 ("nnauv pc ragged", dict(k=200001, nnauv=mlp([16,32,32,32,13]), **at), x13, 400),
 ("nnauv32", dict(k=65536, nnauv=mlp([16,32,32,32,13]), tuning={"mlp32_valu": 2}, **at), x13, 1000),
 ("nnspeed pc", dict(k=65536, nnauv_speed=mlp([15,16,16,16,6]), **at), x13, 3000),            # r04: k_rollout_nnspeed_pc (network wave + pose wave)
 ("nnspeed pc 32", dict(k=65536, nnauv_speed=mlp([15,32,32,6]), **at), x13, 1500),
 ("nnspeed pc ragged", dict(k=200001, nnauv_speed=mlp([15,16,16,16,6]), **at), x13, 500),        # several rounds, an odd tile count, a partial tile
 ("nnspeed mfma32", dict(k=65536, nnauv_speed=mlp([15,16,16,16,6]), tuning={"mlp32_valu": 2}, **at), x13, 1000),
 ("nnspeed valu", dict(k=65536, nnauv_speed=mlp([15,16,16,16,6]), tuning={"mlp32_valu": 1}, **at), x13, 500),
 ("auv pc", dict(k=65536, **auv_task(64)), x13, 4000),                                           # r04: k_rollout_auv_pc (pose wave + velocity wave)
 ("auv pc ragged", dict(k=200001, **auv_task(64)), x13, 600),
 ("auv pc rk4", dict(k=65536, **dict(auv_task(64), auv=dict(auv_task(64)["auv"], rk=4))), x13, 1000),
 ("auv one wave", dict(k=65536, tuning={"gen_one_wave": 1}, **auv_task(64)), x13, 1000),
 # r05: the one-launch step (k_step_pc: seven / five producer waves per tile, a long horizon) and the armed host-synchronous loop
 ("pm2d fused", dict(k=4096, **pm2), np.zeros(4, np.float32), 8000),
 ("pm2d fused 5 prod", dict(k=4096, tuning={"fused_step": 2}, **pm2), np.zeros(4, np.float32), 4000),
 ("pm3d fused K3000", dict(k=3000, **dict(pm, tau=50)), np.zeros(6, np.float32), 4000),
 ("pm3d fused H=120", dict(k=2048, **dict(pm, tau=120)), np.zeros(6, np.float32), 2000),
 ("pm3d armed sync", dict(k=65536, tuning={"armed_us": 500}, **pm), np.zeros(6, np.float32), 3000),
 ("pm2d armed sync", dict(k=4096, tuning={"armed_us": 500}, **pm2), np.zeros(4, np.float32), 4000),
 # r05, second session: the pre-launched pipelined step (two streams, granules for U'), drained and re-entered every 997 steps
 ("pm3d pre-launched", dict(k=65536, tuning={"prelaunch": 1}, **pm), np.zeros(6, np.float32), 6000),
 ("pm2d pre-launched", dict(k=4096, tuning={"prelaunch": 1}, **pm2), np.zeros(4, np.float32), 8000),
 ("pm3d pre K3000", dict(k=3000, tuning={"prelaunch": 1}, **dict(pm, tau=50)), np.zeros(6, np.float32), 4000),
 ("pm3d pre K16384", dict(k=16384, tuning={"prelaunch": 1}, **pm), np.zeros(6, np.float32), 4000),
]
MULT = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # python tools/soak.py 20: twenty times the steps of every case
for name, kw, x0, n in cases:
    n *= MULT
    kw = dict(kw); kw.pop("x0", None)
    h = m.Handle(**kw)
    x = torch.tensor(x0, device="cuda"); u = torch.zeros(kw["a_dim"], device="cuda")
    t0 = time.perf_counter()
    if "armed" in name:  # the host-synchronous loop: next(x) -> u -> plant, every call arming the next step's launch
        xh = np.asarray(x0, np.float32).copy()
        for i in range(n):
            uh = h.next(xh)
            xh[0::2] += 0.1 * xh[1::2]; xh[1::2] += 0.1 * uh  # (a crude plant: what matters is a fresh x per call)
            xh = np.clip(xh, -50, 50)
    else:
        for i in range(n):
            h.next_device(x.data_ptr(), u.data_ptr())
            if "pre" in name and i % 997 == 996: h.synchronize()
    h.synchronize()
    el = time.perf_counter() - t0
    U = h.get_action_sequence()
    print("%-16s %5d steps  %.3f ms/step  step counter %d  U finite %s  |U|max %.3g" % (name, n, 1e3 * el / n, h.get_step_counter(), bool(np.isfinite(U).all()), float(np.abs(U).max())), flush=True)
    assert h.get_step_counter() == n and np.isfinite(U).all()
    h.close()
print("soak ok")
