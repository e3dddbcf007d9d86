"""Why bench.py discards its first event-bracketed pass: the kernels' own durations over successive passes of 200..1000 launches of one handle
(the first pass over fresh HIP events reads ~1.4 us high)."""
import sys, time, os; sys.path.insert(0, ".")
import numpy as np, torch, mppi_tf_amd as m
from mppi_tf_amd.distributed import ShardedController
dev = torch.device("cuda", 0)
cfg = dict(tau=64, s_dim=6, a_dim=3, dt=0.1, mass=1.0, lam=1.0, sigma=(0.25*np.eye(3)).astype(np.float32), goal=[1,0,.5,0,.75,0], seed=1)
x = torch.zeros(6, dtype=torch.float32, device=dev)
ctl = ShardedController(device_index=0, k=65536, exchange="auto", p2p_timeout_ms=1000, **cfg)
h = ctl.backend.h
def batch(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ctl.next(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for _ in range(3): batch(200)
b = [batch(200) for _ in range(100)]
print("ctl.next batches of 200: median %.2f us" % (np.median(b) * 1e6))
xp, up, sp = x.data_ptr(), ctl.u.data_ptr(), torch.cuda.current_stream(dev).cuda_stream
for n_prof in (200, 200, 400, 1000):
    h.profile_begin(n_prof)
    for _ in range(n_prof): h.next_device(xp, up, sp)
    torch.cuda.synchronize(dev)
    r, f, n = h.profile_end()
    print("profile %d launches via sp=%r: rollout %.2f us finish %.2f us" % (n_prof, sp, r * 1e3, f * 1e3))
for n_prof in (200, 400):
    for _ in range(2000): h.next_device(xp, up, sp)
    h.profile_begin(n_prof)
    for _ in range(n_prof): h.next_device(xp, up, sp)
    torch.cuda.synchronize(dev)
    r, f, n = h.profile_end()
    print("after 2000 unprofiled launches, profile %d: rollout %.2f us finish %.2f us" % (n_prof, r * 1e3, f * 1e3))
