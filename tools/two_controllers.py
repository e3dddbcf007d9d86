"""Do two INDEPENDENT controllers on one GPU fill each other's dispatch gaps?  N handles, each on its own stream, steps enqueued round-robin
from one host thread; aggregate control steps per second against one handle alone. configs[2] (K=65536) and the half-size pair (2 x 32768)."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mppi_tf_amd as m

def run(n, K, reps=20, steps=400):
    hs = [m.Handle(k=K, tau=64, s_dim=6, a_dim=3, dt=0.1, lam=1.0, sigma=0.25 * np.eye(3), goal=[1, 0, .5, 0, .75, 0], seed=1 + i) for i in range(n)]
    xs = [torch.zeros(6, device="cuda") for _ in hs]
    us = [torch.zeros(3, device="cuda") for _ in hs]
    def sync():
        torch.cuda.synchronize()
        for h in hs: h.synchronize()
    call = [(h.next_device, x.data_ptr(), u.data_ptr()) for h, x, u in zip(hs, xs, us)]
    for _ in range(300):
        for f, xp, up in call: f(xp, up, None)
    sync()
    ws = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(steps):
            for f, xp, up in call: f(xp, up, None)
        sync()
        ws.append((time.perf_counter() - t0) / steps)
    w = float(np.median(ws))
    print("%d controller(s) x K=%-6d  %.2f us per round of %d steps = %.2f us per control step, %.3g rollouts/s in all" % (n, K, w * 1e6, n, w * 1e6 / n, n * K / w), flush=True)
    for h in hs: h.close()

for _ in range(2):
    run(1, 65536); run(2, 65536); run(3, 65536); run(2, 32768); run(1, 32768); run(4, 16384); run(1, 4096); run(2, 4096); run(4, 4096)
