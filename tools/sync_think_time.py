"""Host-synchronous mppi_next with and without armed launches as a function of the host's think time between two calls (the plant): an armed
launch needs that time to get resident and draw its noise. Usage: python tools/sync_think_time.py [K H a]"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mppi_tf_amd as m
K, H, a = (int(v) for v in (sys.argv[1:4] + ["65536", "64", "3"][len(sys.argv) - 1:]))
cfg = dict(k=K, tau=H, s_dim=2 * a, a_dim=a, dt=0.1, lam=1.0, sigma=0.25 * np.eye(a), goal=([1, 0, .5, 0, .75, 0, .25, 0])[:2 * a])
x = np.zeros(2 * a, np.float32)
for think_us in (0, 2, 5, 10, 20, 50):
    row = []
    for tuning in ({}, {"armed_us": 500}):
        h = m.Handle(tuning=tuning, **cfg)
        ts = []
        for i in range(440):
            t0 = time.perf_counter()
            h.next(x)
            t1 = time.perf_counter()
            if i >= 40:
                ts.append(t1 - t0)
            while time.perf_counter() - t1 < think_us * 1e-6:
                pass
        h.close()
        ts = np.sort(ts) * 1e6
        row.append("%s median %.2f p95 %.2f min %.2f" % ("armed  " if tuning else "unarmed", np.median(ts), ts[int(.95 * len(ts))], ts[0]))
    print("K=%d think %3d us | %s | %s" % (K, think_us, row[0], row[1]), flush=True)
