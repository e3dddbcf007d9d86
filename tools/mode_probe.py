#!/usr/bin/env python3
"""r05 study: the headline step has two modes under rocprofv3 (kernel 14.9 / 16.2 us; profiles/r05_kernel_duration_modes.txt). Is the mode a property
of the HANDLE (buffer addresses) or of the moment? Six identical handles alive at once, measured in turn, three rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mppi_tf_amd as m

def make():
    return m.Handle(k=65536, tau=64, s_dim=6, a_dim=3, dt=0.1, lam=1.0, sigma=0.25 * np.eye(3), goal=[1, 0, .5, 0, .75, 0])

hs = [make() for _ in range(6)]
x, u = torch.zeros(6, device="cuda"), torch.zeros(3, device="cuda")
for rnd in range(3):
    for i, h in enumerate(hs):
        for _ in range(300):
            h.next_device(x.data_ptr(), u.data_ptr())
        h.synchronize()
        ws = []
        for _ in range(10):
            t0 = time.perf_counter()
            for _ in range(400):
                h.next_device(x.data_ptr(), u.data_ptr())
            h.synchronize(); ws.append((time.perf_counter() - t0) / 400)
        h.profile_begin(400)
        for _ in range(400):
            h.next_device(x.data_ptr(), u.data_ptr())
        h.synchronize(); r, f, n = h.profile_end()
        print("round %d handle %d: step %.2f us (min %.2f max %.2f)  kernel %.2f  finish %.2f" % (rnd, i, np.median(ws) * 1e6, min(ws) * 1e6, max(ws) * 1e6, r * 1e3, f * 1e3), flush=True)
