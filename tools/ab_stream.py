"""Which stream a pipelined step is enqueued on: the handle's own non-blocking stream (stream=None), torch's default stream (handle 0 -> hipStreamLegacy),
a torch stream of its own. Wall per step and the kernels' own durations, configs[2]."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mppi_tf_amd as m
side = torch.cuda.Stream()
for name, st in (("own stream", None), ("torch default (legacy)", torch.cuda.default_stream()), ("torch side stream", side)) * 2:
    h = m.Handle(k=65536, tau=64, s_dim=6, a_dim=3, dt=0.1, lam=1.0, sigma=0.25*np.eye(3), goal=[1,0,.5,0,.75,0])
    x, u = torch.zeros(6, device="cuda"), torch.zeros(3, device="cuda")
    torch.cuda.synchronize()
    for _ in range(300): h.next_device(x.data_ptr(), u.data_ptr(), st)
    torch.cuda.synchronize(); h.synchronize()
    ws=[]
    for _ in range(20):
        t0=time.perf_counter()
        for _ in range(400): h.next_device(x.data_ptr(), u.data_ptr(), st)
        torch.cuda.synchronize(); h.synchronize(); ws.append((time.perf_counter()-t0)/400)
    h.profile_begin(400)
    for _ in range(400): h.next_device(x.data_ptr(), u.data_ptr(), st)
    torch.cuda.synchronize(); h.synchronize(); r,f,n = h.profile_end()
    print("%-26s step %.2f us  kernel %.2f us  finish %.2f us" % (name, np.median(ws)*1e6, r*1e3, f*1e3))
    h.close()
