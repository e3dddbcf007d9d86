"""Same-box A/B of the headline kernel: run once with the shipped library and once with MPPI_SO_PATH=build/variants/libmppi_hip_<variant>.so
(tools/ablate.py builds the variants). Prints the pipelined step and the kernels' own durations, three handles each."""
import sys, time, os; sys.path.insert(0, ".")
import numpy as np, torch, mppi_tf_amd as m
print(os.environ.get("MPPI_SO_PATH", "default"))
for rep in range(3):
    h = m.Handle(k=65536, tau=64, s_dim=6, a_dim=3, dt=0.1, lam=1.0, sigma=0.25*np.eye(3), goal=[1,0,.5,0,.75,0])
    x, u = torch.zeros(6, device="cuda"), torch.zeros(3, device="cuda")
    for _ in range(300): h.next_device(x.data_ptr(), u.data_ptr())
    h.synchronize()
    ws=[]
    for _ in range(20):
        t0=time.perf_counter()
        for _ in range(400): h.next_device(x.data_ptr(), u.data_ptr())
        h.synchronize(); ws.append((time.perf_counter()-t0)/400)
    h.profile_begin(400)
    for _ in range(400): h.next_device(x.data_ptr(), u.data_ptr())
    h.synchronize(); r,f,n = h.profile_end()
    print(h.rollout_kernel_name(), "step %.2f us  kernel %.2f us  finish %.2f us" % (np.median(ws)*1e6, r*1e3, f*1e3))
    h.close()
