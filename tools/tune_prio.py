"""The head starts of the four generations of k_rollout_pc's workgroups on a CU (pc_set_prio; MPPI_TUNE_PC_BALANCE = 0x10000 | b3 b2 b1 b0): kernel duration by
its own timestamps and the pipelined step for a list of settings, alternating on one box. Usage: python tools/tune_prio.py [K=65536 H=64 a=3] [hex ...]"""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mppi_tf_amd as m
shape = [int(v[2:]) for v in sys.argv[1:] if v.startswith("K=") or v.startswith("H=") or v.startswith("a=")]
K, H, a = (shape + [65536, 64, 3][len(shape):])
sets = [int(v, 16) for v in sys.argv[1:] if "=" not in v] or [0x0369, 0x036c, 0x034c, 0x033c, 0x033f, 0x0339, 0x0139, 0x0369, 0x0000, 0x0666, 0x0246, 0x0369]
x, u = torch.zeros(2 * a, device="cuda"), torch.zeros(a, device="cuda")
for bias in sets:
    h = m.Handle(k=K, tau=H, s_dim=2 * a, a_dim=a, dt=0.1, lam=1.0, sigma=0.25 * np.eye(a), goal=([1, 0, .5, 0, .75, 0, .25, 0])[:2 * a], tuning={"pc_balance": 0x10000 | bias})
    for _ in range(400): h.next_device(x.data_ptr(), u.data_ptr())
    h.synchronize()
    ws = []
    for _ in range(10):
        t0 = time.perf_counter()
        for _ in range(400): h.next_device(x.data_ptr(), u.data_ptr())
        h.synchronize(); ws.append((time.perf_counter() - t0) / 400)
    for _ in range(2):
        h.profile_begin(400)
        for _ in range(400): h.next_device(x.data_ptr(), u.data_ptr())
        h.synchronize(); r, f, n = h.profile_end()
    print("K=%d H=%d a=%d bias gen3..gen0 = %04x   step %.2f us   kernel %.2f us" % (K, H, a, bias, np.median(ws) * 1e6, r * 1e3), flush=True)
    h.close()
