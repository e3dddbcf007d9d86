"""The pre-launched pipelined step (MPPI_TUNE_PRELAUNCH) against the plain pipelined step at configs[2]: bit-identity of controls, sequence,
costs and step counter after the same number of steps, then wall per step (handle's own streams)."""
import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mppi_tf_amd as m

BAL = int(sys.argv[4], 0) if len(sys.argv) > 4 else None  # MPPI_TUNE_PC_BALANCE: 0 off, 0x10000 | head starts of the four generations


def mk(pre, K=65536, H=64, a=3):
    goal = [1, 0, .5, 0, .75, 0, .2, 0][:2 * a]
    h = m.Handle(k=K, tau=H, s_dim=2 * a, a_dim=a, dt=0.1, lam=1.0, sigma=0.25 * np.eye(a), goal=goal, seed=3)
    if pre: h.set_tuning("prelaunch", 1)
    if BAL is not None: h.set_tuning("pc_balance", BAL)
    return h

def run(h, n, x, u):
    for _ in range(n): h.next_device(x.data_ptr(), u.data_ptr(), None)
    torch.cuda.synchronize(); h.synchronize()

K = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
a = int(sys.argv[3]) if len(sys.argv) > 3 else 3
x = torch.tensor([0.1, 0, -0.2, 0, 0.3, 0, 0.05, 0][:2 * a], device="cuda")
res = []
for pre in (0, 1):
    h = mk(pre, K, H, a); u = torch.zeros(a, device="cuda")
    run(h, 7, x, u)
    res.append((u.cpu().numpy().copy(), h.get_action_sequence().copy(), h.debug_get(m.DBG_COSTS).copy(), h.get_step_counter()))
    run(h, 4, x, u)  # re-enter after the drain
    res[-1] += (u.cpu().numpy().copy(), h.get_action_sequence().copy())
    h.close()
names = ("u", "U", "costs", "step counter", "u after re-entry", "U after re-entry")
ok = True
for n_, p, q in zip(names, res[0], res[1]):
    same = np.array_equal(np.asarray(p), np.asarray(q))
    ok = ok and same
    print("%-18s %s" % (n_, "bit-identical" if same else "DIFFERENT  max|d| = %g" % np.abs(np.asarray(p, np.float64) - np.asarray(q, np.float64)).max()), flush=True)
print("parity:", "ok" if ok else "FAILED", flush=True)
for rep in range(int(os.environ.get('REPS', '3'))):
    for pre in (0, 1):
        h = mk(pre, K, H, a); u = torch.zeros(a, device="cuda")
        run(h, 300, x, u)
        ws = []
        for _ in range(20):
            t0 = time.perf_counter(); run(h, 400, x, u); ws.append((time.perf_counter() - t0) / 400)
        h.profile_begin(400); run(h, 400, x, u); r, f, n = h.profile_end()
        print("%-12s step %.2f us  (%.3g rollouts/s)  kernel %.2f us  finish %.2f us" % ("pre-launched" if pre else "plain", np.median(ws) * 1e6, K / np.median(ws), r * 1e3, f * 1e3), flush=True)
        h.close()
