#!/usr/bin/env python3
"""bench.py — rollouts/s of the MPPI control step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload pm3d|pm2d|pm1d|mlp|mlp32|auv|nnauv|nnspeed] [--horizon H] [--samples K_PER_GPU] [--bf16x3 | --fp-contract]

`--gpus N` with N > 1 works as typed: the parent process — before anything touches a GPU — starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` as a CHILD
and relays its one JSON line and exit code (a process that has initialised the GPU is never re-exec'd). Launched under
torch.distributed.run by someone else (WORLD_SIZE set) it is simply one rank.

A "step" is ONE control step of the hot path: K rollouts x H model steps + costs + soft-min update + shift, with the
state x and the nominal sequence U already resident in HBM and the noise drawn on the device (Philox).
Headline workload at N=1: BASELINE configs[2], point_mass3d analytic, K=65536, H=64 — the configuration the metric is
quoted on. Every BASELINE config is launchable:
    configs[1]  --workload pm2d --samples 4096                 (also a sub-record of the default run; its step is ONE launch, k_step_pc)
    configs[2]  (default)
    configs[3]  --workload mlp                                 (also a sub-record of the default run, with the split-bf16 variant)
    configs[4]  --workload mlp --horizon 128 --gpus 8          (K = 65536 per rank = 524288 in all; a sub-record of every N>1 run)
    (mlp32 / nnauv / auv: the reference's own Dense(32)x3 network, its NNAUVModel shape (s=13, a=6) and its Fossen AUVModel —
     not BASELINE configs; sub-records of the default run)
N>1 is WEAK scaling: every rank keeps --samples rollouts of a (samples x N)-sample controller, one exchange of the
(beta, eta, V) record per step. Rank 0 prints ONE JSON line (value = whole-job rollouts/s, max-over-ranks time).

Timing: W warm-up steps, then batches of EXACTLY K steps, each bracketed by barrier + synchronize on both sides and
reduced with MAX over ranks; batches repeat until --min-time seconds have been TIMED (default 1 s, whatever K is: a 200-step
batch of the analytic workload is 4 ms, one batch is a noisy sample), `value` is the MEDIAN batch; the kernel durations come
from >= 200 launches bracketed by the dispatches' own timestamps. `ms_per_control_step_sync` is the host-synchronous
mppi_next figure, with armed launches (MPPI_TUNE_ARMED_US) where the handle supports them and launch-per-call beside it.
N > 1 lines carry `parity` (sharded_parity below). The line stays below 4 KB: what each field means is written in DESIGN.md §6.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
T_START = time.time()

HID = 256                                 # BASELINE configs[3]: learned 2x256 MLP model_base
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3              # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32
MFMA_BF16_PEAK_TFLOPS = 2500.0
VALU_F32_PEAK_TFLOPS = 157.3              # 256 CUs x 4 SIMDs x 64 FLOP/clk (v_pk_fma_f32) x 2.4 GHz
GOALS = {1: [1.0, 0.0], 2: [1.0, 0.0, 0.0, 0.0], 3: [1.0, 0.0, 0.5, 0.0, 0.75, 0.0]}  # MuJoCo target sites
# workload -> (a_dim, learned model: None | (hidden width, hidden layers))
WORKLOADS = {"pm1d": (1, None), "pm2d": (2, None), "pm3d": (3, None), "mlp": (3, (256, 2)), "mlp32": (3, (32, 3)),
             "auv": (6, None), "nnauv": (6, (32, 3)), "nnspeed": (6, (16, 3))}  # nnspeed: NNAUVModelSpeed, Dense(16)x3 + Dense(6)
GEN = ("auv", "nnauv", "nnspeed")
CONFIG_NAME = {("pm1d", 128, 32, 1): "configs[0]", ("pm2d", 4096, 64, 1): "configs[1]", ("pm3d", 65536, 64, 1): "configs[2]",
               ("mlp", 65536, 64, 1): "configs[3]", ("mlp", 65536, 128, 8): "configs[4]"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="K steps per timed batch (default 200; 20 for the MLP workload, ~5 ms per step)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None,
                    help="default: pm3d (BASELINE configs[2], the metric's config) plus sub-records of the other configs")
    ap.add_argument("--horizon", type=int, default=None, help="H (default 64; 32 for pm1d)")
    ap.add_argument("--samples", type=int, default=None, help="rollouts PER GPU (default 65536; 4096 for pm2d, 128 for pm1d)")
    ap.add_argument("--min-time", type=float, default=1.0, help="repeat the K-step batch until this many seconds are timed (headline only)")
    ap.add_argument("--bf16x3", action="store_true", help="the split-bf16 matrix-core variant of a learned-model workload (MPPI_FLAG_MLP_BF16X3)")
    ap.add_argument("--fp-contract", action="store_true", help="the contracted instance of the point-mass rollout (MPPI_FLAG_FP_CONTRACT: fused multiply-adds; not bit-identical to the reference's rounding)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-subrecords", action="store_true")
    ap.add_argument("--no-prelaunched", action="store_true", help="skip the opt-in pre-launched pipeline's figure (under rocprofv3 --pmc dispatches are serialised: the pipeline cannot form, every step runs into its 20 ms deadline and the figure is skipped anyway)")
    return ap.parse_args()


def run_ranks(args, job_env, timeout_s, extra=()):
    """One child job: N ranks under torch.distributed.run with job_env (MPPI_EXCHANGE, MPPI_RCCL_CALL) set, at most timeout_s seconds.
    -> (parsed JSON line or None, outcome text). The child is its own process group: on a timeout exactly that group is
    killed (SIGTERM, then SIGKILL after 10 s). Nothing is ever re-exec'd and this process never touches a GPU."""
    import signal
    import tempfile
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL and the hipIpc inboxes need on this pool
    env.update(job_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + list(extra)
    sys.stderr.write("bench.py: %d ranks, %s, limit %d s\n" % (args.gpus, " ".join("%s=%s" % kv for kv in sorted(job_env.items())), timeout_s))
    sys.stderr.flush()
    with tempfile.TemporaryFile() as err:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=err, env=env, start_new_session=True)
        timed_out = False
        try:
            out, _ = p.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            timed_out = True
            for sig, wait in ((signal.SIGTERM, 10), (signal.SIGKILL, 10)):
                try:
                    os.killpg(p.pid, sig)  # the group this call started, nothing else
                except ProcessLookupError:
                    pass
                try:
                    out, _ = p.communicate(timeout=wait)
                    break
                except subprocess.TimeoutExpired:
                    out = b""
        err.seek(0)
        etxt = err.read().decode(errors="replace")
    sys.stderr.write(etxt)
    sys.stderr.flush()
    lines = [l for l in out.decode(errors="replace").splitlines() if l.startswith("{") and '"metric"' in l]
    tail = " | ".join(l.strip() for l in etxt.splitlines()[-6:] if l.strip())[-400:]
    if timed_out:
        return None, "failed: no line within %d s (ranks killed); %s" % (timeout_s, tail)
    if not lines:
        return None, "failed: rc=%d; %s" % (p.returncode, tail)
    try:
        return json.loads(lines[-1]), "ok"
    except ValueError:
        return None, "failed: unparsable line"


# The multi-GPU jobs of a parent run, in order (keep one job on the path that has run before). Job 1 is the all-gather
# through torch.distributed — three C calls + one collective per step, the path every earlier round measured (two ranks on one GPU over
# gloo, one rank over RCCL) and the only one with no native bring-up of its own: its line is the one that must come home. Job 2 tries
# the direct exchange (peer stores over xGMI inside the finish kernel); job 3 the one-call path (mppi_shard_step -> ncclAllGather on the
# controller's own communicator) — neither has run between two real devices yet, and a hang in either costs a field, not the line.
JOBS = [("rccl_torch", {"MPPI_EXCHANGE": "rccl", "MPPI_RCCL_CALL": "torch"}),
        ("p2p", {"MPPI_EXCHANGE": "auto", "MPPI_RCCL_CALL": "c"}),
        ("rccl_c", {"MPPI_EXCHANGE": "rccl", "MPPI_RCCL_CALL": "c"})]


def merge_job_lines(got, outcome, order):
    """got: job name -> parsed line (only the jobs that produced one); outcome: job name -> text for the others. -> the line to print:
    the best headline, every job's result under `exchange`, sub-records and the parity verdict carried over from the job that has them."""
    best = max(got, key=lambda j: got[j]["value"])
    line = dict(got[best])
    ex = dict(line.get("exchange") or {})
    for j in order:
        if j in got:
            g = got[j]
            gex = g.get("exchange") or {}
            ex[j] = {"value": r4(g["value"]), "ms_per_step": r4(g["ms_per_step"]), "used": gex.get("used", g["config"].get("exchange")),
                     "call": gex.get("rccl_call")}
            if "parity" in g:
                ex[j]["parity"] = g["parity"]
            if j == "p2p" and ex[j]["used"] != "p2p":
                ex[j]["note"] = "direct exchange not used: %s" % gex.get("direct_exchange_bring_up")
        elif j in outcome:
            ex[j] = outcome[j]
    ex["printed"] = best
    line["exchange"] = ex
    for key in ("sub_records", "parity"):
        if key not in line:
            for j in order:
                if j in got and key in got[j]:
                    line[key] = got[j][key]
                    if key == "parity":
                        line["parity"] = dict(got[j][key], measured_by=j)
                    break
    return line


def self_launch(args):
    """`python bench.py --gpus N` typed directly (what the driver's scaling run does): run the N ranks as CHILD jobs (JOBS above) and relay
    one line. Nothing in this process has touched a GPU (torch is not even imported yet). Each job has a time limit well inside the
    driver's 600 s (MPPI_BENCH_BUDGET_S, default 540 s in all); a job's line is written to stderr the moment it exists."""
    t_start = time.time()
    budget = float(os.environ.get("MPPI_BENCH_BUDGET_S", "540"))
    if os.environ.get("MPPI_BENCH_ONE_GPU") == "1":
        # the one-GPU rehearsal: RCCL refuses two ranks on one device, so the "rccl" job exchanges its records with gloo's all-gather
        # (the torch path, staged through the host) — it walks this function's logic, not RCCL
        jobs = [("rccl_torch", {"MPPI_EXCHANGE": "rccl", "MPPI_RCCL_CALL": "torch"}), ("p2p", {"MPPI_EXCHANGE": "p2p"})]
    elif os.environ.get("MPPI_EXCHANGE"):
        jobs = [(os.environ["MPPI_EXCHANGE"], {"MPPI_EXCHANGE": os.environ["MPPI_EXCHANGE"]})]
    else:
        jobs = JOBS
    got, outcome = {}, {}
    for i, (name, env) in enumerate(jobs):
        left = budget - (time.time() - t_start)
        last = i == len(jobs) - 1
        limit = left - 15 if last else min(240.0, 0.55 * left)
        if limit < 45:
            outcome[name] = "skipped: %d s left of the %d s budget" % (left, budget)
            continue
        # later jobs re-measure the headline only: the configs[4] sub-record (8 ms steps) does not depend on the exchange
        extra = ["--no-subrecords"] if got else []
        line, outcome[name] = run_ranks(args, env, int(limit), extra)
        if line is not None:
            got[name] = line
            sys.stderr.write("bench.py: job %s: %s\n" % (name, json.dumps(line)))
            sys.stderr.flush()
    if not got:
        sys.stderr.write("bench.py: no job produced a line: %s\n" % json.dumps(outcome))
        sys.exit(1)
    line = merge_job_lines(got, outcome, [j for j, _ in jobs]) if len(jobs) > 1 else next(iter(got.values()))
    sys.stdout.write(json.dumps(line) + "\n")
    sys.stdout.flush()
    sys.exit(0)


def direct_exchange_in_children(args, world, rank, dist):
    """Rank mode (the driver's own `torch.distributed.run ... bench.py --gpus N`): this process IS a rank, there is no parent of ours to
    run two jobs. The ranks have just measured the all-gather path (MPPI_EXCHANGE=rccl) in-process; the direct exchange — peer stores
    over xGMI inside the finish kernel, never run between two real devices so far — is now tried in CHILD processes, one per rank (same
    RANK / LOCAL_RANK / WORLD_SIZE, a rendezvous port of their own, MPPI_EXCHANGE=auto, headline only), with a time limit: a fault, a
    hang or a bad exit there costs a field of the line, not the line. The ranks wait for their child (the GPU is theirs
    meanwhile), the group is killed on the limit. -> (rank 0: the child job's parsed line or None, outcome text); other ranks (None, "")"""
    import signal
    import tempfile
    left = float(os.environ.get("MPPI_BENCH_BUDGET_S", "540")) - (time.time() - T_START)
    limit = min(float(os.environ.get("MPPI_BENCH_P2P_LIMIT_S", "150")), left - 30)
    box = [None]
    if rank == 0:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            box[0] = (so.getsockname()[1], limit)
    dist.broadcast_object_list(box, src=0)
    port, limit = box[0]  # every rank takes rank 0's decision
    if limit < 45:
        return None, "skipped: %d s left of the budget" % max(left, 0)
    env = dict(os.environ, MPPI_EXCHANGE="auto", MPPI_BENCH_CHILD="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
        del env[k]  # (TORCHELASTIC_USE_AGENT_STORE would make the children look for the launcher's store on the new port)
    drop = {"--no-subrecords", "--no-cpu-baseline"}
    cmd = [sys.executable, os.path.abspath(__file__)] + [a_ for a_ in sys.argv[1:] if a_ not in drop] + ["--no-subrecords", "--no-cpu-baseline"]
    if rank == 0:
        sys.stderr.write("bench.py: the direct exchange in %d child processes, limit %d s\n" % (world, limit))
        sys.stderr.flush()
    with tempfile.TemporaryFile() as err:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=err, env=env, start_new_session=True)
        out, timed_out = b"", False
        try:
            out, _ = p.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            timed_out = True
            for sig, wait in ((signal.SIGTERM, 10), (signal.SIGKILL, 10)):
                try:
                    os.killpg(p.pid, sig)  # the group this call started, nothing else
                except ProcessLookupError:
                    pass
                try:
                    out, _ = p.communicate(timeout=wait)
                    break
                except subprocess.TimeoutExpired:
                    out = b""
        err.seek(0)
        etxt = err.read().decode(errors="replace")
    if rank != 0:
        return None, ""
    sys.stderr.write(etxt)
    lines = [l for l in (out or b"").decode(errors="replace").splitlines() if l.startswith("{") and '"metric"' in l]
    tail = " | ".join(l.strip() for l in etxt.splitlines()[-4:] if l.strip())[-300:]
    if timed_out:
        return None, "failed: no line within %d s (children killed); %s" % (limit, tail)
    if not lines:
        return None, "failed: rc=%d; %s" % (p.returncode, tail)
    try:
        return json.loads(lines[-1]), "ok"
    except ValueError:
        return None, "failed: unparsable line"


def cfg_of(workload, H):
    import numpy as np
    a = WORKLOADS[workload][0]
    if workload in GEN:  # the reference's AUV task: 13-state quaternion pose, 6 thrusts (config/tasks/static_cost_auv.yaml)
        from mppi_tf_amd.auv import auv_task
        return auv_task(H, learned=(workload != "auv"))
    return dict(tau=H, s_dim=2 * a, a_dim=a, dt=0.1, mass=1.0, lam=1.0, sigma=(0.25 * np.eye(a)).astype(np.float32),
                goal=GOALS[a], seed=1)


def synthetic_mlp(n_in, n_out, seed=0, hid=HID, n_hidden=2):
    """in->hid x n_hidden->out ReLU, U(-1/sqrt(fan_in), 1/sqrt(fan_in)), last layer x0.1, identity normalisation;
    hid=32, n_hidden=3 is the reference's own network shape (nn_model.py:54-60)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    dims = [n_in] + [hid] * n_hidden + [n_out]
    W = [(rng.uniform(-1, 1, (dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(n_hidden + 1)]
    b = [(rng.uniform(-1, 1, dims[i + 1]) / np.sqrt(dims[i])).astype(np.float32) for i in range(n_hidden + 1)]
    W[-1] *= 0.1
    b[-1] *= 0.1
    return dict(W=W, b=b)


_trained = {}


def trained_nnauv():
    """NNAUVModel's Dense(32)x3 network TRAINED here, on the device: 8192 transitions of the rexrov2 Fossen AUVModel (rolled by the
    device model), LearnerBase's normalisation statistics, 300 full-batch Adam steps (mppi_learner_*). The one learned-model number
    that is not on synthetic weights. -> (mlp dict for Handle(nnauv=...), description)"""
    if "nnauv" not in _trained:
        import numpy as np
        import mppi_tf_amd as m
        from mppi_tf_amd.auv import auv_task
        plant = m.AUVModel(actionDim=6, dt=0.1, parameters=auv_task(8)["auv"])
        rng = np.random.default_rng(0)
        n = 8192
        x = rng.standard_normal((n, 13)) * np.array([1, 1, 1, 0, 0, 0, 0, .5, .5, .5, .2, .2, .2])
        q = rng.standard_normal((n, 4)) * 0.3 + np.array([0, 0, 0, 1.0])
        x[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
        u = 1500.0 * rng.standard_normal((n, 6))
        xn = plant.build_step_graph("plant", x[..., None], u[..., None])
        model = m.NNAUVModel()
        learner = m.LearnerBase(model, bufferSize=n)
        learner.add_rb(x[..., None], u[..., None], xn)
        learner.stats()
        first, last = learner.train_all(learningRate=3e-3, epoch=300)
        _trained["nnauv"] = (model.mlp(), "trained on device: %d Fossen transitions, 300 Adam steps, MSE %.3g -> %.3g" % (n, first, last))
    return _trained["nnauv"]


def model_kw_of(workload, mlp):
    return {"nnauv": dict(nnauv=mlp), "nnspeed": dict(nnauv_speed=mlp)}.get(workload, dict(mlp=mlp))


def mlp_of(workload):
    a, net = WORKLOADS[workload]
    if net is None:
        return None
    if workload == "nnauv":  # NNAUVModel: input = state without the position + action (nn_model.py:289-293), output = 13
        return trained_nnauv()[0]
    if workload == "nnspeed":  # NNAUVModelSpeed: 15 inputs (Euler angles, velocities, forces), 6 outputs; synthetic weights (SURVEY §8d recipe)
        return synthetic_mlp(15, 6, 0, *net)
    return synthetic_mlp(3 * a, 2 * a, 0, *net)


def work_per_state_step(workload):
    """algorithmic work of one (k, t) pair: (bytes, flop)"""
    a, net = WORKLOADS[workload]
    s = 13 if workload in GEN else 2 * a
    flop = 6 * s + 5 * a + 3
    # (the 13-state family's lane-per-rollout kernels are priced from their COUNTED instruction stream, roofline_of: no FLOP estimate)
    if net:
        hid, n_hidden = net
        n_in, n_out = {"nnauv": (16, s), "nnspeed": (15, 6)}.get(workload, (s + a, s))
        flop += 2 * (n_in * hid + (n_hidden - 1) * hid * hid + hid * n_out) + (120 if workload == "nnspeed" else 0)  # + Euler angles, quaternion kinematics
    return 12 * a, flop  # bytes: the noise written once and read twice, fp32


def cpu_baseline(workload, H, K, mlp, budget_s=12.0):
    """The CPU restatement (oracle/, OpenMP over samples) timed on this box's host cores on a bounded sample of the SAME
    workload: whole control steps (noise + rollouts + update) at the same H — the same K for the analytic model, K=4096
    for the MLP (stated in `sample`)."""
    import numpy as np
    from oracle import oracle as orc
    a = WORKLOADS[workload][0]
    Kc = K if mlp is None and workload not in GEN else min(K, 4096)
    c = cfg_of(workload, H)
    if workload in GEN:  # the 13-state family: the same task through the CPU restatement's model_base / cost_base slots
        mkw = {"auv": dict(auv=c.get("auv")), "nnauv": dict(nnauv=mlp), "nnspeed": dict(nnauv_speed=mlp)}[workload]
        p = orc.Problem(tau=H, s=13, a=a, dt=c["dt"], lam=c["lam"], sigma=c["sigma"], goal=c["goal"], Q=c["Q"], threads=0, **mkw)
        x, U = np.asarray(c["x0"], np.float32), np.zeros((H, a), np.float32)
    else:
        p = orc.Problem(tau=H, s=2 * a, a=a, dt=0.1, mass=1.0, lam=1.0, sigma=c["sigma"], goal=c["goal"], threads=0, mlp=mlp)
        x, U = np.zeros(2 * a, np.float32), np.zeros((H, a), np.float32)
    eps = orc.noise(1, 0, 0, Kc, H, a, c["sigma"])
    p.next_with_noise(x, U, eps)  # warm-up (page in, spin up the OpenMP team)
    n, t0 = 0, time.perf_counter()
    while True:
        eps = orc.noise(1, n + 1, 0, Kc, H, a, c["sigma"])
        _, U, _ = p.next_with_noise(x, U, eps)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    return {"value": float("%.6g" % (Kc * n / el)), "unit": "rollouts/s", "cores": orc.num_threads(), "kind": "port", "ms_per_step": r4(1e3 * el / n),
            "sample": "%d whole control steps (noise + rollouts + update) of the same workload, K=%d H=%d, OpenMP over samples" % (n, Kc, H)}


ARMED_US = 500  # soft deadline of the armed launches in the synchronous loop below (MPPI_TUNE_ARMED_US)


def sync_record(m, workload, H, K, mlp, **handle_kw):
    """The host-synchronous figure of a workload: mppi_next with armed launches where the handle supports them (the point-mass
    producer/consumer path on a large-BAR system) beside the launch-per-call figure. ms."""
    med, p95 = sync_latency(m, workload, H, K, mlp, **handle_kw)
    rec = {"median": r4(med), "p95": r4(p95), "mode": "launch per call"}
    if mlp is None and workload not in GEN:
        try:
            am, ap = sync_latency(m, workload, H, K, mlp, tuning={"armed_us": ARMED_US}, **handle_kw)
            rec = {"median": r4(am), "p95": r4(ap), "mode": "armed (MPPI_TUNE_ARMED_US=%d)" % ARMED_US, "launch_per_call": {"median": r4(med), "p95": r4(p95)}}
            # the loop above spends ~3 us between two calls (a numpy point mass): an armed launch is then still drawing its noise when x arrives.
            # With a plant of 10 us (any simulator's step) it is ready: the same figure, measured (tools/sync_think_time.py has the curve)
            tm, _ = sync_latency(m, workload, H, K, mlp, tuning={"armed_us": ARMED_US}, think_us=10.0, **handle_kw)
            rec["plant_10us"] = r4(tm)
        except Exception as e:  # no large BAR: MPPI_ERR_UNSUPPORTED
            rec["armed"] = "unavailable: %s" % str(e)[:80]
    return rec


def sync_latency(m, workload, H, K, mlp, steps=200, warmup=20, think_us=0.0, **handle_kw):
    """Host-synchronous closed loop: mppi_next(x)->u with the plant stepped on the host (the shape of the reference's
    loop, main.cpp:37-43). Median / p95 ms per control step."""
    import numpy as np
    a = WORKLOADS[workload][0]
    cfg = cfg_of(workload, H)
    gen = workload in GEN  # 13-state family: the state is held at the task's x0 (no host plant for the Fossen model here)
    x = np.asarray(cfg.pop("x0"), np.float32) if gen else np.zeros(2 * a, np.float32)
    h = m.Handle(k=K, **model_kw_of(workload, mlp), **cfg, **handle_kw)
    if mlp is not None or gen:
        steps, warmup = 20, 3
    dt, ts = 0.1, []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        u = h.next(x)
        t1 = time.perf_counter()
        if i >= warmup:
            ts.append(t1 - t0)
        for j in range(0 if gen else a):  # point-mass plant, fp32, same model
            x[2 * j] = x[2 * j] + dt * x[2 * j + 1] + (dt * dt / 2) * u[j]
            x[2 * j + 1] = x[2 * j + 1] + dt * u[j]
        while think_us and time.perf_counter() - t1 < think_us * 1e-6:  # a plant that takes think_us (the reference's is a MuJoCo step)
            pass
    ts = np.sort(np.asarray(ts)) * 1e3
    h.close()
    return float(np.median(ts)), float(ts[int(0.95 * (len(ts) - 1))])


def prelaunched_record(m, workload, H, K, budget_s=0.25):
    """The PRE-LAUNCHED pipelined step (MPPI_TUNE_PRELAUNCH, opt-in; mppi_capi.hip pre_step): steps alternate between the handle's two
    own streams, step n+1's rollout is resident and draws its noise while step n finishes. Reported BESIDE the headline, never as it:
    the state of step n+1 has to be in place before step n's control exists (a throughput mode — x does not wait for u), and the
    steps do not ride the caller's stream. Same controls bit for bit (tests/test_step_gpu.py). -> {"ms_per_step", "value"} or a note."""
    import numpy as np
    import torch
    a = WORKLOADS[workload][0]
    try:
        h = m.Handle(k=K, tuning={"prelaunch": 1}, **cfg_of(workload, H))
    except Exception as e:
        return {"unavailable": str(e)[:80]}
    try:
        x, u = torch.zeros(2 * a, dtype=torch.float32, device="cuda"), torch.zeros(a, dtype=torch.float32, device="cuda")
        xp, up = x.data_ptr(), u.data_ptr()

        def batch(n):
            t0 = time.perf_counter()
            for _ in range(n):
                h.next_device(xp, up, None)
            h.synchronize()
            return (time.perf_counter() - t0) / n
        batch(20)  # (a pipeline that cannot form — every wait is bounded at 20 ms — shows here, cheaply: the error surfaces at the drain)
        batch(300)
        ws, t0 = [], time.perf_counter()
        while time.perf_counter() - t0 < budget_s or len(ws) < 5:
            ws.append(batch(400))
        assert np.isfinite(u.cpu().numpy()).all()
        el = float(np.median(ws))
        return {"ms_per_step": r4(1e3 * el), "value": r4(K / el)}
    finally:
        h.close()


_kernel_profiles = None


def measured(kernel):
    """What rocprofv3's counters say about `kernel` (profiles/kernels_latest.json, written by tools/summarize_profiles.py from
    separate --pmc passes of this bench): {"valu": instructions per launch by class x issue cycles, "mfma": matrix-pipe busy
    fraction, "traffic": HBM bytes per launch, "stats": rocprofv3 --kernel-trace --stats average}, or (None, why) when the file
    is absent, holds nothing for this kernel, or was taken on other kernel sources than the ones this run executes."""
    global _kernel_profiles
    import mppi_tf_amd as m
    if _kernel_profiles is None:
        try:
            _kernel_profiles = json.load(open(os.path.join(ROOT, "profiles", "kernels_latest.json")))
        except Exception:
            _kernel_profiles = {}
    d = _kernel_profiles.get("kernels", {}).get(kernel)
    if d is None:
        return None, "profiles/kernels_latest.json holds nothing for %s" % kernel
    sha = m.build.source_sha()
    if d.get("code_sha") != sha:
        return None, "profiles/kernels_latest.json has %s of sources %s, this run executes %s" % (kernel, d.get("code_sha"), sha)
    return d, None


def valu_floor(v):
    """(floor_us, SIMD-cycles the launch's vector instructions need, peak G SIMD-cycle/s) from a `valu` profile record"""
    cyc = sum(v["insts_per_launch"][c] * v["cycles_per_inst"][c] for c in v["insts_per_launch"])
    return cyc / (v["simds"] * v["clock_mhz"]), cyc, v["simds"] * v["clock_mhz"] * 1e-3


def sharded_parity(rn, workload, K, H, steps=3, normalize=False, unsharded_ref=True):
    """OUTSIDE every timed region (the first multi-GPU line must carry a correctness verdict, not just a time).
    A fresh K x world controller takes `steps` steps from (x0, U = 0, step 0); then
      ranks_bit_identical            every rank's controls and final U, gathered and compared bit for bit (the finish is replicated: SURVEY §8e);
      sharded_vs_unsharded_max_abs   rank 0 runs the SAME steps on ONE unsharded handle of K x world samples (same seed, global Philox
                                     counters: the same noise) and reports max |difference| over the controls and the final U (bar 2e-6).
    A collective: every rank calls it with the same arguments."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import mppi_tf_amd as m
    from mppi_tf_amd.distributed import ShardedController
    mlp = mlp_of(workload)
    cfg = cfg_of(workload, H)
    x = torch.tensor(cfg.pop("x0"), dtype=torch.float32, device=rn.dev) if "x0" in cfg else torch.zeros(cfg["s_dim"], dtype=torch.float32, device=rn.dev)
    kw = dict(model_kw_of(workload, mlp), **cfg)
    if normalize:
        kw["normalize_cost"] = True
    ex = os.environ.get("MPPI_EXCHANGE", "auto")
    ctl = ShardedController(device_index=rn.local_rank, k=K * rn.world, exchange="rccl" if (normalize and ex == "p2p") else ex, p2p_timeout_ms=1000, **kw)
    us = [ctl.next(x).clone() for _ in range(steps)]
    torch.cuda.synchronize(rn.dev)
    mine = np.concatenate([torch.stack(us).flatten().cpu().numpy(), ctl.backend.action_sequence().numpy().ravel()]).astype(np.float32)
    rows = [None] * rn.world
    if dist.is_initialized():
        dist.all_gather_object(rows, mine.tobytes())
    else:
        rows = [mine.tobytes()]
    out = {"steps": steps, "exchange": ctl.exchange, "ranks_bit_identical": all(r_ == rows[0] for r_ in rows), "finite": bool(np.isfinite(mine).all())}
    del ctl
    if unsharded_ref and rn.rank == 0:
        h = m.Handle(k=K * rn.world, device=rn.local_rank, **kw)
        xh = x.cpu().numpy()
        ref = [h.next(xh).copy() for _ in range(steps)]
        ref = np.concatenate([np.asarray(ref).ravel(), h.get_action_sequence().ravel()]).astype(np.float32)
        h.close()
        out["sharded_vs_unsharded_max_abs"] = float(np.abs(ref - mine).max())
    return out


class Runner:
    def __init__(self, args, dev, world, rank, local_rank):
        self.args, self.dev, self.world, self.rank, self.local_rank = args, dev, world, rank, local_rank

    def barrier(self):
        import torch
        import torch.distributed as dist
        torch.cuda.synchronize(self.dev)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(self.dev)

    def cdev(self):
        """where the bookkeeping collectives run: the GPU under RCCL, the host in the one-GPU gloo rehearsal"""
        import torch.distributed as dist
        return self.dev if (not dist.is_initialized() or dist.get_backend() == "nccl") else "cpu"

    def max_over_ranks(self, v):
        import torch
        import torch.distributed as dist
        t = torch.tensor([v], dtype=torch.float64, device=self.cdev())
        if dist.is_initialized():
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def per_rank(self, v):
        import torch
        import torch.distributed as dist
        t = torch.tensor([v], dtype=torch.float64, device=self.cdev())
        if not dist.is_initialized():
            return [float(v)]
        out = torch.zeros(self.world, dtype=torch.float64, device=self.cdev())
        dist.all_gather_into_tensor(out, t)
        return [float(q) for q in out.cpu()]

    def run(self, workload, K, H, steps, warmup, min_time, **handle_kw):
        """-> dict of measurements of one workload (K rollouts per rank)"""
        import numpy as np
        import torch
        from mppi_tf_amd.distributed import ExchangeTimeout, ShardedController
        a, net = WORKLOADS[workload]
        mlp = mlp_of(workload)
        cfg = cfg_of(workload, H)
        x = torch.zeros(cfg["s_dim"], dtype=torch.float32, device=self.dev)
        if "x0" in cfg:
            x = torch.tensor(cfg.pop("x0"), dtype=torch.float32, device=self.dev)
        model_kw = model_kw_of(workload, mlp)
        ctl = ShardedController(device_index=self.local_rank, k=K * self.world, exchange=os.environ.get("MPPI_EXCHANGE", "auto"),
                                p2p_timeout_ms=1000, **model_kw, **cfg, **handle_kw)
        assert ctl.backend.h.k_local == K
        rank_el = [0.0]

        def steps_timed(n):
            """n steps between two barriers. Every rank runs the SAME collective sequence whatever happens to its exchange:
            barrier, then the agreement on `bad`, then (if bad anywhere) resync on every rank and once more."""
            for attempt in range(2):
                self.barrier()
                t0 = time.perf_counter()
                bad = 0
                try:
                    for _ in range(n):
                        ctl.next(x)
                except ExchangeTimeout:
                    bad = 1
                torch.cuda.synchronize(self.dev)
                rank_el[0] = time.perf_counter() - t0
                self.barrier()
                el = time.perf_counter() - t0
                if not bad:
                    try:
                        ctl.check()
                    except ExchangeTimeout:
                        bad = 1
                if self.max_over_ranks(bad) == 0:
                    return self.max_over_ranks(el)
                ctl.resync()
            raise RuntimeError("the all-gather path cannot time out")

        steps_timed(warmup)
        # batches of EXACTLY `steps` steps until --min-time seconds have been TIMED (a cap of 200 batches made the
        # driver's --steps 20 run 79 ms of GPU work, over before the clocks had settled). The cap is wall time (barriers and Python
        # frames between batches included), rank 0's decision broadcast so every rank runs the same number of collectives.
        batches, t_wall = [steps_timed(steps)], time.perf_counter()
        while True:
            more = sum(batches) < min_time and time.perf_counter() - t_wall < 4.0 * min_time + 1.0 and len(batches) < 100000
            if self.max_over_ranks(1.0 if more else 0.0) == 0.0:
                break
            batches.append(steps_timed(steps))
        el = float(np.median(batches))
        rank_ms = [1e3 * q / steps for q in self.per_rank(rank_el[0])]  # each rank's own time for the last batch, before the barrier
        # kernel durations: the same K steps with HIP events bound to each launch (the dispatches' own begin/end)
        h = ctl.backend.h
        # >= 200 launches whatever --steps is (the driver's --steps 20 gave a 20-launch average, 10 % above steady state);
        # the millisecond-scale learned-model steps keep their batch size (their kernels are long enough to be their own steady state)
        n_prof = max(steps, 200) if el / steps < 1e-3 else steps
        # The pass runs TWICE and the first is discarded: a HIP event is backed by a signal the runtime allocates when the event is first
        # recorded, and a pass over fresh events starves the queue while it does — the first 200 event-bracketed launches of a handle read
        # 16.4 us for the headline kernel, the second 200 15.6, every later pass 15.0 (tools/dbg_prof.py; r04's driver line: 17.32 against
        # 15.79 under rocprofv3). Steady state is what the roofline is about.
        for attempt in range(2 if el / steps < 1e-3 else 1):
            h.profile_begin(n_prof)
            if self.world == 1 and ctl.exchange == "none":
                # the timed region keeps the GPU's queue full; so must this pass, or the kernels are timed on a GPU that idles between them
                # (the Python frames of ShardedController.next cost more than an event-bracketed launch leaves): the handle directly
                xp, up, sp = x.data_ptr(), ctl.u.data_ptr(), torch.cuda.current_stream(self.dev)
                for _ in range(n_prof):
                    h.next_device(xp, up, sp)
            else:
                for _ in range(n_prof):
                    ctl.next(x)
            torch.cuda.synchronize(self.dev)
            roll_ms, fin_ms, n_done = h.profile_end()
        n_prof = n_done
        assert np.isfinite(ctl.u.cpu().numpy()).all(), "non-finite control after %d profiled steps of %s (exchange %s): u = %s, first record (beta, eta) = %s" % (
            n_prof, workload, ctl.exchange, ctl.u.cpu().numpy(), ctl.records[:2].cpu().numpy())
        bytes_ss, flop_ss = work_per_state_step(workload)
        state_steps = K * H
        res = {"workload": workload, "K_per_gpu": K, "H": H, "a_dim": a, "s_dim": cfg["s_dim"], "steps": steps, "batches_s": batches,
               "rollouts_per_s": K * self.world * steps / el, "ms_per_step": 1e3 * el / steps, "rank_ms_per_step": rank_ms,
               "kernel": h.rollout_kernel_name(), "kernel_ms_avg": roll_ms, "finish_kernel_ms_avg": fin_ms, "launches_timed": n_prof,
               "algorithmic_bytes_per_launch": bytes_ss * state_steps + 8 * K, "algorithmic_flop_per_launch": flop_ss * state_steps,
               "exchange": ctl.exchange, "p2p_note": ctl.p2p_note, "rccl_note": ctl.rccl_note, "record_size": h.record_size, "mlp": mlp}
        del ctl
        return res


def r4(v):
    return None if v is None else float("%.4g" % v)


def roofline_of(r):
    """The bound each kernel is actually on: `mfma` for the learned 2x256 / Dense(32) models (exact-fp32
    matrix cores; the split-bf16 variants against the bf16 peak), `valu_issue` for the lane-per-rollout kernels (the analytic
    point mass — its noise never leaves the chip, so HBM is idle (`traffic`) — the Fossen AUVModel, the small networks on the
    vector ALU; NNAUVModelSpeed's default kernel runs its layers on the matrix cores and is priced as `mfma`). `frac` of a valu_issue kernel = floor / kernel time, floor = the launch's COUNTED vector
    instructions by class (rocprofv3 SQ_INSTS_VALU_*) x the issue cycles per instruction of the class measured on this part
    (profiles/r02_valu_issue.json) / (1024 SIMDs x 2.4 GHz) — no hand-estimated FLOP count.
    `algorithmic_hbm_frac` keeps SURVEY §8d's materialised-noise figure for the BASELINE point-mass configurations."""
    kus = 1e3 * r["kernel_ms_avg"]
    base = {"kernel": r["kernel"].replace("mppi::", ""), "kernel_us": r4(kus), "finish_kernel_us": r4(1e3 * r["finish_kernel_ms_avg"]),
            "launches_timed": r["launches_timed"]}
    prof, why = measured(r["kernel"])
    prof = prof or {}
    traffic = (prof.get("traffic") or {}).get("hbm_bytes_per_launch")
    traffic = int(traffic) if traffic is not None else None
    valu_bound = r["mlp"] is None or "k_rollout_gen" in r["kernel"] or "mlp_small" in r["kernel"]
    if not valu_bound:
        flop = r["algorithmic_flop_per_launch"]
        bx3 = "bx3" in r["kernel"]
        tf = (3 if bx3 else 1) * flop / (kus * 1e-6) / 1e12 if kus > 0 else 0.0
        peak = MFMA_BF16_PEAK_TFLOPS if bx3 else MFMA_F32_PEAK_TFLOPS
        mf = prof.get("mfma") or {}
        base.update({"bound": "mfma", "achieved": r4(tf), "peak": peak, "unit": "TFLOP/s", "frac": r4(tf / peak), "traffic": traffic,
                     "mfma_busy_frac": r4(mf.get("mfma_busy_frac")), "vector_insts_per_mfma": r4(mf.get("other_vector_insts_per_mfma")),
                     "algorithmic_flop_per_launch": flop, "profiles": prof.get("tag")})
        if bx3:
            base["algorithmic_TFLOP_per_s"] = r4(tf / 3)
        return base
    base.update({"bound": "valu_issue", "unit": "G SIMD-cycle/s", "traffic": traffic})
    if r["workload"] in ("pm1d", "pm2d", "pm3d"):  # the BASELINE point-mass configurations: SURVEY §8d's byte model beside it (keyed on the workload — ADVICE r04)
        alg = r["algorithmic_bytes_per_launch"]
        base.update({"algorithmic_bytes_per_launch": alg, "algorithmic_hbm_frac": r4(alg / (kus * 1e-6) / 1e9 / HBM_PEAK_GBS) if kus > 0 else None})
    v = prof.get("valu")
    if v is None:
        base.update({"achieved": None, "peak": None, "frac": None, "pmc_note": why or "no VALU-class counters for this kernel"})
        return base
    floor_us, cyc, peak = valu_floor(v)
    busy = v.get("active_quad_cycles_per_launch")
    base.update({"achieved": r4(cyc / (kus * 1e-6) / 1e9) if kus > 0 else None, "peak": r4(peak), "frac": r4(floor_us / kus) if kus > 0 else None,
                 "floor_us": r4(floor_us), "valu_busy_us": r4(busy * 4.0 / v["simds"] / v["clock_mhz"]) if busy else None,
                 "valu_insts_per_launch": int(sum(v["insts_per_launch"].values())), "profiles": prof.get("tag")})
    try:  # what an EMPTY kernel of this launch shape reads between the same two timestamps (tools/micro/dispatch_overhead.hip): fixed cost of a dispatch
        fixed = json.load(open(os.path.join(ROOT, "profiles", "dispatch_latest.json")))["empty_kernel_event_us"]
        if kus > fixed:
            base.update({"dispatch_fixed_us": fixed, "frac_of_execution": r4(floor_us / (kus - fixed))})
    except Exception:
        pass
    return base


def sub_record(s):
    name = CONFIG_NAME.get((s["workload"], s["K_per_gpu"], s["H"], 1)) or CONFIG_NAME.get((s["workload"], s["K_per_gpu"], s["H"], 8)) or \
        {"mlp32": "ref Dense(32)x3", "nnauv": "ref NNAUVModel s13 a6", "auv": "ref Fossen AUVModel rk2", "nnspeed": "ref NNAUVModelSpeed Dense(16)x3"}.get(s["workload"], "")
    if "bx3" in s["kernel"]:
        name += " +BF16X3"
    if s["kernel"].endswith("3, 0>") and "k_rollout_pc" in s["kernel"]:
        name += " +FP_CONTRACT"
    rf = roofline_of(s)
    # (the line has to stay below 4 KB: a sub-record keeps the kernel, its bound, frac = achieved / peak and the time; peaks and units are
    # DESIGN.md §4's — 157.3 TFLOP/s exact-fp32 MFMA, 2500 bf16, 2458 G SIMD-cycle/s for valu_issue; K = 65536, H = 64 unless given)
    keep = ("kernel", "bound", "frac", "kernel_us", "floor_us", "mfma_busy_frac", "algorithmic_TFLOP_per_s")
    d = {"config": name, "value": r4(s["rollouts_per_s"]), "ms_per_step": r4(s["ms_per_step"]),
         "roofline": {k: rf[k] for k in keep if rf.get(k) is not None}}
    if (s["K_per_gpu"], s["H"]) != (65536, 64):
        d["K"], d["H"] = s["K_per_gpu"], s["H"]
    if s["workload"] in ("pm2d", "pm3d"):
        d["steps_per_batch"] = s["steps"]
    if rf.get("traffic") is not None and name.startswith("configs") and "+" not in name:  # (HBM bytes per launch: the BASELINE configurations' own records)
        d["roofline"]["traffic"] = int(rf["traffic"])
    if s["workload"] == "nnauv" and "bx3" not in s["kernel"]:
        d["weights"] = trained_nnauv()[1]
    return d


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0 and args.gpus > 1:
        self_launch(args)  # never returns
    world = max(world, 1)
    # Only the JSON line may reach stdout: native libraries print there too (RCCL's version banner when the first
    # communicator comes up), so fd 1 points at stderr for the whole run and the line is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product has no CPU path")
    # MPPI_BENCH_ONE_GPU=1: REHEARSAL of the N>1 path on a one-GPU box — every rank drives its shard on cuda:0, rendezvous and
    # bookkeeping over gloo, records over the direct exchange (hipIpc inboxes). RCCL refuses two ranks on one device, so this is
    # the only way to walk the self-launch / per-rank / sub-record code before an 8-GPU node does. The line says "rehearsal".
    rehearsal = os.environ.get("MPPI_BENCH_ONE_GPU") == "1" and world > 1
    # rank mode with nobody having chosen an exchange: the all-gather path first, in-process (its line is kept), the direct exchange
    # afterwards in child processes (direct_exchange_in_children)
    two_phase = world > 1 and not os.environ.get("MPPI_EXCHANGE") and os.environ.get("MPPI_BENCH_CHILD") != "1"
    if two_phase:
        os.environ["MPPI_EXCHANGE"] = "rccl"
        os.environ.setdefault("MPPI_RCCL_CALL", "torch")  # in-process: the path with no native bring-up of its own (a hang here could not be recovered from)
    if rehearsal:
        local_rank = 0
        os.environ.setdefault("MPPI_EXCHANGE", "p2p")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("MPPI_FORCE_EXCHANGE") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import mppi_tf_amd as m

    if world > 1 or dist.is_initialized():
        # the sharded step and its collective run on torch's CURRENT stream: make that an ordinary stream of its own, so that RCCL is
        # never handed the legacy default stream's handle (ShardedController would pass hipStreamLegacy for it)
        torch.cuda.set_stream(torch.cuda.Stream(dev))

    headline = args.workload or "pm3d"
    a, net = WORKLOADS[headline]
    is_mlp = net is not None
    H = args.horizon or (32 if headline == "pm1d" else 64)
    K = args.samples or {"pm1d": 128, "pm2d": 4096}.get(headline, 65536)
    steps = args.steps if args.steps is not None else (20 if is_mlp else 200)
    rn = Runner(args, dev, world, rank, local_rank)
    head_kw = dict(mlp_bf16x3=True) if args.bf16x3 else (dict(fp_contract=True) if args.fp_contract else {})
    r = rn.run(headline, K, H, steps, args.warmup if not is_mlp else min(args.warmup, 3), args.min_time, **head_kw)

    subs = []
    if args.workload is None and not args.no_subrecords:
        if world == 1:  # the other single-GPU BASELINE configs, so that one driver run measures them all
            # (sub-records are not the contract's "exactly K steps": their batches are at least 200 steps, so that the barrier + synchronize at both
            # ends of a batch — 20 us against a 9 us step — does not read as step time under the driver's --steps 20; the batch size is in the record)
            steps_sub = max(steps, 200)
            subs.append(rn.run("pm2d", 4096, 64, steps_sub, args.warmup, args.min_time))
            subs.append(rn.run("pm3d", 65536, 64, steps_sub, args.warmup, 0.3, fp_contract=True))  # MPPI_FLAG_FP_CONTRACT: the opt-in contracted instance
            subs.append(rn.run("mlp", 65536, 64, 20, 3, 0.0))
            subs.append(rn.run("mlp", 65536, 64, 20, 3, 0.0, mlp_bf16x3=True))
            # (sub-second steps: batches repeated for 0.15 s and the median taken, as for the headline — a single 20-step batch of a 0.2 ms step
            # is over before the clocks have settled and read 10 % high: NNAUVModel 0.340 against 0.306 ms)
            subs.append(rn.run("mlp32", 65536, 64, 50, 5, 0.15))
            subs.append(rn.run("mlp32", 65536, 64, 50, 5, 0.15, mlp_bf16x3=True))
            for w, kw in (("nnauv", {}), ("nnauv", dict(mlp_bf16x3=True)), ("auv", {}), ("nnspeed", {})):
                try:
                    subs.append(rn.run(w, 65536, 64, 20, 3, 0.15, **kw))
                except Exception as e:  # a sub-record must never cost the headline
                    sys.stderr.write("bench.py: sub-record %s skipped: %s\n" % (w, e))
        else:  # configs[4]'s per-GPU shape on every rank: K = 65536 x N, H = 128, learned 2x256 model (C5 itself at N = 8)
            # (rehearsal on one GPU: the analytic model — two MLP shards cannot be co-resident on one device, k_rollout_mlp2 owns whole CUs)
            subs.append(rn.run("pm3d" if rehearsal else "mlp", 65536, 128, 10, 2, 0.0))

    parity = None
    if world > 1 or dist.is_initialized():
        try:
            parity = sharded_parity(rn, headline, K, H)
            if not is_mlp and headline not in GEN:
                parity["normalize_cost"] = sharded_parity(rn, headline, K, H, normalize=True)
            if subs:  # configs[4]'s shape: the cross-rank check only (an unsharded 524288-sample MLP step is its own benchmark)
                parity["sub_record"] = sharded_parity(rn, subs[0]["workload"], subs[0]["K_per_gpu"], subs[0]["H"], steps=2, unsharded_ref=False)
        except Exception as e:  # the verdict is a field of the line: its failure is reported there, the measured line still goes out
            parity = {"error": str(e)[:300]}

    out = None
    if rank == 0:
        name = CONFIG_NAME.get((headline, K, H, world), CONFIG_NAME.get((headline, K, H, 1), "not a BASELINE configuration"))
        s_dim = r["s_dim"]
        b = r["batches_s"]
        out = {
            "metric": "rollouts/s (one control step = K rollouts x H steps), point_mass3d H=64",
            "value": float("%.7g" % r["rollouts_per_s"]), "unit": "rollouts/s",
            "n_gpus": world, "steps": r["steps"], "warmup": args.warmup,
            "ms_per_step": float("%.6g" % r["ms_per_step"]), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "batches": {"n": len(b), "median_s": r4(float(np.median(b))), "max_s": r4(max(b))},
            "config": {"workload": "%s %s, K=%d H=%d per GPU (%s), on-device Philox noise, device-resident x/U"
                                   % (headline, ("learned %dx%d MLP model_base" % (net[1], net[0])) if is_mlp else "analytic model", K, H,
                                      ("BASELINE " + name) if name.startswith("configs") else name),
                       "K_global": K * world, "K_per_gpu": K, "H": H, "s_dim": s_dim, "a_dim": a,
                       "lambda": float(cfg_of(headline, H).get("lam", 1.0)), "sigma": "1500*I (6 thrusts, N)" if a == 6 else "0.25*I", "dt": 0.1,
                       "parallelism": "K-shard x%d" % world, "exchange": r["exchange"], "record_floats": r["record_size"]},
            "roofline": dict(roofline_of(r), code_sha=__import__("mppi_tf_amd").build.source_sha()),  # the kernel sources this run executed: tools/summarize_profiles.py tags the profile with it
        }
        if world > 1 or dist.is_initialized():
            out["rccl_ranks"] = dist.get_world_size() if dist.is_initialized() else 1
            if rehearsal:
                out["rehearsal"] = "MPPI_BENCH_ONE_GPU=1: %d ranks share ONE GPU over gloo + hipIpc; not a scaling measurement" % world
            call = r["rccl_note"] if r["rccl_note"] != "not requested" else "torch.distributed all_gather (three C calls + one collective call per step)"
            out["exchange"] = {"used": r["exchange"], "direct_exchange_bring_up": r["p2p_note"], "rccl_call": call if r["exchange"] == "rccl" else None}
            if parity is not None:
                out["parity"] = parity
            out["rank_ms_per_step"] = [r4(q) for q in r["rank_ms_per_step"]]
        if subs:
            out["sub_records"] = [sub_record(s) for s in subs]
            if world > 1:
                out["sub_records"][0]["config"] = "configs[4]" if world == 8 else "configs[4] per-GPU shape, K=%d over %d GPUs" % (65536 * world, world)
                out["sub_records"][0]["exchange"] = subs[0]["exchange"]
                out["sub_records"][0]["rank_ms_per_step"] = [r4(q) for q in subs[0]["rank_ms_per_step"]]
        if world == 1:
            out["ms_per_control_step_sync"] = sync_record(m, headline, H, K, r["mlp"], **head_kw)
            if not is_mlp and headline not in GEN and not head_kw and not args.no_prelaunched:
                try:
                    out["prelaunched"] = dict(prelaunched_record(m, headline, H, K), mode="MPPI_TUNE_PRELAUNCH=1, opt-in: DESIGN 5.2")
                except Exception as e:  # an opt-in path's figure must never cost the line
                    sys.stderr.write("bench.py: pre-launched figure skipped: %s\n" % e)
            for sr, s_ in zip(out.get("sub_records", []), subs):  # configs[1]: the synchronous figure too
                if s_["workload"] == "pm2d" and not args.no_prelaunched:
                    try:
                        q = prelaunched_record(m, "pm2d", s_["H"], s_["K_per_gpu"])
                        if "ms_per_step" in q:
                            sr["prelaunched_ms"] = q["ms_per_step"]
                    except Exception as e:
                        sys.stderr.write("bench.py: pm2d pre-launched figure skipped: %s\n" % e)
                    try:
                        q = sync_record(m, "pm2d", s_["H"], s_["K_per_gpu"], None)
                        sr["sync_ms"] = {k: q[k] for k in ("median", "p95") if k in q}
                        sr["sync_ms"]["armed"] = q["mode"].startswith("armed")
                    except Exception as e:
                        sys.stderr.write("bench.py: pm2d synchronous figure skipped: %s\n" % e)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(headline, H, K, r["mlp"])
        if two_phase:
            # the measured all-gather line leaves this process BEFORE the direct exchange — never run between two real
            # devices — is tried on the same GPUs: to stderr and to a file, so that a fault there costs a field of the line, not the line
            keep = json.dumps(out)
            sys.stderr.write("bench.py: all-gather line (kept): %s\n" % keep)
            sys.stderr.flush()
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "bench_allgather_line.json"), "w") as fh:
                    fh.write(keep + "\n")
            except OSError:
                pass

    p2p_line, p2p_outcome = (None, None)
    if two_phase:
        try:
            p2p_line, p2p_outcome = direct_exchange_in_children(args, world, rank, dist)
        except Exception as e:  # whatever happens there, the measured line is printed
            p2p_line, p2p_outcome = None, "failed: %s" % e

    if rank == 0:
        if two_phase:  # both outcomes; the better headline is the line's
            ex = dict(out.get("exchange") or {})
            ex["rccl"] = {"value": r4(out["value"]), "ms_per_step": r4(out["ms_per_step"]), "used": r["exchange"]}
            if p2p_line is not None:
                used = (p2p_line.get("exchange") or {}).get("used")
                ex["p2p"] = {"value": r4(p2p_line["value"]), "ms_per_step": r4(p2p_line["ms_per_step"]), "used": used}
                if used != "p2p":
                    ex["p2p"]["note"] = "direct exchange not used: %s" % (p2p_line.get("exchange") or {}).get("direct_exchange_bring_up")
                if p2p_line["value"] > out["value"]:
                    for key in ("value", "ms_per_step", "batches", "roofline", "rank_ms_per_step", "steps"):
                        if key in p2p_line:
                            out[key] = p2p_line[key]
                    out["config"]["exchange"] = p2p_line["config"].get("exchange")
                    ex["used"], ex["direct_exchange_bring_up"] = used, (p2p_line.get("exchange") or {}).get("direct_exchange_bring_up")
                    ex["printed"] = "p2p"
                else:
                    ex["printed"] = "rccl"
            else:
                ex["p2p"], ex["printed"] = p2p_outcome, "rccl"
            out["exchange"] = ex
        line = json.dumps(out)
        # the line must stay below 4 KB (what the driver keeps of a run's output): shed the least informative fields first if it does not
        for shed in ("weights", "floor_us", "mfma_busy_frac", "algorithmic_TFLOP_per_s", "steps_per_batch", "bound"):
            if len(line) < 3900:
                break
            for sr in out.get("sub_records", []):
                if not sr["config"].startswith("configs") or shed == "steps_per_batch":
                    sr.pop(shed, None)
                    sr.get("roofline", {}).pop(shed, None)
            line = json.dumps(out)
        while len(line) >= 4090 and out.get("sub_records"):  # last resort: the headline must reach the driver whole — the last sub-records go
            sys.stderr.write("bench.py: line of %d bytes: sub-record %s dropped\n" % (len(line), out["sub_records"][-1]["config"]))
            out["sub_records"].pop()
            line = json.dumps(out)
        sys.stdout.flush()
        os.write(real_stdout, (line + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
