#!/usr/bin/env python3
"""bench.py — rollouts/s of the MPPI control step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload pm3d|pm2d|pm1d|mlp|mlp32] [--horizon H] [--samples K_PER_GPU]
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE control step of the hot path: K rollouts x H model steps + costs + soft-min update + shift, with the
state x and the nominal sequence U already resident in HBM and the noise drawn on the device (Philox).
Headline workload at N=1: BASELINE configs[2], point_mass3d analytic, K=65536, H=64 — the configuration the metric is
quoted on. Every BASELINE config is launchable:
    configs[1]  --workload pm2d --samples 4096                 (also a sub-record of the default run)
    configs[2]  (default)
    configs[3]  --workload mlp                                 (also a sub-record of the default run, with the split-bf16 variant)
    configs[4]  --workload mlp --horizon 128 --gpus 8          (K = 65536 per rank = 524288 in all)
    (--workload mlp32: the reference's own Dense(32) x3 network, not a BASELINE config; a sub-record of the default run)
N>1 is WEAK scaling: every rank keeps --samples rollouts of a (samples x N)-sample controller, one exchange of the
(beta, eta, V) record per step. Rank 0 prints ONE JSON line (value = whole-job rollouts/s, max-over-ranks time).

Timing: W warm-up steps, then batches of EXACTLY K steps, each bracketed by barrier + synchronize on both sides and
reduced with MAX over ranks; batches repeat until --min-time seconds have been timed (a 200-step batch of the analytic
workload is 4 ms: one batch is a noisy sample), `value` is the MEDIAN batch; all batch times are in the line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HID = 256                                 # BASELINE configs[3]: learned 2x256 MLP model_base
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3              # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32
MFMA_BF16_PEAK_TFLOPS = 2500.0
GOALS = {1: [1.0, 0.0], 2: [1.0, 0.0, 0.0, 0.0], 3: [1.0, 0.0, 0.5, 0.0, 0.75, 0.0]}  # MuJoCo target sites
# workload -> (a_dim, learned model: None | (hidden width, hidden layers))
WORKLOADS = {"pm1d": (1, None), "pm2d": (2, None), "pm3d": (3, None), "mlp": (3, (256, 2)), "mlp32": (3, (32, 3))}
CONFIG_NAME = {("pm1d", 128, 32, 1): "BASELINE configs[0]", ("pm2d", 4096, 64, 1): "BASELINE configs[1]",
               ("pm3d", 65536, 64, 1): "BASELINE configs[2]", ("mlp", 65536, 64, 1): "BASELINE configs[3]",
               ("mlp", 65536, 128, 8): "BASELINE configs[4] (K = 524288 over 8 GPUs)"}


def cfg_of(a, H):
    return dict(tau=H, s_dim=2 * a, a_dim=a, dt=0.1, mass=1.0, lam=1.0, sigma=(0.25 * np.eye(a)).astype(np.float32),
                goal=GOALS[a], seed=1)


def synthetic_mlp(a=3, seed=0, hid=HID, n_hidden=2):
    """(s+a)->256->256->s ReLU, U(-1/sqrt(fan_in), 1/sqrt(fan_in)), last layer x0.1, identity normalisation;
    hid=32, n_hidden=3 is the reference's own network shape (nn_model.py:54-60)."""
    rng = np.random.default_rng(seed)
    dims = [3 * a] + [hid] * n_hidden + [2 * a]
    W = [(rng.uniform(-1, 1, (dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(n_hidden + 1)]
    b = [(rng.uniform(-1, 1, dims[i + 1]) / np.sqrt(dims[i])).astype(np.float32) for i in range(n_hidden + 1)]
    W[-1] *= 0.1
    b[-1] *= 0.1
    return dict(W=W, b=b)


def work_per_state_step(a, mlp):
    """algorithmic work of one (k, t) pair: (bytes, flop)"""
    s = 2 * a
    flop = 6 * s + 5 * a + 3
    if mlp:
        hid, n_hidden = mlp
        flop += 2 * ((s + a) * hid + (n_hidden - 1) * hid * hid + hid * s)
    return 12 * a, flop  # bytes: the noise written once and read twice, fp32


def cpu_baseline(a, H, K, mlp, budget_s=12.0):
    """The CPU restatement (oracle/, OpenMP over samples) timed on this box's host cores on a bounded sample of the SAME
    workload: whole control steps (noise + rollouts + update) at the same H — the same K for the analytic model, K=4096
    for the MLP (stated in `sample`)."""
    from oracle import oracle as orc
    Kc = K if mlp is None else min(K, 4096)
    c = cfg_of(a, H)
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
    return {"value": Kc * n / el, "unit": "rollouts/s", "cores": orc.num_threads(), "kind": "port",
            "ms_per_step": 1e3 * el / n,
            "sample": "%d whole control steps of point_mass%dd%s K=%d H=%d (Philox noise + rollouts + update), "
                      "OpenMP over samples; the reference itself (TensorFlow) is not runnable here"
                      % (n, a, "" if mlp is None else " + %dx%d MLP model" % (len(mlp["W"]) - 1, mlp["W"][0].shape[1]), Kc, H)}


def sync_latency(m, a, H, K, mlp, steps=200, warmup=20):
    """Host-synchronous closed loop: mppi_next(x)->u with the plant stepped on the host (the shape of the reference's
    loop, main.cpp:37-43). Median / p95 ms per control step."""
    h = m.Handle(k=K, mlp=mlp, **cfg_of(a, H))
    if mlp is not None:
        steps, warmup = 20, 3
    x = np.zeros(2 * a, np.float32)
    dt, ts = 0.1, []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        u = h.next(x)
        t1 = time.perf_counter()
        if i >= warmup:
            ts.append(t1 - t0)
        for j in range(a):  # point-mass plant, fp32, same model
            x[2 * j] = x[2 * j] + dt * x[2 * j + 1] + (dt * dt / 2) * u[j]
            x[2 * j + 1] = x[2 * j + 1] + dt * u[j]
    ts = np.sort(np.asarray(ts)) * 1e3
    h.close()
    return float(np.median(ts)), float(ts[int(0.95 * (len(ts) - 1))])


def measured(name):
    """profiles/<name>_latest.json (written by tools/summarize_profiles.py from rocprofv3 --pmc passes), or None when it
    is absent or was taken on other kernel sources than the ones this run executes."""
    import mppi_tf_amd as m
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name + "_latest.json")))
    except Exception:
        return None, "no profiles/%s_latest.json" % name
    sha = m.build.source_sha()
    if d.get("code_sha") != sha:
        return None, "profiles/%s_latest.json was measured on kernel sources %s, this run executes %s" % (name, d.get("code_sha"), sha)
    return d, None


def valu_roofline(kernel, kernel_ms):
    """The limiter of the analytic rollout kernel is vector-instruction issue, not HBM: instructions per launch by class
    (rocprofv3 SQ_INSTS_VALU_* counters) x issue cycles per instruction of that class (tools/micro/valu_issue.hip, an
    ISA-verified micro-benchmark, several waves per SIMD) / (1024 SIMDs x the clock) = the time the launch's vector
    instructions need at full issue rate."""
    d, why = measured("valu")
    if d is None or d.get("kernel") != kernel:
        return {"note": why or "profiles/valu_latest.json describes %s, this run launched %s" % (d.get("kernel"), kernel)}
    cyc = sum(d["insts_per_launch"][c] * d["cycles_per_inst"][c] for c in d["insts_per_launch"])
    floor_us = cyc / (d["simds"] * d["clock_mhz"])
    out = dict(d)
    out.update({"floor_us": floor_us, "kernel_us": 1e3 * kernel_ms, "frac": floor_us / (1e3 * kernel_ms) if kernel_ms > 0 else None})
    return out


class Runner:
    def __init__(self, args, dev, world, rank, local_rank):
        self.args, self.dev, self.world, self.rank, self.local_rank = args, dev, world, rank, local_rank

    def barrier(self):
        torch.cuda.synchronize(self.dev)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, v):
        t = torch.tensor([v], dtype=torch.float64, device=self.dev)
        if dist.is_initialized():
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def run(self, workload, K, H, steps, warmup, min_time, **handle_kw):
        """-> dict of measurements of one workload (K rollouts per rank)"""
        from mppi_tf_amd.distributed import ExchangeTimeout, ShardedController
        a, net = WORKLOADS[workload]
        is_mlp = net is not None
        mlp = synthetic_mlp(a, 0, *net) if is_mlp else None
        x = torch.zeros(2 * a, dtype=torch.float32, device=self.dev)
        ctl = ShardedController(device_index=self.local_rank, k=K * self.world, mlp=mlp, exchange=os.environ.get("MPPI_EXCHANGE", "auto"),
                                p2p_timeout_ms=1000, **cfg_of(a, H), **handle_kw)
        assert ctl.backend.h.k_local == K

        def steps_timed(n):
            """n steps between two barriers; a missed direct-exchange deadline sends every rank to the all-gather path"""
            nonlocal ctl
            for attempt in range(2):
                self.barrier()
                t0 = time.perf_counter()
                try:
                    for _ in range(n):
                        ctl.next(x)
                    self.barrier()
                    ctl.check()
                    bad = 0
                except ExchangeTimeout:
                    bad = 1
                el = time.perf_counter() - t0
                if self.max_over_ranks(bad) == 0:
                    return self.max_over_ranks(el)
                ctl.resync()
            raise RuntimeError("the all-gather path cannot time out")

        steps_timed(warmup)
        batches = [steps_timed(steps)]
        n_more = int(min(200, max(0, np.ceil((min_time - batches[0]) / max(batches[0], 1e-9)))))
        n_more = int(self.max_over_ranks(n_more))
        for _ in range(n_more):
            batches.append(steps_timed(steps))
        el = float(np.median(batches))
        # kernel durations: the same K steps with HIP events bound to each launch (the dispatches' own begin/end)
        h = ctl.backend.h
        n_prof = min(steps, 200)
        h.profile_begin(n_prof)
        for _ in range(n_prof):
            ctl.next(x)
        torch.cuda.synchronize(self.dev)
        roll_ms, fin_ms, n_prof = h.profile_end()
        assert np.isfinite(ctl.u.cpu().numpy()).all()
        bytes_ss, flop_ss = work_per_state_step(a, net)
        state_steps = K * H
        res = {"workload": workload, "K_per_gpu": K, "H": H, "a_dim": a, "steps": steps, "batches_s": batches,
               "rollouts_per_s": K * self.world * steps / el, "ms_per_step": 1e3 * el / steps,
               "kernel": h.rollout_kernel_name(), "kernel_ms_avg": roll_ms, "finish_kernel_ms_avg": fin_ms, "launches_timed": n_prof,
               "algorithmic_bytes_per_launch": bytes_ss * state_steps + 8 * K, "algorithmic_flop_per_launch": flop_ss * state_steps,
               "exchange": ctl.exchange, "p2p_note": ctl.p2p_note, "record_size": h.record_size, "mlp": mlp}
        del ctl
        return res


def roofline_of(r):
    if r["workload"] in ("mlp", "mlp32"):
        tf = r["algorithmic_flop_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e12 if r["kernel_ms_avg"] > 0 else 0.0
        note = ("exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): the 1e-5-class parity target rules out plain bf16; weights stationary in "
                "registers (a0-a255), activations in LDS. On gfx950 the f32-input MFMA runs at the f32 vector rate and does not "
                "overlap with the vector ALU (tools/micro/mfma_f32_shadow.hip): the attainable rate is peak x MFMA cycles / "
                "(MFMA + vector cycles) — the MFMAs alone measure 156 TFLOP/s (tools/micro/mlp2_bench.hip), layer 3, relu, the "
                "state update and the layer-1 MFMAs are the rest")
        if r["workload"] == "mlp32":
            note = ("the reference's own network shape (Dense(32, relu) x3 + Dense(s), nn_model.py:54-60) on k_rollout_mlp32: a 32-wide "
                    "layer is one v_mfma_f32_32x32x2_f32 tile and the accumulator layout of one layer is the B-operand layout of the "
                    "next, so the layers chain through registers (weights and biases stationary, no LDS, no barrier); 2 waves x 32 "
                    "rollouts per tile, the output layer on the vector ALU")
        mf, why = measured("mfma")  # PMC passes of tools/collect_profiles.sh <tag> mlp, tagged with the kernel sources' hash
        ok = mf is not None and mf.get("kernel") == r["kernel"]
        return {"bound": "mfma", "kernel": r["kernel"], "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / MFMA_F32_PEAK_TFLOPS, "traffic": mf.get("hbm_bytes_per_launch") if ok else None,
                "mfma_busy_frac": mf.get("mfma_busy_frac") if ok else None,
                "other_vector_insts_per_launch": mf.get("other_vector_insts_per_launch") if ok else None,
                "pmc_note": None if ok else (why or "profiles/mfma_latest.json describes another kernel instance"),
                "algorithmic_flop_per_launch": r["algorithmic_flop_per_launch"],
                "kernel_ms_avg": r["kernel_ms_avg"], "finish_kernel_ms_avg": r["finish_kernel_ms_avg"], "launches_timed": r["launches_timed"],
                "note": note}
    ach = r["algorithmic_bytes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e9 if r["kernel_ms_avg"] > 0 else 0.0
    tr, why = measured("traffic")
    traffic = tr.get("hbm_bytes_per_launch") if tr and tr.get("kernel") == r["kernel"] else None
    return {"bound": "hbm", "kernel": r["kernel"], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_note": None if traffic is not None else (why or "profiles/traffic_latest.json describes another kernel instance"),
            "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"], "algorithmic_flop_per_launch": r["algorithmic_flop_per_launch"],
            "kernel_ms_avg": r["kernel_ms_avg"], "finish_kernel_ms_avg": r["finish_kernel_ms_avg"], "launches_timed": r["launches_timed"],
            "timing": "HIP events bound to each launch of the kernel on its stream (hipExtLaunchKernel start/stop events = the "
                      "dispatch's own begin/end, the quantity rocprofv3 reports in profiles/)",
            "valu": valu_roofline(r["kernel"], r["kernel_ms_avg"]),
            "note": "achieved = ALGORITHMIC bytes (SURVEY 8d: noise written once + read twice, 12*a B per state-step) per launch / "
                    "kernel time; the noise is generated and consumed on-chip, so the physical HBM traffic (`traffic`, PMC) is "
                    "~20x lower and the algorithmic rate can exceed the HBM peak: frac says how the kernel compares with ANY "
                    "kernel that materialises the noise, not how close it is to its own limit. Its limiter is vector-instruction "
                    "issue (Philox4x32-10 + Box-Muller): `valu` prices the launch's instructions at the measured issue rates."}


def main():
    # Only the JSON line may reach stdout: native libraries print there too (RCCL's version banner when the first
    # communicator comes up), so fd 1 points at stderr for the whole run and the line is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="K steps per timed batch (default 200; 20 for the MLP workload, ~5 ms per step)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None,
                    help="default: pm3d (BASELINE configs[2], the metric's config) plus sub-records of configs[1] and [3]")
    ap.add_argument("--horizon", type=int, default=None, help="H (default 64; 32 for pm1d)")
    ap.add_argument("--samples", type=int, default=None, help="rollouts PER GPU (default 65536; 4096 for pm2d, 128 for pm1d)")
    ap.add_argument("--min-time", type=float, default=0.2, help="repeat the K-step batch until this many seconds are timed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-subrecords", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("MPPI_FORCE_EXCHANGE") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import mppi_tf_amd as m

    headline = args.workload or "pm3d"
    a, net = WORKLOADS[headline]
    is_mlp = net is not None
    H = args.horizon or (32 if headline == "pm1d" else 64)
    K = args.samples or {"pm1d": 128, "pm2d": 4096}.get(headline, 65536)
    steps = args.steps if args.steps is not None else (20 if is_mlp else 200)
    rn = Runner(args, dev, world, rank, local_rank)
    r = rn.run(headline, K, H, steps, args.warmup if not is_mlp else min(args.warmup, 3), args.min_time)

    subs = []
    if args.workload is None and world == 1 and not args.no_subrecords:
        # the other single-GPU BASELINE configs, so that one driver run measures them all
        subs.append(rn.run("pm2d", 4096, 64, steps, args.warmup, args.min_time))
        subs.append(rn.run("mlp", 65536, 64, 20, 3, 0.0))
        subs.append(rn.run("mlp", 65536, 64, 20, 3, 0.0, mlp_bf16x3=True))
        subs.append(rn.run("mlp32", 65536, 64, 50, 5, 0.0))

    if rank == 0:
        cfg = cfg_of(a, H)
        name = CONFIG_NAME.get((headline, K, H, world), CONFIG_NAME.get((headline, K, H, 1), "not a BASELINE configuration"))
        out = {
            "metric": "rollouts/s (one control step = K rollouts x H steps), point_mass3d H=64",
            "value": r["rollouts_per_s"], "unit": "rollouts/s",
            "n_gpus": world, "steps": r["steps"], "warmup": args.warmup,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "state_steps_per_s": r["rollouts_per_s"] * H,
            "batches": {"n": len(r["batches_s"]), "steps_each": r["steps"], "seconds": r["batches_s"],
                        "what": "value = K*N*steps / median batch; every batch is bracketed by barrier + synchronize, max over ranks"},
            "config": {"workload": "point_mass%dd %s, K=%d H=%d per GPU (%s), on-device Philox noise, device-resident x/U"
                                   % (a, ("learned %dx%d MLP model_base" % (net[1], net[0])) if is_mlp else "analytic model", K, H, name),
                       "K_global": K * world, "K_per_gpu": K, "H": H, "s_dim": 2 * a, "a_dim": a,
                       "lambda": 1.0, "sigma": "0.25*I", "dt": 0.1, "mass": 1.0,
                       "parallelism": "K-shard x%d, %s" % (world, {
                           "none": "single shard, no exchange",
                           "p2p": "records (%d floats) exchanged as peer stores over xGMI inside the finish kernel (%s)"
                                  % (r["record_size"], r["p2p_note"]),
                           "rccl": "one RCCL all-gather of %d floats per step (direct exchange: %s)"
                                   % (r["record_size"], r["p2p_note"])}[r["exchange"]]),
                       "exchange": r["exchange"]},
            "roofline": roofline_of(r),
        }
        if subs:
            def sub(s):
                d = {"config": CONFIG_NAME.get((s["workload"], s["K_per_gpu"], s["H"], 1), "the reference's Dense(32) x3 network (nn_model.py:54-60), not a BASELINE configuration"
                                               if s["workload"] == "mlp32" else "") + (" + MPPI_FLAG_MLP_BF16X3" if "bx3" in s["kernel"] else ""),
                     "workload": s["workload"], "K": s["K_per_gpu"], "H": s["H"], "value": s["rollouts_per_s"], "unit": "rollouts/s",
                     "ms_per_step": s["ms_per_step"], "steps": s["steps"], "batches": len(s["batches_s"]), "roofline": roofline_of(s)}
                if "bx3" in s["kernel"]:  # the matrix cores execute 3 bf16 products per fp32 term
                    ex = 3 * s["algorithmic_flop_per_launch"] / (s["kernel_ms_avg"] * 1e-3) / 1e12
                    d["roofline"].update({"peak": MFMA_BF16_PEAK_TFLOPS, "achieved": ex, "frac": ex / MFMA_BF16_PEAK_TFLOPS,
                                          "algorithmic_TFLOP_per_s": ex / 3,
                                          "note": "split-bf16: 3 v_mfma_f32_32x32x16_bf16 products per fp32 term (achieved = executed bf16 "
                                                  "FLOP/s against the bf16 dense peak); sample costs within 1e-6 relative of fp64"})
                return d
            out["sub_records"] = [sub(s) for s in subs]
        if world == 1:
            med, p95 = sync_latency(m, a, H, K, r["mlp"])
            out["ms_per_control_step_sync"] = {"median": med, "p95": p95,
                                               "what": "host-synchronous mppi_next(x)->u incl. H2D x, D2H u, closed-loop steps"}
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(a, H, K, r["mlp"])
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
