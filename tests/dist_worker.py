"""Worker of tests/test_distributed_gloo.py: one rank of the K-sharded control loop on CPU (gloo).

The sharding logic under test is the product's (mppi_tf_amd.distributed.ShardedController: shard
bounds, record layout, ONE all-gather per step in rank order, replicated finish). The per-shard
arithmetic is supplied by a CPU test double built on the oracle (tests may use oracle/; the product
backend is HipShardBackend and needs a GPU)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402


class OracleShardBackend:
    device = torch.device("cpu")

    def __init__(self, rank, world, k, tau, s_dim, a_dim, sigma, goal, lam, seed, dt=0.1, mass=1.0, normalize_cost=False):
        self.normalize = bool(normalize_cost)
        from mppi_tf_amd.distributed import shard_bounds
        self.lo, self.hi = shard_bounds(k, rank, world)
        self.tau, self.a, self.s, self.lam, self.seed = tau, a_dim, s_dim, lam, seed
        self.sigma = np.asarray(sigma, np.float32)
        self.p = orc.Problem(tau=tau, s=s_dim, a=a_dim, dt=dt, mass=mass, lam=lam, sigma=sigma, goal=goal)
        self.U = np.zeros((tau, a_dim), np.float32)
        self.step_no = 0
        self.record_size = 2 + tau * a_dim

    def partial(self, x, record):
        eps = orc.noise(self.seed, self.step_no, self.lo, self.hi - self.lo, self.tau, self.a, self.sigma)
        c = self.p.rollout_cost(x.numpy(), self.U, eps).astype(np.float64)
        beta = c.min()
        e = np.exp(-(c - beta) / self.lam)
        V = np.tensordot(e, eps.astype(np.float64), axes=(0, 0)).ravel()
        record.copy_(torch.from_numpy(np.concatenate([[beta, e.sum()], V]).astype(np.float32)))

    # normalizeCost (controller_base.py:468-474): this shard's cost range, then the record of the costs normalised with the agreed one
    def cost_range(self, x, rng):
        self.eps_n = orc.noise(self.seed, self.step_no, self.lo, self.hi - self.lo, self.tau, self.a, self.sigma)
        self.c_n = self.p.rollout_cost(x.numpy(), self.U, self.eps_n)
        rng.copy_(torch.tensor([-self.c_n.min(), self.c_n.max()], dtype=torch.float32))  # {-min, max}: both reduce with MAX

    def partial_normalized(self, x, rng, record):
        mn, mx = -np.float32(rng[0].item()), np.float32(rng[1].item())
        c = ((self.c_n - mn) / (mx - mn)).astype(np.float64)
        beta = c.min()
        e = np.exp(-(c - beta) / self.lam)
        V = np.tensordot(e, self.eps_n.astype(np.float64), axes=(0, 0)).ravel()
        record.copy_(torch.from_numpy(np.concatenate([[beta, e.sum()], V]).astype(np.float32)))

    def finish(self, records, n_records, u):
        Unew = orc.combine_records(records.numpy().reshape(n_records, -1), self.U, self.lam)
        u.copy_(torch.from_numpy(Unew[0].copy()))
        self.U = np.vstack([Unew[1:], np.zeros((1, self.a), np.float32)])
        self.step_no += 1


class P2PDoubleBackend(OracleShardBackend):
    """Adds the direct-exchange hooks of HipShardBackend so that ShardedController's bring-up (export -> gather the
    handles -> open -> attach -> probe x3, a vote after each phase) runs on CPU. mode: "ok" | "open_fails" |
    "probe_fails" (the failure happens on the LAST rank only: the vote must still send every rank to the all-gather).
    The step itself moves the records with a gloo all-gather: only the control flow is under test here."""

    def __init__(self, rank, world, mode, **cfg):
        super().__init__(rank, world, **cfg)
        self.rank, self.world, self.mode = rank, world, mode
        self.attached, self.probes, self.p2p_steps, self.failed = None, 0, 0, False

    def p2p_export(self):
        return 1000 + self.rank, b"ipc-handle-of-rank-%02d" % self.rank

    def p2p_open(self, ipc_handle):
        if self.mode == "open_fails" and self.rank == self.world - 1:
            raise RuntimeError("hipIpcOpenMemHandle: invalid argument (test double)")
        assert ipc_handle.startswith(b"ipc-handle-of-rank-")
        return 1000 + int(ipc_handle[-2:])

    def p2p_attach(self, ptrs, timeout_ms):
        assert ptrs == [1000 + g for g in range(self.world)] and timeout_ms > 0
        self.attached = list(ptrs)

    def p2p_probe(self):
        self.probes += 1
        return not (self.mode == "probe_fails" and self.rank == self.world - 1)

    def p2p_step(self, x, u):
        if self.mode == "deadline" and self.p2p_steps == 1 and not self.failed:
            # what HipShardBackend does when the pinned flag is up (MPPI_ERR_EXCHANGE): refuse before enqueuing. Here the
            # ranks have meanwhile drifted apart the way zero-update steps let them (a different U and counter per rank).
            self.failed = True
            self.U = self.U + np.float32(1e-3 * (self.rank + 1))
            self.step_no += self.rank
            e = RuntimeError("mppi status 8: direct exchange: a packet missed its deadline")
            e.status = 8
            raise e
        self.p2p_steps += 1
        rec = torch.zeros(self.record_size)
        recs = torch.zeros(self.world * self.record_size)
        self.partial(x, rec)
        dist.all_gather_into_tensor(recs, rec)
        self.finish(recs, self.world, u)

    def p2p_timed_out(self):
        return False

    # state hand-over used by ShardedController.resync
    def action_sequence(self):
        return torch.from_numpy(self.U.copy())

    def step_counter(self):
        return self.step_no

    def set_state(self, U, step):
        self.U, self.step_no = U.numpy().astype(np.float32).copy(), int(step)


def run(cfg, n_steps, world_override=None):
    from mppi_tf_amd.distributed import ExchangeTimeout, ShardedController
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    mode = os.environ.get("MPPI_TEST_P2P", "")
    be = P2PDoubleBackend(rank, world, mode, **cfg) if mode else OracleShardBackend(rank, world, **cfg)
    ctl = ShardedController(backend=be)
    A, B = orc.pm_matrices(0.1, 1.0, cfg["s_dim"], cfg["a_dim"])
    x = np.zeros(cfg["s_dim"], np.float32)
    us = []
    resyncs = 0
    for _ in range(n_steps):
        try:
            u = ctl.next(torch.from_numpy(x.copy())).numpy().copy()
        except ExchangeTimeout:  # every rank: drop the direct path, adopt rank 0's U and step counter, go on over the all-gather
            ctl.resync()
            resyncs += 1
            u = ctl.next(torch.from_numpy(x.copy())).numpy().copy()
        us.append(u.tolist())
        x = orc.model_step(A, B, x[None], u[None])[0]
    ctl.check()
    return dict(rank=rank, world=world, lo=be.lo, hi=be.hi, u=us, U=be.U.tolist(), exchange=ctl.exchange, note=ctl.p2p_note,
                p2p_steps=getattr(be, "p2p_steps", 0), probes=getattr(be, "probes", 0), resyncs=resyncs,
                step_no=be.step_no)


if __name__ == "__main__":
    cfg = json.loads(os.environ["MPPI_TEST_CFG"])
    out = os.environ["MPPI_TEST_OUT"]
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    res = run(cfg, int(os.environ["MPPI_TEST_STEPS"]))
    with open("%s.%d" % (out, res["rank"]), "w") as fh:
        json.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()
