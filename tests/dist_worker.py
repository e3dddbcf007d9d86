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

    def __init__(self, rank, world, k, tau, s_dim, a_dim, sigma, goal, lam, seed, dt=0.1, mass=1.0):
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

    def finish(self, records, n_records, u):
        Unew = orc.combine_records(records.numpy().reshape(n_records, -1), self.U, self.lam)
        u.copy_(torch.from_numpy(Unew[0].copy()))
        self.U = np.vstack([Unew[1:], np.zeros((1, self.a), np.float32)])
        self.step_no += 1


def run(cfg, n_steps, world_override=None):
    from mppi_tf_amd.distributed import ShardedController
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    be = OracleShardBackend(rank, world, **cfg)
    ctl = ShardedController(backend=be)
    A, B = orc.pm_matrices(0.1, 1.0, cfg["s_dim"], cfg["a_dim"])
    x = np.zeros(cfg["s_dim"], np.float32)
    us = []
    for _ in range(n_steps):
        u = ctl.next(torch.from_numpy(x.copy())).numpy().copy()
        us.append(u.tolist())
        x = orc.model_step(A, B, x[None], u[None])[0]
    return dict(rank=rank, world=world, lo=be.lo, hi=be.hi, u=us, U=be.U.tolist())


if __name__ == "__main__":
    cfg = json.loads(os.environ["MPPI_TEST_CFG"])
    out = os.environ["MPPI_TEST_OUT"]
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    res = run(cfg, int(os.environ["MPPI_TEST_STEPS"]))
    with open("%s.%d" % (out, res["rank"]), "w") as fh:
        json.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()
