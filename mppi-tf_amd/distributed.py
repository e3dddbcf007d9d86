"""K-sharding of the MPPI step across GPUs: one process per GPU, ONE collective per control step.

SURVEY.md §8e (new work — the reference is single-device): rank g owns samples
[g·K/G, (g+1)·K/G); x, U, goal, Σ⁻¹ are replicated; Philox counters use the GLOBAL sample index,
so results do not depend on G.  Per control step each rank
  1. rolls its shard and reduces it to one record (β_g, η_g, V_g[tau·a])       (HIP kernels)
  2. all-gathers the records: G·(2+tau·a) floats, e.g. 8 × 386 × 4 B = 12.4 KB (RCCL over xGMI;
     latency-bound, far below any per-link bandwidth limit)
  3. combines them in rank order with r_g = exp(-(β_g-β)/λ) and applies U' = U + V/η, shift —
     replicated on every rank, bit-identical across ranks (same inputs, same fixed order).
Two realisations of step 2, same bits: (i) ONE RCCL all-gather between two kernel launches, and (ii) the
direct exchange of include/mppi_c.h (mppi_shard_p2p_*): the finish kernel itself stores the record into every
peer's inbox over xGMI and spins for theirs — no collective launch on the critical path of a ~25 µs step.
(ii) is brought up with a self-test and a vote over all ranks; if any rank cannot map a peer, or a probe
packet does not arrive, every rank uses (i).  MPPI_EXCHANGE=rccl|p2p|auto (default auto) picks.
normalizeCost=True (controller_base.py:468-474) needs the min and max cost over ALL samples: such a controller runs a second,
2-float collective per step (ONE all-reduce(MAX) of {-min, max}) between mppi_shard_cost_range and mppi_shard_partial_normalized,
and always exchanges its records by all-gather.
torch is plumbing here: device buffers, the current stream, and torch.distributed (backend
"nccl" is RCCL on ROCm; "gloo" drives the CPU test of this file's logic with a test backend).
"""
import os

import torch
import torch.distributed as dist


class ExchangeTimeout(RuntimeError):
    """The direct record exchange missed a deadline; see ShardedController.next / resync."""


class HipShardBackend:
    """The product backend: this rank's shard on its GPU through the C-ABI (mppi_shard_partial /
    mppi_shard_finish). Fails loudly when the HIP library or the GPU is missing."""

    def __init__(self, rank, world, device_index=0, **cfg):
        from ._lib import Handle
        if not torch.cuda.is_available():
            raise RuntimeError("HipShardBackend needs a GPU: there is no CPU fallback")
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.h = Handle(shard_rank=rank, shard_count=world, device=device_index, **cfg)
        self.record_size, self.a = self.h.record_size, self.h.a
        self.normalize = bool(cfg.get("normalize_cost", False))

    def _stream(self):
        """torch's CURRENT stream, for Handle's stream arguments (Handle._stream maps torch's default stream, whose handle is 0, to
        hipStreamLegacy: the C-ABI reads 0 as "the handle's own stream", which nothing of torch's is ordered against — the record would race
        with the collective that reads it; found by the two-rank rehearsal of r04, where the gathered records were one step stale)."""
        return torch.cuda.current_stream(self.device)

    def partial(self, x, record):
        self.h.shard_partial(x.data_ptr(), record.data_ptr(), self._stream())

    def finish(self, records, n_records, u):
        self.h.shard_finish(records.data_ptr(), n_records, u.data_ptr(), self._stream())

    # normalizeCost: the global min / max cost is a second, 2-float exchange (mppi_shard_cost_range / _partial_normalized)
    def cost_range(self, x, rng):
        self.h.shard_cost_range(x.data_ptr(), rng.data_ptr(), self._stream())

    def partial_normalized(self, x, rng, record):
        self.h.shard_partial_normalized(x.data_ptr(), rng.data_ptr(), record.data_ptr(), self._stream())

    def step(self, x, u):
        """unsharded whole step (mppi_next_device): no record round trip"""
        self.h.next_device(x.data_ptr(), u.data_ptr(), self._stream())

    def shard_step(self, x, u, coll):
        """the whole sharded step in ONE C call (mppi_shard_step): record -> the collectives `coll` points at -> finish"""
        self.h.shard_step(x.data_ptr(), u.data_ptr(), coll, self._stream())

    def make_collectives(self, rank, world, group):
        """this rank's own RCCL communicator behind an mppi_collectives struct (collective over `group`)"""
        from .rccl import RcclComm
        return RcclComm(rank, world, group)

    # direct exchange (mppi_shard_p2p_*)
    def p2p_export(self):
        return self.h.p2p_export()

    def p2p_open(self, ipc_handle):
        return self.h.p2p_open(ipc_handle)

    def p2p_attach(self, ptrs, timeout_ms):
        self.h.p2p_attach(ptrs, timeout_ms)

    def p2p_probe(self):
        return self.h.p2p_probe(self._stream())

    def p2p_step(self, x, u):
        self.h.p2p_step(x.data_ptr(), u.data_ptr(), self._stream())

    def p2p_timed_out(self):
        return self.h.p2p_timed_out()

    def action_sequence(self):
        torch.cuda.current_stream(self.device).synchronize()
        return torch.from_numpy(self.h.get_action_sequence())

    def step_counter(self):
        torch.cuda.current_stream(self.device).synchronize()
        return self.h.get_step_counter()

    def set_state(self, U, step):
        torch.cuda.current_stream(self.device).synchronize()
        self.h.set_action_sequence(U.numpy())
        self.h.set_step_counter(step)


class ShardedController:
    """next(x) on every rank of `group` = one MPPI control step over all K samples.

    backend: an object with .device, .record_size, .a, .partial(x, record), .finish(records, n, u)
    (default: HipShardBackend built from cfg; tests inject a CPU test double over gloo).
    """

    def __init__(self, backend=None, group=None, device_index=0, exchange=None, p2p_timeout_ms=2000, **cfg):
        """cfg: Handle arguments (k = GLOBAL sample count, tau, s_dim, a_dim, sigma, goal, mlp, ...).
        exchange: "auto" (direct exchange if its self-test passes on every rank, else the all-gather), "p2p"
        (direct exchange or raise), "rccl" (all-gather); default from MPPI_EXCHANGE, else "auto".
        MPPI_RCCL_CALL=torch keeps the all-gather path on torch.distributed (three C calls + one collective call per step) instead
        of mppi_shard_step with the controller's own communicator (A/B timing, tools/time_sharded.py)."""
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = backend if backend is not None else HipShardBackend(self.rank, self.world, device_index, **cfg)
        dev, n = self.backend.device, self.backend.record_size
        self.record = torch.zeros(n, dtype=torch.float32, device=dev)
        self.records = torch.zeros(self.world * n, dtype=torch.float32, device=dev)
        self.u = torch.zeros(self.backend.a, dtype=torch.float32, device=dev)
        # normalizeCost (controller_base.py:468-474) needs the min and max cost over ALL ranks' samples: a second collective per step
        self.normalize = bool(getattr(self.backend, "normalize", False))
        self.range = torch.zeros(2, dtype=torch.float32, device=dev)
        # MPPI_FORCE_EXCHANGE=1: take the sharded path (records -> exchange -> finish) even with one rank
        # (exercises the N>1 code path on a single-GPU box)
        self.force_exchange = os.environ.get("MPPI_FORCE_EXCHANGE") == "1"
        exchange = exchange or os.environ.get("MPPI_EXCHANGE", "auto")
        if exchange not in ("auto", "p2p", "rccl"):
            raise ValueError("exchange must be auto, p2p or rccl")
        self.p2p, self.p2p_note = False, "not requested"
        sharded = self.world > 1 or self.force_exchange
        if self.normalize and exchange == "p2p":
            raise RuntimeError("normalize_cost: the direct exchange carries the records only; use exchange='rccl' or 'auto'")
        if sharded and exchange != "rccl" and hasattr(self.backend, "p2p_export") and not self.normalize:
            self.p2p, self.p2p_note = self._bring_up_p2p(p2p_timeout_ms)
            if exchange == "p2p" and not self.p2p:
                raise RuntimeError("direct exchange requested but unavailable: " + self.p2p_note)
        self.exchange = "none" if not sharded else ("p2p" if self.p2p else "rccl")
        # The collective path as ONE C call per step (mppi_shard_step calling ncclAllGather itself) where the backend offers it and
        # the job runs on RCCL (or has one rank); else three calls + a torch.distributed collective (gloo tests, fallback).
        self.rccl, self.rccl_note = None, "not requested"
        if self.exchange == "rccl" and os.environ.get("MPPI_RCCL_CALL", "c") != "torch":
            self.rccl, self.rccl_note = self._bring_up_rccl()

    def _vote(self, ok):
        """True iff ok on every rank"""
        if not dist.is_initialized():
            return bool(ok)
        on_gpu = dist.get_backend(self.group) == "nccl"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.backend.device if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(t.item())

    def _bring_up_rccl(self):
        """-> (RcclComm or None, note). A collective when the job has more than one rank: every rank takes the same branch."""
        if not hasattr(self.backend, "make_collectives"):
            return None, "backend without mppi_shard_step"
        if dist.is_initialized() and self.world > 1 and dist.get_backend(self.group) != "nccl":
            return None, "the job's backend is not RCCL"
        comm, note = None, ""
        # two votes: ncclCommInitRank is itself a collective — a rank that cannot even load the library must say so BEFORE the others
        # enter it, or they would wait in it for a rank that never comes
        try:
            from .rccl import load
            load()
            have = True
        except Exception as e:
            have, note = False, "librccl: %s" % e
        if not self._vote(have):
            return None, note or "a peer's process holds no librccl"
        try:
            comm = self.backend.make_collectives(self.rank, self.world, self.group)
        except Exception as e:  # ncclCommInitRank refused, ...
            note = "RCCL communicator: %s" % e
        if not self._vote(comm is not None):
            if comm is not None:
                comm.close()
            return None, note or "a peer could not create its RCCL communicator"
        return comm, "one C call per step (mppi_shard_step -> ncclAllGather), own communicator of %d rank(s)" % self.world

    def _bring_up_p2p(self, timeout_ms):
        """export -> exchange IPC handles -> open -> attach -> probe x3, with a vote after each phase that can fail.
        Every rank takes the same branch: the votes are collectives."""
        note, ptrs = "", None
        try:
            own_ptr, ipc = self.backend.p2p_export()
            ok = True
        except Exception as e:  # HIP refused the uncached allocation or the IPC export
            own_ptr, ipc, ok, note = None, None, False, "export: %s" % e
        if dist.is_initialized():
            handles = [None] * self.world
            dist.all_gather_object(handles, ipc, group=self.group)
        else:
            handles = [ipc]
        if ok:
            try:
                ptrs = [own_ptr if g == self.rank else self.backend.p2p_open(handles[g]) for g in range(self.world)]
                self.backend.p2p_attach(ptrs, timeout_ms)
            except Exception as e:
                ok, note = False, "open/attach: %s" % e
        if not self._vote(ok):
            return False, note or "a peer could not map the inboxes"
        for _ in range(3):
            ok = self.backend.p2p_probe() and ok
        if not self._vote(ok):
            return False, "probe packets did not arrive on every rank"
        return True, "self-test passed on %d rank(s)" % self.world

    def next(self, x):
        """x: float32 tensor [s] on the backend's device (replicated on every rank). Returns u [a]
        (device tensor, valid in stream order; identical on every rank).

        Direct exchange: a packet that misses its deadline never hangs the step and never writes garbage — the
        affected columns of U get a zero update and a flag is raised (include/mppi_c.h). The flag is looked at when the
        NEXT step is enqueued (and by check()): next() then raises ExchangeTimeout on this rank. Every rank that
        depends on the late or dead peer runs into its own deadline within one more step and raises too. Controls
        returned since the failing step are zero-update controls; U and the step counter may differ between ranks
        until resync() (a collective: call it on every rank) has copied them from rank 0 — the controller then
        continues on the all-gather path."""
        if self.world == 1 and hasattr(self.backend, "step") and not self.force_exchange:
            self.backend.step(x, self.u)
            return self.u
        if self.p2p:
            try:
                self.backend.p2p_step(x, self.u)
            except Exception as e:
                if getattr(e, "status", None) == 8:  # MPPI_ERR_EXCHANGE: refused before anything was enqueued
                    raise ExchangeTimeout(str(e)) from None
                raise
            return self.u
        if self.rccl is not None:
            self.backend.shard_step(x, self.u, self.rccl.coll)
            return self.u
        collective = self.world > 1 or (self.force_exchange and dist.is_initialized())
        if self.normalize:
            self.backend.cost_range(x, self.range)  # {-min, max} of this rank's costs
            if collective:  # the global min and max in ONE all-reduce: MAX over {-min, max}
                dist.all_reduce(self.range, op=dist.ReduceOp.MAX, group=self.group)
            self.backend.partial_normalized(x, self.range, self.record)
        else:
            self.backend.partial(x, self.record)
        if collective:
            if self.records.is_cuda and dist.get_backend(self.group) != "nccl":
                # (the one-GPU rehearsal of bench.py: GPU shards, gloo rendezvous — the records take the host's all-gather)
                torch.cuda.current_stream(self.backend.device).synchronize()
                host = torch.empty(self.records.shape, dtype=torch.float32)
                dist.all_gather_into_tensor(host, self.record.cpu(), group=self.group)
                self.records.copy_(host)
            else:
                dist.all_gather_into_tensor(self.records, self.record, group=self.group)
            self.backend.finish(self.records, self.world, self.u)
        else:
            self.backend.finish(self.record, 1, self.u)
        return self.u

    def next_checked(self, x):
        """next(x) for a host loop that ACTUATES u at once: waits for the step, looks at the deadline flag and raises
        ExchangeTimeout BEFORE returning, so a zero-update control is never handed out (next() alone reports a missed
        deadline one step late by design: it only enqueues). Costs one stream synchronisation per step."""
        u = self.next(x)
        dev = self.backend.device
        if getattr(dev, "type", "cpu") == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        self.check()
        return u

    def check(self):
        """After synchronising the stream: raise if a direct-exchange spin ever hit its deadline (the controls since
        then are zero-update controls; see next())."""
        if self.p2p and self.backend.p2p_timed_out():
            raise ExchangeTimeout("direct exchange: a packet did not arrive before the deadline")

    def resync(self):
        """Collective (every rank): leave the direct exchange for good, copy rank 0's nominal sequence and Philox
        step counter to every rank, continue on the all-gather path. Call it on all ranks after ExchangeTimeout."""
        self.p2p, self.exchange = False, "rccl"
        self.p2p_note = "closed after a missed deadline"
        if self.rccl is None and os.environ.get("MPPI_RCCL_CALL", "c") != "torch":
            self.rccl, self.rccl_note = self._bring_up_rccl()
        U = self.backend.action_sequence()
        step = torch.tensor([self.backend.step_counter()], dtype=torch.int64)
        if dist.is_initialized() and self.world > 1:
            on_gpu = dist.get_backend(self.group) == "nccl"
            dev = self.backend.device if on_gpu else "cpu"
            U, step = U.to(dev), step.to(dev)
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast(U, src=src, group=self.group)
            dist.broadcast(step, src=src, group=self.group)
        self.backend.set_state(U.cpu(), int(step.item()))


def shard_bounds(k, rank, world):
    """[lo, hi) of the samples rank owns — the same integer arithmetic as mppi_create."""
    return rank * k // world, (rank + 1) * k // world
