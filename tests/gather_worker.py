"""Worker of test_allgather_path_across_processes_is_stream_ordered: every rank drives its shard on cuda:0 (one GPU box), rendezvous
and the record all-gather over gloo — ShardedController's torch path (three calls + one collective per step) on torch's DEFAULT
stream. r04 found that path one step stale: torch's default stream has the handle 0, which the C-ABI reads as "the handle's own
(non-blocking) stream", so the record kernel and the collective that reads its output were unordered. Each step is compared with
the same step through partial -> in-process gather -> finish on handles held here, and a SECOND controller's first step with the
first controller's first step (a stale record would leak the previous controller's last one into it)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mppi_tf_amd as m  # noqa: E402
from mppi_tf_amd.distributed import ShardedController  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    K, H, a = 65536, 64, 3  # long enough kernels that an unordered read of the record WOULD be early
    cfg = dict(k=K, tau=H, s_dim=6, a_dim=a, dt=0.1, mass=1.0, lam=1.0, sigma=(0.25 * np.eye(a)).astype(np.float32),
               goal=np.array([1, 0, 0.5, 0, 0.75, 0], np.float32), seed=5)
    x = torch.tensor([0.2, 0.1, -0.3, 0, 0.5, -0.1], device="cuda")
    first = None
    for rep in range(2):
        ctl = ShardedController(device_index=0, exchange="rccl", **cfg)
        assert ctl.exchange == "rccl" and ctl.rccl is None  # gloo job: the torch path
        ref = [m.Handle(shard_rank=g, shard_count=world, **cfg) for g in range(world)]
        n = ref[0].record_size
        recs = torch.zeros(world * n, device="cuda")
        u_ref = torch.zeros(a, device="cuda")
        for step in range(4):
            u = ctl.next(x)
            torch.cuda.synchronize()
            for g, h in enumerate(ref):
                h.shard_partial(x.data_ptr(), recs[g * n:(g + 1) * n].data_ptr())
                h.synchronize()
            for h in ref:
                h.shard_finish(recs.data_ptr(), world, u_ref.data_ptr())
                h.synchronize()
            np.testing.assert_array_equal(u.cpu().numpy(), u_ref.cpu().numpy())
            if step == 0:
                if first is None:
                    first = u.cpu().numpy().copy()
                else:
                    np.testing.assert_array_equal(u.cpu().numpy(), first)
        del ctl, ref
    dist.barrier()
    print("GATHER_WORKER_OK rank %d" % rank, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
