"""Worker of test_direct_exchange_across_processes_over_hipipc: every rank drives its shard on cuda:0 (one GPU box),
rendezvous over gloo. Checks the direct exchange against handles of the all-gather path held in the same process."""
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
    K, H, a = 4096, 32, 3
    cfg = dict(k=K, tau=H, s_dim=6, a_dim=a, dt=0.1, mass=1.0, lam=1.0, sigma=(0.5 * np.eye(a)).astype(np.float32),
               goal=np.array([1, 0, 0.5, 0, 0.75, 0], np.float32), seed=5)
    ctl = ShardedController(device_index=0, exchange="p2p", **cfg)
    assert ctl.exchange == "p2p", ctl.p2p_note
    # the same step through partial -> (in-process) gather -> finish, all shards held here
    ref = [m.Handle(shard_rank=g, shard_count=world, **cfg) for g in range(world)]
    n = ref[0].record_size
    recs = torch.zeros(world * n, device="cuda")
    u_ref = torch.zeros(a, device="cuda")
    x = torch.tensor([0.2, 0.1, -0.3, 0, 0.5, -0.1], device="cuda")
    for step in range(5):
        u = ctl.next(x)
        torch.cuda.synchronize()
        ctl.check()
        for g, h in enumerate(ref):
            h.shard_partial(x.data_ptr(), recs[g * n:(g + 1) * n].data_ptr())
            h.synchronize()
        for h in ref:
            h.shard_finish(recs.data_ptr(), world, u_ref.data_ptr())
            h.synchronize()
        np.testing.assert_array_equal(u.cpu().numpy(), u_ref.cpu().numpy())
        np.testing.assert_array_equal(ctl.backend.action_sequence().numpy(), ref[0].get_action_sequence())
    dist.barrier()
    print("P2P_WORKER_OK rank %d: %s" % (rank, ctl.p2p_note), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
