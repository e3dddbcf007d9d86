"""Worker of test_direct_exchange_with_eight_shards_in_one_process (needs GPU_MAX_HW_QUEUES >= 9 in its environment BEFORE the
HIP runtime comes up, hence its own process): 8 shards of one controller on cuda:0, each handle on its own stream, exchange
their records inside k_finish_cols_xchg; compared bitwise with the partial -> gather -> finish path."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mppi_tf_amd as m  # noqa: E402

G = 8


def run(K, H, a, steps):
    cfg = dict(k=K, tau=H, s_dim=2 * a, a_dim=a, dt=0.1, mass=1.0, lam=1.0, sigma=(0.5 * np.eye(a)).astype(np.float32),
               goal=np.array([1, 0, 0.5, 0, 0.75, 0], np.float32)[:2 * a], seed=5)
    ref = [m.Handle(shard_rank=g, shard_count=G, **cfg) for g in range(G)]
    hs = [m.Handle(shard_rank=g, shard_count=G, **cfg) for g in range(G)]
    ptrs = [h.p2p_export(want_ipc=False)[0] for h in hs]
    for h in hs:
        h.p2p_attach(ptrs, timeout_ms=2000)
    with ThreadPoolExecutor(G) as ex:  # a probe synchronises its stream: all shards must be in flight together
        for _ in range(3):
            assert all(ex.map(lambda h: h.p2p_probe(), hs)), "probe packets missing"
    n = ref[0].record_size
    x = torch.tensor([0.2, 0.1, -0.3, 0, 0.5, -0.1][:2 * a], device="cuda")
    recs = torch.zeros(G * n, device="cuda")
    u_ref = torch.zeros(a, device="cuda")
    us = [torch.zeros(a, device="cuda") for _ in range(G)]
    for step in range(steps):
        for g, h in enumerate(ref):
            h.shard_partial(x.data_ptr(), recs[g * n:(g + 1) * n].data_ptr())
            h.synchronize()
        for h in ref:
            h.shard_finish(recs.data_ptr(), G, u_ref.data_ptr())
            h.synchronize()
        for g, h in enumerate(hs):  # enqueue only: the kernels of all 8 shards meet on the GPU
            h.p2p_step(x.data_ptr(), us[g].data_ptr())
        for h in hs:
            h.synchronize()
            assert not h.p2p_timed_out(), "a packet missed its deadline at step %d" % step
        for g, h in enumerate(hs):
            np.testing.assert_array_equal(us[g].cpu().numpy(), u_ref.cpu().numpy())
            np.testing.assert_array_equal(h.get_action_sequence(), ref[0].get_action_sequence())
            assert h.get_step_counter() == step + 1


def main():
    run(8 * 4096, 32, 3, 4)     # 97 columns x 8 shards of finish workgroups
    run(8 * 8192, 64, 3, 3)     # configs[2]'s record (194 floats) at 8 shards
    print("P2P8_WORKER_OK", flush=True)


if __name__ == "__main__":
    main()
