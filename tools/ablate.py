#!/usr/bin/env python3
"""Build timing-only ablation variants of the tile kernel (CPU box), to be timed on the GPU box with
tools/time_step.py. Outputs are wrong by construction; only kernel time matters."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mppi_tf_amd  # noqa
from mppi_tf_amd import build

VARIANTS = {
    "full": [],
    "philox_block_major": ["MPPI_PHILOX_BLOCK_MAJOR"],  # r03's Philox order against r04's round-major one (same results)
    "no_philox": ["MPPI_ABLATE_PHILOX"],
    "no_rollout": ["MPPI_ABLATE_ROLLOUT"],
    "no_wsum": ["MPPI_ABLATE_WSUM"],
    "only_philox": ["MPPI_ABLATE_ROLLOUT", "MPPI_ABLATE_WSUM"],
    "only_rollout": ["MPPI_ABLATE_PHILOX", "MPPI_ABLATE_WSUM"],
    "nothing": ["MPPI_ABLATE_PHILOX", "MPPI_ABLATE_ROLLOUT", "MPPI_ABLATE_WSUM"],
    # k_finish_cols stopped after stage n (where do its ~4.3 us go): 0 entry, 1 record loads, 2 min over the records, (full = all)
    # the two-wave pipelines (k_rollout_nnspeed_pc): which wave a step waits for
    "pc_no_network": ["MPPI_PC_ABL=1"],
    "pc_no_pose": ["MPPI_PC_ABL=2"],
    "pc_neither": ["MPPI_PC_ABL=3"],
    "kind_per_step": ["MPPI_PC_KIND_PER_STEP"],  # k_rollout_pc: the action-cost form tested in every step (r04) instead of once around the producers' loop (r05; tools/ab_kind.py)
    "consumer_boost": ["MPPI_PC_CONSUMER_BOOST=1"],  # k_rollout_pc: the consumer wave one priority level above its progress level
    "finish_s0": ["MPPI_FINISH_STAGE=0"],
    "finish_s1": ["MPPI_FINISH_STAGE=1"],
    "finish_s2": ["MPPI_FINISH_STAGE=2"],
}
if __name__ == "__main__":
    for name, defs in VARIANTS.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        print(build.build_variant(name, defs))
