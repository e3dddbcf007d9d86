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
    "no_philox": ["MPPI_ABLATE_PHILOX"],
    "no_rollout": ["MPPI_ABLATE_ROLLOUT"],
    "no_wsum": ["MPPI_ABLATE_WSUM"],
    "only_philox": ["MPPI_ABLATE_ROLLOUT", "MPPI_ABLATE_WSUM"],
    "only_rollout": ["MPPI_ABLATE_PHILOX", "MPPI_ABLATE_WSUM"],
    "nothing": ["MPPI_ABLATE_PHILOX", "MPPI_ABLATE_ROLLOUT", "MPPI_ABLATE_WSUM"],
    "nokeep": ["MPPI_PC_NO_KEEP"],
}
if __name__ == "__main__":
    for name, defs in VARIANTS.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        print(build.build_variant(name, defs))
