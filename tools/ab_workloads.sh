#!/bin/bash
# Same-box A/B of bench workloads between the shipped library and variant builds:  tools/ab_workloads.sh "<workload ...>" <variant .so | default> ...
# prints ms per step and the rollout kernel's own time for every (workload, library), twice, alternating.
wls="$1"; shift
for rep in 1 2; do for wl in $wls; do for so in "$@"; do
  if [ "$so" = default ]; then unset MPPI_SO_PATH; else export MPPI_SO_PATH="$so"; fi
  timeout -k 10 120 python bench.py --workload $wl --no-cpu-baseline --no-subrecords --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-8s %-50s step %.4f ms  kernel %.1f us' % ('$wl', '$so', d['ms_per_step'], d['roofline']['kernel_us']))" || exit 1
done; done; done
