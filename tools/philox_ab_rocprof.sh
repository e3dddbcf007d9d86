#!/bin/bash
# Run ON THE GPU BOX: the headline kernel under rocprofv3 --kernel-trace --stats on ONE box, r04's round-major Philox (the shipped library) against r03's
# block-major order (build/variants/libmppi_hip_philox_block_major.so, tools/ablate.py philox_block_major), three alternating pairs -> gpurun_out/philox_ab_rocprof.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/philox_ab_rocprof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $R/bench.py --workload pm3d --steps 200 --warmup 20 --no-cpu-baseline --no-subrecords --min-time 0.2"
for i in 1 2 3; do
  unset MPPI_SO_PATH
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/new$i -- $BENCH > $OUT/new$i.log 2>&1
  export MPPI_SO_PATH=$R/build/variants/libmppi_hip_philox_block_major.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/old$i -- $BENCH > $OUT/old$i.log 2>&1
done
unset MPPI_SO_PATH
cd $R
for d in new1 old1 new2 old2 new3 old3; do
  f=$(ls $OUT/$d/*/*kernel_stats.csv | head -1)
  echo "$d $(grep k_rollout_pc $f | head -1 | awk -F, '{print "calls " $2 "  avg_ns " $4 "  min_ns " $6}')"
done | tee $R/gpurun_out/philox_ab_rocprof.txt
