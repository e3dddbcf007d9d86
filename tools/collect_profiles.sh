#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects the rocprofv3 evidence for bench.py —
# kernel-trace stats, and PMC counters in SEPARATE passes (never combined with sys/hip traces):
# SQ occupancy/issue counters, then FETCH_SIZE, then WRITE_SIZE. Results land in gpurun_out/profiles_<tag>/;
# tools/summarize_profiles.py turns them into the files committed under profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
rm -rf $OUT   # never mix passes of different runs (gpurun merges gpurun_out/ back over what is already there: clear the local copy too)
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
WORKLOAD=${2:-pm3d}
STEPS=200; [ "$WORKLOAD" = "mlp" ] && STEPS=10
BENCH="python3 $R/bench.py --workload $WORKLOAD --steps $STEPS --warmup 3 --no-cpu-baseline --no-subrecords --min-time 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
# VALU instructions by class (for the issue-rate floor, roofline.valu) and the busy cycles of the issue ports: own passes
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/pmc_sq3 -- $BENCH > $OUT/pmc_sq3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_grbm -- $BENCH > $OUT/pmc_grbm.log 2>&1
if [ "$WORKLOAD" = "mlp" ]; then  # matrix-core occupancy of the MLP kernels (own pass; a refused counter only loses this pass)
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/pmc_mfma -- $BENCH > $OUT/pmc_mfma.log 2>&1
fi
cd $R
grep -h '"metric"' $OUT/*.log | head -1 > $OUT/bench_under_profiler.json
ls -R $OUT | head -40
