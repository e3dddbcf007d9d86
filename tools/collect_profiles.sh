#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects the rocprofv3 evidence behind bench.py's roofline fields, for every
# kernel the bench line quotes —
#   tools/collect_profiles.sh TAG WORKLOAD[:bx3|:fma] [...]        e.g.  r05 pm3d pm3d:fma pm2d mlp mlp:bx3 mlp32 mlp32:bx3 nnauv nnauv:bx3 auv nnspeed
# per workload: --kernel-trace --stats, then PMC counters in SEPARATE passes (never combined with sys/hip/marker traces): the VALU
# instruction classes, the busy cycles of the issue ports, the matrix pipe, GRBM_GUI_ACTIVE, and (point mass / 2x256 MLP) FETCH_SIZE and
# WRITE_SIZE. Results land in gpurun_out/profiles_<tag>/<workload>/; tools/summarize_profiles.py turns them into the files under profiles/.
set -u
TAG=${1:-r04}
shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
[ -z "${MPPI_PROFILES_APPEND:-}" ] && rm -rf $OUT   # never mix passes of different runs (gpurun merges gpurun_out/ back over what is already there: clear the local copy too); MPPI_PROFILES_APPEND=1: a second call of the same collection
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for SPEC in "$@"; do
  WL=${SPEC%%:*}
  EXTRA=""; [ "$SPEC" != "$WL" ] && EXTRA="--bf16x3"
  [ "$SPEC" = "$WL:fma" ] && EXTRA="--fp-contract"   # the contracted instance of the point-mass rollout (MPPI_FLAG_FP_CONTRACT)
  [ "$SPEC" = "$WL:k4096" ] && EXTRA="--samples 4096"
  NAME=${SPEC/:/_}
  D=$OUT/$NAME
  mkdir -p $D
  STEPS=100
  case $WL in mlp) STEPS=10;; mlp32|nnauv|auv|nnspeed) STEPS=30;; esac
  BENCH="python3 $R/bench.py --workload $WL $EXTRA --steps $STEPS --warmup 3 --no-cpu-baseline --no-subrecords --no-prelaunched --min-time 0"
  echo "== $NAME: $BENCH"
  # the stats pass times the kernel in steady state: batches repeated for 0.3 s, as bench.py's timed region does (a 100-step run alone is over
  # before the clocks have ramped: 18.1 us for the headline kernel against 15.7-16.0 us here, profiles/r04_philox_ab_rocprofv3.txt)
  STATS_BENCH=${BENCH/--min-time 0/--min-time 0.3}
  # (the stats pass keeps the opt-in pre-launched pipeline's figure: under --kernel-trace it forms — k_step_pc<.., 4 | 5> then appear with a duration that INCLUDES their wait
  # for U'; the counter passes serialise dispatches, the pipeline cannot form there and is left out: --no-prelaunched)
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- ${STATS_BENCH/--no-prelaunched /} > $D/stats.log 2>&1
  find $D/stats -name '*kernel_trace.csv' -delete   # tens of thousands of dispatch rows: only the stats summary travels back (gpurun merges <= 64 MiB)
  # VALU instructions by class (the issue-rate floor) and the busy cycles of the issue ports: own passes
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $D/pmc_sq2 -- $BENCH > $D/pmc_sq2.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $D/pmc_sq3 -- $BENCH > $D/pmc_sq3.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES --kernel-trace --output-format csv -d $D/pmc_sq -- $BENCH > $D/pmc_sq.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D/pmc_grbm -- $BENCH > $D/pmc_grbm.log 2>&1
  case $WL in pm3d|pm2d|pm1d|auv) ;; *)  # matrix-core occupancy of the learned-model kernels (a refused counter only loses this pass)
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --kernel-trace --output-format csv -d $D/pmc_mfma -- $BENCH > $D/pmc_mfma.log 2>&1;;
  esac
  case $WL in pm3d|mlp)
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/pmc_fetch -- $BENCH > $D/pmc_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/pmc_write -- $BENCH > $D/pmc_write.log 2>&1;;
  esac
  grep -h '"metric"' $D/*.log | head -1 > $D/bench_under_profiler.json
  python3 $R/tools/slim_counters.py $D   # only what summarize_profiles.py reads travels back (gpurun merges <= 64 MiB)
done
cd $R
ls $OUT
