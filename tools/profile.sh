#!/bin/bash
# rocprofv3 passes for the bench workload (run on the GPU box through gpurun).
# usage: tools/profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (--no-extra: the other BASELINE configurations launch the SAME kernel on other workloads and would mix into its average)
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extra $@"
# 1) kernel trace + stats (same command as the bench line)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
# 2) PMC passes (own runs, counters only), shorter workload
PARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-counts --no-extra $@"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1 -- python3 $R/bench.py $PARGS > $OUT/pmc_sq1.log 2>&1 || { echo pmc1 failed; tail -5 $OUT/pmc_sq1.log; exit 1; }
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py $PARGS > $OUT/pmc_sq2.log 2>&1 || { echo pmc2 failed; tail -5 $OUT/pmc_sq2.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $PARGS > $OUT/pmc_fetch.log 2>&1 || { echo fetch failed; tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $PARGS > $OUT/pmc_write.log 2>&1 || { echo write failed; tail -5 $OUT/pmc_write.log; exit 1; }
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- python3 $R/bench.py $PARGS > $OUT/pmc_grbm.log 2>&1 || echo grbm failed
find $OUT -name '*.csv' | head -40
