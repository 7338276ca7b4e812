#!/bin/bash
# PMC comparison of kernel variants: tools/pmc_ab.sh <tag> <spp> <variants "0 1"> [chunk]
set -o pipefail
TAG=$1; SPP=$2; VARS=$3; CH=${4:-0}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in $VARS; do
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/v${V}_a -- python3 $R/tools/gpu_sweep.py $SPP $CH $V > $OUT/v${V}_a.log 2>&1 || { echo fail a $V; tail -3 $OUT/v${V}_a.log; }
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/v${V}_b -- python3 $R/tools/gpu_sweep.py $SPP $CH $V > $OUT/v${V}_b.log 2>&1 || { echo fail b $V; tail -3 $OUT/v${V}_b.log; }
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/v${V}_c -- python3 $R/tools/gpu_sweep.py $SPP $CH $V > $OUT/v${V}_c.log 2>&1 || { echo fail c $V; tail -3 $OUT/v${V}_c.log; }
done
python3 - <<PY
import csv,glob,collections
for V in "$VARS".split():
    agg=collections.defaultdict(float); n=0
    for part in "abc":
        for f in glob.glob(f"$OUT/v{V}_{part}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if 'render_kernel' in r['Kernel_Name'] and int(r['Grid_Size'])>1000000:
                    agg[r['Counter_Name']]+=float(r['Counter_Value'])
    # two timed launches + none small: normalise per launch (2 reps)
    print("variant",V,{k:f"{v:.4g}" for k,v in sorted(agg.items())})
PY
