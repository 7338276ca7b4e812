#!/bin/bash
# PMC comparison of alternative builds of the library: tools/pmc_libs.sh <tag> <spp> <variant> <lib.so|-> [...]
# (counters only, own passes; RTMI_LIB is exported before rocprofv3 starts: nothing re-execs behind it)
set -o pipefail
TAG=$1; SPP=$2; VAR=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  N=$(basename $L .so)
  if [ "$L" = "-" ]; then unset RTMI_LIB; N=intree; else export RTMI_LIB=$R/$L; fi
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/${N}_a -- python3 $R/tools/gpu_sweep.py $SPP 0 $VAR > $OUT/${N}_a.log 2>&1 || { echo fail a $N; tail -3 $OUT/${N}_a.log; }
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $OUT/${N}_b -- python3 $R/tools/gpu_sweep.py $SPP 0 $VAR > $OUT/${N}_b.log 2>&1 || { echo fail b $N; tail -3 $OUT/${N}_b.log; }
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, os
out = sys.argv[1]
for d in sorted(set(os.path.basename(p)[:-2] for p in glob.glob(out + "/*_a"))):
    agg = collections.defaultdict(float)
    for part in "ab":
        for f in glob.glob(f"{out}/{d}_{part}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if 'render_kernel' in r['Kernel_Name']:
                    agg[r['Counter_Name']] += float(r['Counter_Value'])
    if not agg: continue
    v = agg
    line = {k: f"{x:.4g}" for k, x in sorted(agg.items())}
    util = v['SQ_THREAD_CYCLES_VALU'] / (64 * v['SQ_ACTIVE_INST_VALU']) if v.get('SQ_ACTIVE_INST_VALU') else 0
    print(d, line, f"lane_util={util:.3f}")
PY
