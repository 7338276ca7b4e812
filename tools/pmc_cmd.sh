#!/bin/bash
# PMC passes (counters only) over any tools/*.py command: tools/pmc_cmd.sh <tag> <script.py> [args...]
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/a -- python3 $R/tools/$1 "${@:2}" > $OUT/a.log 2>&1 || { echo fail a; tail -3 $OUT/a.log; }
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $OUT/b -- python3 $R/tools/$1 "${@:2}" > $OUT/b.log 2>&1 || { echo fail b; tail -3 $OUT/b.log; }
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/c -- python3 $R/tools/$1 "${@:2}" > $OUT/c.log 2>&1 || { echo fail c; tail -3 $OUT/c.log; }
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for part in "abc":
    for f in glob.glob(f"{out}/{part}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if 'render_kernel' in r['Kernel_Name']:
                agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
v = {k: agg[k] / max(1, n[k]) for k in agg}
print({k: f"{x:.4g}" for k, x in sorted(v.items())})
if v.get('SQ_ACTIVE_INST_VALU'):
    print("lane_util %.3f" % (v['SQ_THREAD_CYCLES_VALU'] / (64 * v['SQ_ACTIVE_INST_VALU'])))
if v.get('GRBM_GUI_ACTIVE') and v.get('SQ_INSTS_VALU'):
    cyc = v['GRBM_GUI_ACTIVE'] / 8
    print("kernel cycles %.4g, valu_busy %.3f, waves/SIMD %.2f" % (cyc, v['SQ_INSTS_VALU'] / 1024 * 2 / cyc, v.get('SQ_WAVE_CYCLES', 0) * 4 / (1024 * cyc)))
PY
