#!/bin/bash
# Instruction-fetch and instruction-mix counters of the render kernel (own PMC passes, counters only): tools/pmc_ifetch.sh <tag> [spp]
set -o pipefail
TAG=${1:-if}; SPP=${2:-256}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmci_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $R/tools/gpu_ab.py child $SPP 0 > $OUT/a.log 2>&1 || { echo fail a; tail -3 $OUT/a.log; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS --output-format csv -d $OUT/b -- python3 $R/tools/gpu_ab.py child $SPP 0 > $OUT/b.log 2>&1 || { echo fail b; tail -3 $OUT/b.log; }
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/c -- python3 $R/tools/gpu_ab.py child $SPP 0 > $OUT/c.log 2>&1 || { echo fail c; tail -3 $OUT/c.log; }
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(float); n = collections.defaultdict(int)
for part in "abc":
    for f in glob.glob(f"{out}/{part}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if 'render_kernel' in r['Kernel_Name']:
                agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(agg):
    print(f"{k}: {agg[k] / max(1, n[k]):.5g} (per launch, {n[k]} launches)")
PY
