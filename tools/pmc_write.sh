#!/bin/bash
# HBM write bytes of the render kernel (PMC WRITE_SIZE, own pass): tools/pmc_write.sh <tag> <spp> [lib.so ...]   ('-' = in-tree)
set -o pipefail
TAG=$1; SPP=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcw_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for LIB in "$@"; do
  N=$(basename $LIB .so)
  if [ "$LIB" = "-" ]; then unset RTMI_LIB; N=intree; else export RTMI_LIB=$R/$LIB; fi
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$N -- python3 $R/tools/gpu_sweep.py $SPP 0 0 > $OUT/$N.log 2>&1 || { echo fail $N; tail -3 $OUT/$N.log; }
done
python3 - <<PY
import csv,glob,collections,os
for d in sorted(glob.glob("$OUT/*/")):
    tot=0.0;n=0
    for f in glob.glob(d+"/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if 'render_kernel' in r['Kernel_Name'] and r['Counter_Name']=='WRITE_SIZE':
                tot+=float(r['Counter_Value']); n+=1
    print(os.path.basename(d.rstrip('/')), "launches", n, "WRITE_SIZE per launch (KB units -> GB): %.3f" % (tot/max(n,1)*1024/1e9) if n else "none")
PY
