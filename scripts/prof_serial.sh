#!/bin/bash
# usage (GPU box): scripts/prof_serial.sh <tag> <workload>  -> per-bin table + serialised per-kernel times
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SPADA_SERIAL_BINS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $REPO/scripts/bin_profile.py "$@" > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
grep -v "^\[" $OUT/run.log | grep -v rocprofv3 | tail -40
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/p_kernel_stats.csv")))
for r in rows[:30]:
    n=r["Name"].replace("spada::","").split("(")[0].replace("void ","")
    print(f'{n:28s} calls {int(r["Calls"]):4d} avg {float(r["AverageNs"])/1e3:9.1f} us  {float(r["Percentage"]):5.1f}%')
PY
