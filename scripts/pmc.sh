#!/bin/bash
# usage (GPU box): scripts/pmc.sh <tag> "<counters>" <workload...>  -> per-kernel counter sums (one pass)
set -e
TAG=$1; shift
CNT=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SPADA_SERIAL_BINS=1
rocprofv3 --pmc $CNT --output-format csv -d $OUT -o p -- python3 $REPO/scripts/perf_probe.py "$@" > $OUT/run.log 2>&1 || { tail -20 $OUT/run.log; exit 1; }
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in rows:
    n = r["Kernel_Name"].replace("spada::", "").split("(")[0].replace("void ", "")
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == rows[0]["Counter_Name"]: calls[n] += 1
names = sorted({r["Counter_Name"] for r in rows})
print("kernel".ljust(44), "calls", " ".join(x.rjust(22) for x in names))
for n in sorted(agg, key=lambda k: -max(agg[k].values())):
    print(n[:44].ljust(44), f"{calls[n]:5d}", " ".join(f"{agg[n][x] / max(calls[n], 1):22.0f}" for x in names))
PY
