#!/bin/bash
# development aid (GPU box): scripts/dev/trace_bench.sh <bench.py arguments> -- per-kernel durations (rocprofv3 --kernel-trace) of a bench run, steady launches
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05/trace_bench
rm -rf $OUT; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/bench.py --no-cpu-baseline --steps 6 --warmup 3 "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/t_kernel_trace.csv")):
    n = r["Kernel_Name"].split("(")[0].replace("spada::", "").replace("void ", "")
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -statistics.median(kv[1])):
    if n.startswith("k_"): print("%-34s launches %3d  median %.1f us  max %.1f" % (n[:34], len(v), statistics.median(v), max(v)))
PY
tail -1 $OUT/log.txt | cut -c1-200
