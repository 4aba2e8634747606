#!/bin/bash
# development aid (GPU box): scripts/dev/trace_chunk.sh <chunk> ... -- kernel durations (rocprofv3 --kernel-trace) of the symbolic + numeric calls on chunks of R-MAT 22
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05/trace_chunk
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/scripts/probe_chunks.py 22 69 "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/t_kernel_trace.csv")):
    n = r["Kernel_Name"].split("(")[0].replace("spada::", "").replace("void ", "")
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -max(kv[1])):
    if max(v) > 20: print("%-34s launches %3d  us per launch: %s" % (n[:34], len(v), " ".join("%.0f" % x for x in v if x > 5)))
PY
