#!/bin/bash
# development aid (GPU box): scripts/dev/timeline.sh <workload> -- start / end / duration (us) of every kernel of ONE steady one-pass step (rocprofv3 kernel trace)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-webbase-1M}
OUT=$REPO/gpurun_out/r06/timeline_$WL
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline --workload $WL > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("spada::", "").replace("void ", "").split("(")[0]) for r in csv.DictReader(open(f)))
idx = [i for i, k in enumerate(ks) if k[2].startswith("k_entry_stats")]
# the last one-pass step (a k_task<2 ...> between two k_entry_stats); bench.py ends with steps of the two-phase contract
pairs = [(idx[j], idx[j + 1]) for j in range(len(idx) - 1) if any(k[2].startswith("k_task<2") for k in ks[idx[j]:idx[j + 1]])]
i0, i1 = pairs[len(pairs) // 2]   # (a step of the timed loop: the last ones are first calls of fresh contexts)
t0 = ks[i0][0]
for s, e, n in ks[i0:i1]:
    print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f}  {n[:60]}")
PY
