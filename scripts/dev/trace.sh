#!/bin/bash
# development aid (GPU box): scripts/dev/trace.sh "<probe workloads>"  -- rocprofv3 kernel trace of scripts/probe_tasks.py, per-kernel median / min
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
for w in $1; do
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace/$w
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace/$w -o kt -- python3 $GRAFT_REPO_ROOT/scripts/probe_tasks.py $w > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/trace/$w/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("spada::", "").replace("void ", "")
    agg[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("== $w")
for n, v in sorted(agg.items(), key=lambda x: -sum(x[1])):
    v2 = sorted(v)
    print(f"{n[:44]:44s} n={len(v):4d} median {v2[len(v2)//2]:9.1f} us  min {v2[0]:9.1f}")
PY
done
