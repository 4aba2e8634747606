#!/bin/bash
# development aid (GPU box): per-stage clock ticks of the batch task (libspada_dbg.so) + the kernel trace of the release library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg1
OUT=$PWD/gpurun_out/dbg1/out.txt
: > $OUT
SPADA_LIB_PATH=$PWD/spada_sim_amd/lib/libspada_dbg.so timeout 300 python scripts/probe_tasks.py $1 >> $OUT 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/dbg1/kt -o kt -- python3 $GRAFT_REPO_ROOT/scripts/probe_tasks.py webbase > /dev/null 2>&1
python3 - <<PY >> $OUT
import csv, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/dbg1/kt/**/*kernel_trace.csv", recursive=True)[0]
seen = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0][:60]
    if "k_task" in n and n not in seen:
        seen[n] = r
        print(n, {k: r[k] for k in r if any(s in k for s in ("VGPR", "SGPR", "Segment", "Workgroup_Size", "Grid_Size", "Scratch", "LDS"))})
PY
cat $OUT
