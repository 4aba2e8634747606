#!/bin/bash
# development aid (GPU box): scripts/dev/trace_chunk.sh "<library variants>" <chunk>  -- rocprofv3 kernel trace of one chunk of R-MAT 22
# (scripts/probe_chunks.py 22 69 <chunk>) per variant library: per-kernel totals
REPO=${GRAFT_REPO_ROOT:-/root/repo}
CH=${2:-34}
mkdir -p $REPO/gpurun_out/trace; cd /tmp && export TMPDIR=/tmp
for v in $1; do
export SPADA_LIB_PATH=$REPO/spada_sim_amd/lib/libspada_$v.so
rm -rf $REPO/gpurun_out/trace/chunk_$v
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/trace/chunk_$v -o kt -- python3 $REPO/scripts/probe_chunks.py 22 69 $CH > $REPO/gpurun_out/trace/chunk_$v.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$REPO/gpurun_out/trace/chunk_$v/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("spada::", "").replace("void ", "")
    agg[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("== $v chunk $CH")
for n, v in sorted(agg.items(), key=lambda x: -sum(x[1]))[:16]:
    v2 = sorted(v)
    print(f"{n[:44]:44s} n={len(v):4d} median {v2[len(v2)//2]:9.1f} us  min {v2[0]:9.1f}  total {sum(v):10.1f}")
PY
done
