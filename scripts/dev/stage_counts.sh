#!/bin/bash
# development aid (GPU box): SQ instruction counts and time of k_task<NUMERIC> for the early-stop variants libspada_stopK.so
# (scripts/build_variant.sh stopK -DSPADA_BT_STOP=K) and the full library: the difference between consecutive variants is what a
# stage of the batch task costs.   usage: scripts/dev/stage_counts.sh <probe workload name> <out dir>
WL=${1:-webbase}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${2:-stages}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in stop1 stop2 stop3 stop4 stop5 stop6 spgemm; do
  export SPADA_LIB_PATH=$REPO/spada_sim_amd/lib/libspada_$v.so
  [ -f $SPADA_LIB_PATH ] || continue
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD \
      --output-format csv -d $OUT/$v -o p -- python3 $REPO/scripts/probe_tasks.py $WL > $OUT/$v.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_t -o p -- python3 $REPO/scripts/probe_tasks.py $WL > $OUT/${v}_t.log 2>&1
done
python3 - <<PY
import csv, collections, os
for v in ("stop1","stop2","stop3","stop4","stop5","stop6","spgemm"):
    p = "$OUT/%s/p_counter_collection.csv" % v
    if not os.path.exists(p): continue
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(p)):
        if "k_task<1" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    t = None
    ps = "$OUT/%s_t/p_kernel_stats.csv" % v
    if os.path.exists(ps):
        for r in csv.DictReader(open(ps)):
            if "k_task<1" in r["Name"]: t = float(r["AverageNs"]) / 1e3
    print(v, "us", t, {k: round(agg[k] / n[k] / 1e6, 1) for k in sorted(agg)})
PY
