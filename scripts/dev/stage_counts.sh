#!/bin/bash
# development aid (GPU box): SQ instruction counts and time of k_task<NUMERIC> for the early-stop variants libspada_stopK.so
# (scripts/build_variant.sh stopK -DSPADA_BT_STOP=K): the difference between consecutive cut points is what a stage of the batch
# task costs.   usage: scripts/dev/stage_counts.sh <probe workload name> <out dir> [variants ...]
WL=${1:-webbase}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/${2:-stages}
shift 2
VARS=${*:-stop10 stop1 stop11 stop12 stop13 stop2 stop14 stop3 stop15 stop16 stop17 stop18 stop19 stop5 stop20 stop6 stop21 spgemm}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  export SPADA_LIB_PATH=$REPO/spada_sim_amd/lib/libspada_$v.so
  [ -f $SPADA_LIB_PATH ] || continue
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD \
      --output-format csv -d $OUT/$v -o p -- python3 $REPO/scripts/probe_tasks.py $WL > $OUT/$v.log 2>&1
done
python3 - <<PY
import csv, collections, os
prev = None
for v in "$VARS".split():
    p = "$OUT/%s/p_counter_collection.csv" % v
    if not os.path.exists(p): continue
    agg = collections.defaultdict(float); n = collections.Counter(); t = []
    for r in csv.DictReader(open(p)):
        if "k_task<1" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            if r["Counter_Name"] == "SQ_INSTS_VALU": t.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    d = {k: agg[k] / n[k] / 1e6 for k in agg}
    us = sorted(t)[len(t) // 2] if t else 0
    line = f"{v:8s} us {us:7.1f} valu {d.get('SQ_INSTS_VALU',0):6.1f} salu {d.get('SQ_INSTS_SALU',0):6.1f} lds {d.get('SQ_INSTS_LDS',0):5.1f} vmem_rd {d.get('SQ_INSTS_VMEM_RD',0):4.1f}"
    if prev: line += f"   | delta us {us - prev[0]:6.1f} valu {d.get('SQ_INSTS_VALU',0) - prev[1]:6.1f} lds {d.get('SQ_INSTS_LDS',0) - prev[2]:5.1f}"
    prev = (us, d.get('SQ_INSTS_VALU',0), d.get('SQ_INSTS_LDS',0))
    print(line)
PY
