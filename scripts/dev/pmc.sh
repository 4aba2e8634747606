#!/bin/bash
# development aid (GPU box): scripts/dev/pmc.sh <probe workload> "<library variants>"  -- one rocprofv3 --pmc pass of scripts/probe_tasks.py
# per variant (spada_sim_amd/lib/libspada_<variant>.so): time and SQ instruction counts of the three modes of k_task, per launch (millions)
WL=${1:-webbase}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in ${2:-spgemm}; do
  export SPADA_LIB_PATH=$REPO/spada_sim_amd/lib/libspada_$v.so
  rm -rf $OUT/$v
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD \
      --output-format csv -d $OUT/$v -o p -- python3 $REPO/scripts/probe_tasks.py $WL > $OUT/$v.log 2>&1
  python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter); t = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/$v/p_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("spada::", "").replace("void ", "")
    if not k.startswith("k_task<"): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    if r["Counter_Name"] == "SQ_INSTS_VALU": t[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(agg):
    d = {c: agg[k][c] / n[k][c] / 1e6 for c in agg[k]}
    us = sorted(t[k])[len(t[k]) // 2]
    print(f"$v $WL {k:18s} us {us:7.1f} valu {d.get('SQ_INSTS_VALU',0):6.1f} salu {d.get('SQ_INSTS_SALU',0):6.1f} lds {d.get('SQ_INSTS_LDS',0):5.1f} vmem_rd {d.get('SQ_INSTS_VMEM_RD',0):4.1f} "
          f"valu_busy {d.get('SQ_ACTIVE_INST_VALU',0):6.1f} wave_cycles {d.get('SQ_WAVE_CYCLES',0):7.1f} wait {d.get('SQ_WAIT_ANY',0):7.1f} lds_conflict {d.get('SQ_LDS_BANK_CONFLICT',0):5.1f}")
PY
done
