#!/bin/bash
# GPU box: SQ instruction counts of the NUMERIC task kernel cut short behind successive stages of the batch task (libspada_stop<k>.so built by
# `scripts/dev/ablate.sh` with masks 256 512 768 1280 1536 -> stop 1 2 3 5 6) next to the full kernel; one --pmc pass each
cd "$(dirname "$0")/../.."
REPO=$PWD
OUT=$REPO/gpurun_out/r05/stages
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for l in $REPO/spada_sim_amd/lib/libspada_abl256.so $REPO/spada_sim_amd/lib/libspada_abl512.so $REPO/spada_sim_amd/lib/libspada_abl768.so $REPO/spada_sim_amd/lib/libspada_abl1280.so $REPO/spada_sim_amd/lib/libspada_abl1536.so $REPO/spada_sim_amd/lib/libspada_spgemm.so; do
  n=$(basename $l .so)
  SPADA_LIB_PATH=$l rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/$n -o p -- python3 $REPO/scripts/dev/run_two_phase.py webbase > $OUT/$n.log 2>&1
  python3 - <<PY
import csv, collections
agg = collections.defaultdict(float); dur = {}
for r in csv.DictReader(open("$OUT/$n/p_counter_collection.csv")):
    if "k_task<1" not in r["Kernel_Name"]: continue
    agg[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
ids = sorted(dur, key=lambda d: dur[d])
d = ids[len(ids) // 2] if ids else None
print("$n", {c: round(v / 1e6, 1) for (i, c), v in sorted(agg.items()) if i == d}, "us", dur.get(d, 0) / 1e3, "launches", len(ids))
PY
done
