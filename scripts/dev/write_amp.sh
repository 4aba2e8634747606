#!/bin/bash
# development aid (GPU box): scripts/dev/write_amp.sh -- WRITE_SIZE of the one-pass task kernel per steady launch for the release library and
# the two measurement builds of SPADA_WA_PROBE (wrong results): 1 = the tasks of the older range path not run, 2 = no chain (no status word
# stored or read, task t stores at t * 1500).  Build the variants first:  scripts/build_variant.sh wa1 -DSPADA_WA_PROBE=1 ; ... wa2 -DSPADA_WA_PROBE=2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in spgemm wa1 wa2; do
  OUT=$REPO/gpurun_out/r06/wa_$v
  rm -rf $OUT; mkdir -p $OUT
  SPADA_LIB_PATH=$REPO/spada_sim_amd/lib/libspada_$v.so timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o p -- python3 $REPO/scripts/dev/loop_fused.py webbase 8 > $OUT/log.txt 2>&1
  python3 - <<PY
import csv, statistics
rows = [r for r in csv.DictReader(open("$OUT/p_counter_collection.csv")) if "k_task<2" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
per = {}
for r in rows: per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
v = sorted(per.values())
print("$v: k_task<2> WRITE_SIZE per launch, median of %d launches: %.1f MB" % (len(v), statistics.median(v) * 1024 / 1e6))
PY
done
