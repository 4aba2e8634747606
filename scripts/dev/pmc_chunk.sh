#!/bin/bash
# development aid (GPU box): scripts/dev/pmc_chunk.sh <chunk> -- SQ / traffic counters and durations of the BIG-row kernels on one chunk of R-MAT 22
REPO=${GRAFT_REPO_ROOT:-/root/repo}
CH=${1:-34}
OUT=$REPO/gpurun_out/r05/pmc_chunk$CH
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/sq -o p -- python3 $REPO/scripts/probe_chunks.py 22 69 $CH > $OUT/sq.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/wr -o p -- python3 $REPO/scripts/probe_chunks.py 22 69 $CH > $OUT/wr.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/rd -o p -- python3 $REPO/scripts/probe_chunks.py 22 69 $CH > $OUT/rd.log 2>&1
python3 - <<PY
import csv, collections
for d in ("sq", "wr", "rd"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(dict)
    for r in csv.DictReader(open("$OUT/%s/p_counter_collection.csv" % d)):
        n = r["Kernel_Name"].split("(")[0].replace("spada::", "").replace("void ", "")
        if not n.startswith("k_"): continue
        agg[n][(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        dur[n][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for n in sorted(agg, key=lambda n: -max(dur[n].values())):
        ids = sorted(dur[n], key=lambda i: dur[n][i]); i = ids[-1]   # the longest launch
        print(d, n[:40], "us %.1f" % dur[n][i], {c: round(v / 1e6, 2) for (j, c), v in agg[n].items() if j == i}, "launches", len(ids))
PY
grep "== chunk\|symbolic\|numeric" $OUT/sq.log | cut -c1-400
