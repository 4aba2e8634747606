#!/bin/bash
# usage (GPU box): scripts/collect_traffic.sh <workload>
# Two separate PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots) plus a
# kernel-trace/stats pass of the SAME bench command; writes profiles/r01_traffic_<workload>.json and
# profiles/r01_<workload>_kernel_stats.csv.  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.
set -e
WL=${1:-webbase-1M}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/traffic_$WL
mkdir -p $OUT $REPO/profiles
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $WL"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $CMD > $OUT/stats.log 2>&1
cp $OUT/stats/p_kernel_stats.csv $REPO/profiles/r01_${WL}_kernel_stats.csv   # profiles/ on the box is not merged back: gpurun_out/ is
python3 - <<PY
import csv, json, collections
def per_kernel(path, counter):
    agg = collections.defaultdict(float); calls = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        n = r["Kernel_Name"].replace("spada::", "").split("(")[0].replace("void ", "").strip()   # full template name
        agg[n] += float(r["Counter_Value"]); calls[n] += 1
    return {n: (agg[n] / calls[n], calls[n]) for n in agg}
f = per_kernel("$OUT/fetch/p_counter_collection.csv", "FETCH_SIZE")
w = per_kernel("$OUT/write/p_counter_collection.csv", "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), KiB per launch averaged over all launches of "
                "bench.py --steps 10 --warmup 2; bytes = KiB * 1024.  gfx950: FETCH_SIZE under-counts wide coalesced streams "
                "by 2x (MI355X_MICROARCH.md); the gathers here are 4-8 B wide, so the raw value is reported uncorrected and a "
                "x2 upper bound next to it.  The inputs (<= 60 MB) are Infinity-Cache resident, so fetched bytes << algorithmic.",
       "_command": "$CMD"}
for n in sorted(set(f) | set(w)):
    fb = f.get(n, (0, 0))[0] * 1024; wb = w.get(n, (0, 0))[0] * 1024
    out[n] = {"fetch_bytes_per_launch": fb, "fetch_bytes_per_launch_x2": 2 * fb, "write_bytes_per_launch": wb,
              "hbm_bytes_per_launch": fb + wb, "launches": f.get(n, (0, 0))[1]}
json.dump(out, open("$REPO/profiles/r01_traffic_$WL.json", "w"), indent=1)
json.dump(out, open("$OUT/r01_traffic_$WL.json", "w"), indent=1)
for n in out:
    if n.startswith("k_num") or n.startswith("k_sym_flat"): print(n, out[n])
PY
