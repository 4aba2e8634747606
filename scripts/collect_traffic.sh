#!/bin/bash
# usage (GPU box): scripts/collect_traffic.sh <workload> [round-tag]
# Two separate PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots) plus a
# kernel-trace/stats pass of the SAME bench command; writes gpurun_out/traffic_<workload>/ (merged back by gpurun), from where
# r02_traffic_<workload>.json and r02_<workload>_kernel_stats.csv are copied into profiles/.
# FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.
set -e
WL=${1:-webbase-1M}
TAG=${2:-r06}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/traffic_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $WL"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $CMD > $OUT/stats.log 2>&1
cp $OUT/stats/p_kernel_stats.csv $OUT/${TAG}_${WL}_kernel_stats.csv
# the launches of the one-pass task kernel one by one (the summary's average includes the aborted run of the first call)
python3 - <<PY
import csv
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open("$OUT/stats/p_kernel_trace.csv")) if "k_task<2" in r["Kernel_Name"])
open("$OUT/${TAG}_${WL}_k_task_launches.txt", "w").write("# per-launch durations (us) of k_task<2, 2048> in the kernel trace behind ${TAG}_${WL}_kernel_stats.csv ($CMD)\n" + " ".join(f"{x:.2f}" for x in d) + "\n")
PY
python3 $REPO/scripts/pmc_per_launch.py $OUT/${TAG}_pmc_$WL.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over: $CMD; KiB." $OUT/fetch $OUT/write
python3 - <<PY
import json
src = json.load(open("$OUT/${TAG}_pmc_$WL.json"))
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per STEADY launch (median; a launch shorter than 5 % of "
                "the kernel's median duration -- the aborted run of a context's first call -- is dropped: launches_used / launches_seen) of "
                "bench.py --steps 10 --warmup 2; bytes = KiB * 1024.  gfx950: FETCH_SIZE under-counts wide coalesced streams "
                "by 2x (MI355X_MICROARCH.md); the gathers of this path are 4-8 B wide, so the raw value is reported uncorrected "
                "(fetch_bytes_per_launch) with the x2 upper bound next to it; hbm_bytes_per_launch uses the x2 bound.  The inputs "
                "(A, B, descriptors: < 100 MB) are Infinity-Cache resident, so fetched bytes << algorithmic bytes.",
       "_command": "$CMD"}
for n, d in sorted(src.items()):
    if not n.startswith("k_"): continue
    fb = d.get("FETCH_SIZE", 0.0) * 1024; wb = d.get("WRITE_SIZE", 0.0) * 1024
    out[n] = {"fetch_bytes_per_launch": fb, "fetch_bytes_per_launch_x2": 2 * fb, "write_bytes_per_launch": wb,
              "hbm_bytes_per_launch": 2 * fb + wb, "launches_used": d.get("launches_used"), "launches_seen": d.get("launches_seen")}
json.dump(out, open("$OUT/${TAG}_traffic_$WL.json", "w"), indent=1)
for n in out:
    if n.startswith("k_"): print(n, out[n])
PY
