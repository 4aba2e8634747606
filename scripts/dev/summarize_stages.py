#!/usr/bin/env python3
"""development aid: the per-variant table of scripts/dev/stage_counts.sh from a directory of its rocprofv3 outputs
   usage: summarize_stages.py <gpurun_out/stages_xxx> [variant ...]"""
import csv, collections, os, sys
out = sys.argv[1]
order = sys.argv[2:] or ["stop10", "stop1", "stop11", "stop12", "stop13", "stop2", "stop14", "stop3", "stop15", "stop16", "stop17", "stop18",
                         "stop19", "stop4", "stop5", "stop20", "stop6", "stop21", "spgemm"]
prev = None
for v in order:
    p = os.path.join(out, v, "p_counter_collection.csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(float); n = collections.Counter(); t = []
    for r in csv.DictReader(open(p)):
        if "k_task<1" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            if r["Counter_Name"] == "SQ_INSTS_VALU":
                t.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    d = {k: agg[k] / n[k] / 1e6 for k in agg}
    us = sorted(t)[len(t) // 2] if t else 0
    line = (f"{v:8s} us {us:7.1f} valu {d.get('SQ_INSTS_VALU', 0):6.1f} salu {d.get('SQ_INSTS_SALU', 0):6.1f} lds {d.get('SQ_INSTS_LDS', 0):5.1f} "
            f"vmem_rd {d.get('SQ_INSTS_VMEM_RD', 0):4.1f} wave_cycles {d.get('SQ_WAVE_CYCLES', 0):7.1f} wait_any {d.get('SQ_WAIT_ANY', 0):7.1f}")
    if prev:
        line += f"   | delta us {us - prev[0]:6.1f} valu {d.get('SQ_INSTS_VALU', 0) - prev[1]:6.1f} lds {d.get('SQ_INSTS_LDS', 0) - prev[2]:5.1f}"
    prev = (us, d.get('SQ_INSTS_VALU', 0), d.get('SQ_INSTS_LDS', 0))
    print(line)
