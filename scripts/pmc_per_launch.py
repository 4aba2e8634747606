#!/usr/bin/env python3
"""Per-kernel, per-LAUNCH counter values from rocprofv3 --pmc passes (p_counter_collection.csv), steady launches only.

The first call of a context may run its pipeline twice (workspace growth); the aborted run launches every kernel and most of them
return at once.  Averaging such a launch in dilutes every per-launch figure (round 4's SQ / traffic files did).  Here a launch is
dropped when its duration (End - Start timestamp of the dispatch) is below 5 % of the median duration of that kernel's launches;
`launches_used` / `launches_seen` say what was kept, and the value reported is the MEDIAN over the kept launches (the mean is
printed beside it).

usage: pmc_per_launch.py OUT.json NOTE PASSDIR [PASSDIR ...]        (each PASSDIR holds p_counter_collection.csv)
"""
import collections
import csv
import json
import statistics
import sys


def short(n):
    return n.replace("spada::", "").replace("void ", "").split("(")[0].strip()


def per_launch(path):
    """{kernel: {dispatch_id: {"dur": ns, counter: value}}} (a counter split over several rows -- one per XCD -- is summed)"""
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        n = short(r["Kernel_Name"])
        if not n.startswith("k_"):
            continue
        d = out[n][int(r["Dispatch_Id"])]
        d[r["Counter_Name"]] += float(r["Counter_Value"])
        d["dur"] = float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out


def steady(launches):
    durs = sorted(v["dur"] for v in launches.values())
    med = durs[len(durs) // 2]
    return {k: v for k, v in launches.items() if v["dur"] >= 0.05 * med}


def summarize(passdirs):
    res = collections.defaultdict(dict)
    for p in passdirs:
        for kernel, launches in per_launch(p + "/p_counter_collection.csv").items():
            keep = steady(launches)
            counters = sorted({c for v in keep.values() for c in v if c != "dur"})
            for c in counters:
                vals = [v[c] for v in keep.values()]
                res[kernel][c] = statistics.median(vals)
                res[kernel][c + "__mean"] = sum(vals) / len(vals)
            res[kernel]["launches_seen"] = len(launches)
            res[kernel]["launches_used"] = len(keep)
    return res


if __name__ == "__main__":
    out_path, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    res = {"_note": note + "  Per launch: median over the steady launches (a launch shorter than 5 % of the kernel's median duration -- the "
                            "aborted run of a context's first call -- is dropped; launches_used / launches_seen); <counter>__mean = their mean."}
    res.update(summarize(dirs))
    json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
    for n, d in sorted(res.items()):
        if n.startswith("k_task"):
            print(n, {k: (round(v / 1e6, 2) if isinstance(v, float) else v) for k, v in sorted(d.items()) if not k.endswith("__mean")}, "(millions)")
