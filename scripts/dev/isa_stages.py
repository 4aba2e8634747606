#!/usr/bin/env python3
"""development aid: static instruction counts between the BT_MARK comments of a kernel in the gfx950 assembly
   usage: isa_stages.py <file.s> <mangled kernel name>"""
import sys, collections
lines = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
i = next(k for k, l in enumerate(lines) if l.startswith(name + ":"))
cur, order, cnt = "(before)", [], collections.defaultdict(collections.Counter)
while not lines[i].startswith(".Lfunc_end"):
    l = lines[i].strip()
    i += 1
    if "BT_MARK" in l:
        cur = l.split("BT_MARK")[1].strip()
        if cur not in order:
            order.append(cur)
        continue
    if not l or l[0] in ";." or l.endswith(":"):
        continue
    op = l.split()[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
    cnt[cur][kind] += 1
print("instructions AFTER each marker up to the next one (static; loops count once)")
for k in ["(before)"] + order:
    c = cnt[k]
    print(f"{k:24s} valu {c['valu']:5d} salu {c['salu']:5d} lds {c['lds']:4d} vmem {c['vmem']:4d}")
