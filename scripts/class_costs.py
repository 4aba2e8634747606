#!/usr/bin/env python3
"""What the task kernel spends per ROW CLASS (GPU box; verdict r4 item 3): C = A' * A on the webbase-1M surrogate where A' keeps the
rows of ONE class of A (COPY / SMALL / SOLO / BIG, the classes of k_row_class_cut) and all other rows are empty.  The product's rows of the
kept class are exactly the rows the full product has, computed by the same tasks minus the packing with rows of other classes.
Prints per class: rows, products, nnz(C), tasks, the task kernel's time (best of 6, HIP events) and ns per 1000 products; with
--pmc-friendly it only runs every class once (for a rocprofv3 --pmc pass around it)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spada_sim_amd as S

LIMIT, EMAX, PMAX, SMALL = 2040, 512, 2048, 512
once = "--pmc-friendly" in sys.argv
name = next((a for a in sys.argv[1:] if not a.startswith("--")), "webbase")
W = {"webbase": (S.GEN_WEBBASE_LIKE, 0, 0, 12347), "cop20k": (S.GEN_COP20K_LIKE, 0, 0, 12346), "rmat16": (S.GEN_RMAT, 16, 16, 22)}
m = S.generate(*W[name])
n = m.shape[0]
ptr = m.indptr.astype(np.int64)
idx = m.indices.astype(np.int64)
rlen = np.diff(ptr)
elen = rlen[idx]                                       # products of every A entry (B = A)
P = np.add.reduceat(np.concatenate([elen, [0]]), ptr[:-1]) * (rlen > 0)
emax = np.maximum.reduceat(np.concatenate([elen, [0]]), ptr[:-1]) * (rlen > 0)
cls = np.full(n, "small", dtype=object)
cls[P > SMALL] = "solo"
cls[(P > LIMIT) | (rlen > EMAX) | (emax > PMAX)] = "big"
cls[(rlen == 1) & (P <= PMAX) & (P > 0)] = "copy"
cls[P == 0] = "empty"

eng = S.Engine()
db = eng.upload(m)


def run(mask, label):
    keep = np.repeat(mask, rlen)
    lens = np.where(mask, rlen, 0)
    a = S.CsMat(m.shape, np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64), m.indices[keep], m.data[keep])
    da = eng.upload(a)
    cap = max(int(P[mask].sum()), 1)
    best = None
    for _ in range(1 if once else 6):
        eng.fused_owned(da, db, 0, n, cap)
        st = eng.stats()
        if best is None or st["ms_task"] < best["ms_task"]:
            best = st
    eng.free(da)
    st = best
    pre = st["ms_row_stats"] + st["ms_big_expand"] + st["ms_cut"]
    print(f"{label:6s} rows {int(mask.sum()):8d}  products {st['nprod']:10d}  nnz(C) {st['c_nnz']:10d}  tasks {st['n_tasks']:6d}  "
          f"k_task {st['ms_task'] * 1e3:7.1f} us  ({st['ms_task'] * 1e6 / max(st['nprod'], 1) * 1e3:6.2f} ns / 1000 products, "
          f"{st['nprod'] / max(st['n_tasks'], 1):6.0f} products / task)  before the task kernel {pre * 1e3:6.1f} us", flush=True)
    return st


print(f"# {name}: C = A' * A with A' = the rows of one class of A (other rows emptied); k_task by HIP events, best of {1 if once else 6}")
run(np.ones(n, bool), "all")
for c in ("copy", "small", "solo", "big"):
    run(cls == c, c)
run((cls == "small") | (cls == "solo") | (cls == "copy"), "no-big")
run((cls == "small") | (cls == "solo"), "hashed")
eng.free(db)
eng.close()
