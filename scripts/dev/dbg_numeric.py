#!/usr/bin/env python3
"""development aid: one-pass vs two-phase product of a small matrix, first mismatch located (row, class)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spada_sim_amd as S
from oracle import oracle

kind, p0, p1, seed = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (0, 11, 8, 5)))
m = S.generate(kind, p0, p1, seed)
ao = oracle.Csr(m.shape[0], m.shape[1], m.indptr, m.indices, m.data)
ref = oracle.spgemm_spa(ao, ao)
eng = S.Engine()
d = eng.upload(m)
cap = S.count_products(m, m, 0, m.shape[0])
p, i, v, nnz = eng.fused_owned(d, d, 0, m.shape[0], cap)
c1 = eng.download(p, i, v, m.shape[0], nnz, m.shape[1])
st = eng.stats()
print("fused: nnz", nnz, "ref", ref.nnz, "tasks", st["n_tasks"], "classes", st["cls_rows"][:5])
nnz2 = eng.symbolic(d, d, 0, m.shape[0])
p, i, v = eng.numeric_owned()
c2 = eng.download(p, i, v, m.shape[0], nnz2, m.shape[1])
lens = np.diff(m.indptr.astype(np.int64))
prod = np.array([lens[m.indices[int(m.indptr[r]):int(m.indptr[r + 1])].astype(np.int64)].sum() for r in range(m.shape[0])])
for name, c in (("fused", c1), ("two-phase", c2)):
    ok_p = np.array_equal(c.indptr, ref.indptr)
    ok_i = ok_p and np.array_equal(c.indices, ref.indices)
    ok_v = ok_i and bool(np.all(np.abs(c.data - ref.data) <= 1e-9 * np.abs(ref.data)))
    print(name, "indptr", ok_p, "indices", ok_i, "values", ok_v)
    if ok_p and not ok_v:
        bad = np.nonzero((c.indices != ref.indices) | ~(np.abs(c.data - ref.data) <= 1e-9 * np.abs(ref.data)))[0]
        rows = np.searchsorted(ref.indptr.astype(np.int64), bad, side="right") - 1
        ur = np.unique(rows)
        print("  bad entries", len(bad), "in", len(ur), "rows; first rows", ur[:10], "lens", lens[ur[:10]], "prods", prod[ur[:10]])
        r = int(ur[0])
        s, e = int(ref.indptr[r]), int(ref.indptr[r + 1])
        print("  row", r, "ref cols", ref.indices[s:e][:12], "got", c.indices[s:e][:12])
        print("  ref vals", ref.data[s:e][:6], "got", c.data[s:e][:6])
eng.free(d)
