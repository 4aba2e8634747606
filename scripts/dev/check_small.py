#!/usr/bin/env python3
"""development aid: a few small inputs with BIG rows through both entry points against the CPU oracle (quick bisecting on the GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spada_sim_amd as S
from oracle import oracle
for kind, p0, p1, seed in ((S.GEN_RMAT, 12, 16, 5), (S.GEN_RMAT, 14, 16, 9), (S.GEN_UNIFORM, 3000, 80, 2)):
    a = S.generate(kind, p0, p1, seed)
    ao = oracle.Csr(a.shape[0], a.shape[1], a.indptr, a.indices, a.data)
    ref = oracle.spgemm_spa(ao, ao)
    eng = S.Engine()
    for name, fn in (("two-phase", eng.spgemm), ("one-pass", eng.spgemm_fused), ("one-pass again", eng.spgemm_fused)):
        c = fn(a, a)
        st = eng.stats()
        ok = c.nnz() == ref.nnz and np.array_equal(c.indptr, ref.indptr) and np.array_equal(c.indices, ref.indices)
        print(f"{kind} {p0} {name:15s} nnz {c.nnz()} ref {ref.nnz} {'OK' if ok else 'WRONG'}  tasks {st['n_tasks']} big rows {st['cls_rows'][4]} spilled {st['spill_rows']} runs {st['pipeline_runs']}", flush=True)
    eng.close()
