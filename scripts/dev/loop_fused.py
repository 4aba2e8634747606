#!/usr/bin/env python3
"""development aid: N one-pass calls on a surrogate (for counter passes); results are NOT checked (measurement builds give wrong ones)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spada_sim_amd as S
W = {"webbase": (S.GEN_WEBBASE_LIKE, 0, 0, 12347), "cop20k": (S.GEN_COP20K_LIKE, 0, 0, 12346)}
kind, p0, p1, seed = W[sys.argv[1]]
m = S.generate(kind, p0, p1, seed)
eng = S.Engine(); d = eng.upload(m)
cap = S.count_products(m, m, 0, m.shape[0])
for _ in range(int(sys.argv[2])):
    try:
        eng.fused_owned(d, d, 0, m.shape[0], cap)
    except Exception as e:
        print("call failed:", e)
st = eng.stats()
print(st["ms_task"], st["c_nnz"], st["n_tasks"])
