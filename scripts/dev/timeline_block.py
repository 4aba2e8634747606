#!/usr/bin/env python3
"""development aid: a few one-pass calls on one row block of the web surrogate's 8-way partition (for a rocprofv3 kernel trace)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spada_sim_amd as S
m = S.generate(S.GEN_WEBBASE_LIKE, 0, 0, 12347)
eng = S.Engine(); d = eng.upload(m)
b = S.partition_rows(m, m, 8)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
r0, r1 = int(b[k]), int(b[k + 1])
cap = S.count_products(m, m, r0, r1)
for _ in range(6):
    eng.fused_owned(d, d, r0, r1, cap)
print(eng.stats()["ms_fused_call"])
