#!/usr/bin/env python3
"""development aid (GPU box): spilled rows / products of one workload and the phase times"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spada_sim_amd as S
m = S.generate(S.GEN_WEBBASE_LIKE, 0, 0, 12347)
eng = S.Engine(); d = eng.upload(m); cap = S.count_products(m, m, 0, m.shape[0])
for it in range(4):
    eng.fused_owned(d, d, 0, m.shape[0], cap); st = eng.stats()
print({k: st[k] for k in ("spill_rows", "scratch_products", "n_tasks", "ms_big_expand", "ms_task", "pipeline_runs", "multi_pass_tasks")})
