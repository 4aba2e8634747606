#!/usr/bin/env python3
"""development aid: wall time per call of a tight loop of one-pass calls (what bench.py times), phase timing on / off"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spada_sim_amd as S
W = {"webbase": (S.GEN_WEBBASE_LIKE, 0, 0, 12347), "cop20k": (S.GEN_COP20K_LIKE, 0, 0, 12346), "mc2depi": (S.GEN_MC2DEPI_LIKE, 0, 0, 12349)}
for name in sys.argv[1:] or ["webbase"]:
    m = S.generate(*W[name])
    eng = S.Engine()
    d = eng.upload(m)
    cap = S.count_products(m, m, 0, m.shape[0])
    for pt in (True, False):
        eng.set_phase_timing(pt)
        for _ in range(5):
            eng.fused_owned(d, d, 0, m.shape[0], cap)
        t0 = time.perf_counter()
        dev = 0.0
        N = 50
        for _ in range(N):
            eng.fused_owned(d, d, 0, m.shape[0], cap)
            dev += eng.stats()["ms_fused_call"]
        wall = (time.perf_counter() - t0) / N * 1e3
        print(f"{name}: phase timing {pt}: wall per call {wall:.4f} ms, device {dev / N:.4f} ms, task {eng.stats()['ms_task']:.3f}")
    eng.free(d)
    eng.close()
