#!/usr/bin/env python3
"""development aid (GPU box): wall time per one-pass call in a tight loop (as bench.py's timed loop: reused caller buffers, one raw
statistics read per call) for the library named by SPADA_LIB_PATH.  usage: ab_wall.py workload [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spada_sim_amd as S
W = {"webbase": (S.GEN_WEBBASE_LIKE, 0, 0, 12347), "cop20k": (S.GEN_COP20K_LIKE, 0, 0, 12346), "cage12": (S.GEN_CAGE12_LIKE, 0, 0, 12348),
     "mc2depi": (S.GEN_MC2DEPI_LIKE, 0, 0, 12349)}
for name in sys.argv[1].split(","):
    kind, p0, p1, seed = W[name]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    m = S.generate(kind, p0, p1, seed)
    eng = S.Engine(); d = eng.upload(m)
    cap = S.count_products(m, m, 0, m.shape[0])
    eng.set_phase_timing(False)
    for _ in range(10):
        eng.fused_owned(d, d, 0, m.shape[0], cap)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            eng.fused_owned(d, d, 0, m.shape[0], cap)
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    st = eng.stats()
    print(f"{name:8s} {best:.4f} ms per call (best of 3 x {n}); device {st['ms_fused_call']:.4f}, task kernel {st['ms_task']:.4f}", flush=True)
    eng.free(d); eng.close()
