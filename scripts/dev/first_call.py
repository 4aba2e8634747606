#!/usr/bin/env python3
"""development aid (GPU box): the first one-pass call of fresh contexts on the web surrogate (wall), after one throw-away context has warmed
the runtime.  usage: first_call.py [workload] [contexts]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spada_sim_amd as S
W = {"webbase": (S.GEN_WEBBASE_LIKE, 0, 0, 12347), "cop20k": (S.GEN_COP20K_LIKE, 0, 0, 12346), "mc2depi": (S.GEN_MC2DEPI_LIKE, 0, 0, 12349)}
kind, p0, p1, seed = W[sys.argv[1] if len(sys.argv) > 1 else "webbase"]
m = S.generate(kind, p0, p1, seed)
cap = S.count_products(m, m, 0, m.shape[0])
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    t0 = time.perf_counter(); eng = S.Engine(); t1 = time.perf_counter()
    d = eng.upload(m); t2 = time.perf_counter()
    eng.fused_owned(d, d, 0, m.shape[0], cap); t3 = time.perf_counter()
    st = eng.stats()
    eng.fused_owned(d, d, 0, m.shape[0], cap); t4 = time.perf_counter()
    print(f"context {k}: create {1e3 * (t1 - t0):.3f} ms, upload {1e3 * (t2 - t1):.3f}, first call {1e3 * (t3 - t2):.3f} (engine's own clock {st['ms_wall_call']:.3f}, device {st['ms_fused_call']:.3f}, "
          f"runs {st['pipeline_runs']}, workspaces {st['workspace_bytes'] / 1e6:.0f} MB), second call {1e3 * (t4 - t3):.3f}", flush=True)
    eng.free(d); eng.close()
