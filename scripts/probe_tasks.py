#!/usr/bin/env python3
"""Per-phase timing of the task pipeline on the synthetic workloads (development aid, GPU box): one-pass and two-phase."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spada_sim_amd as S

W = {"webbase": (S.GEN_WEBBASE_LIKE, 0, 0, 12347), "cop20k": (S.GEN_COP20K_LIKE, 0, 0, 12346),
     "cage12": (S.GEN_CAGE12_LIKE, 0, 0, 12348), "mc2depi": (S.GEN_MC2DEPI_LIKE, 0, 0, 12349),
     "rmat16": (S.GEN_RMAT, 16, 16, 22), "rmat18": (S.GEN_RMAT, 18, 16, 22)}
names = sys.argv[1:] or ["webbase", "cop20k", "cage12", "mc2depi", "rmat16"]
eng = S.Engine()
for name in names:
    kind, p0, p1, seed = W[name]
    m = S.generate(kind, p0, p1, seed)
    d = eng.upload(m)
    cap = S.count_products(m, m, 0, m.shape[0])
    best = None
    for it in range(8):
        t0 = time.perf_counter()
        eng.fused_owned(d, d, 0, m.shape[0], cap)
        wall = (time.perf_counter() - t0) * 1e3
        st = eng.stats()
        if best is None or st["ms_fused_call"] < best[0]["ms_fused_call"]:
            best = (st, wall)
    st, wall = best
    dev = st["ms_fused_call"]
    print(f"== {name}: rows {m.shape[0]} nnzA {m.nnz()} nprod {st['nprod']} nnzC {st['c_nnz']}  tasks {st['n_tasks']} limit {st['task_product_limit']} multi-pass {st['multi_pass_tasks']} runs {st['pipeline_runs']}")
    print(f"   one pass: device {dev:.3f} ms (wall {wall:.3f})  stats {st['ms_row_stats']:.3f} big {st['ms_big_expand']:.3f} cut {st['ms_cut']:.3f} task {st['ms_task']:.3f}")
    print(f"   {st['c_nnz'] / dev / 1e6:.2f} G nnzC/s   read {st['bytes_read'] / dev / 1e6:.1f} GB/s ({st['bytes_read'] / dev / 1e6 / 8000 * 100:.2f}% of 8 TB/s)")
    print(f"   class rows {st['cls_rows'][:5]}  class products {st['cls_prod'][:5]}")
    best = None
    for it in range(5):
        eng.symbolic(d, d, 0, m.shape[0])
        s1 = eng.stats()
        eng.numeric_owned()
        s2 = eng.stats()
        tot = s1["ms_symbolic_call"] + s2["ms_numeric_call"]
        if best is None or tot < best[0]:
            best = (tot, s1, s2)
    tot, s1, s2 = best
    print(f"   two phase: {tot:.3f} ms = symbolic {s1['ms_symbolic_call']:.3f} (count kernel {s1['ms_task']:.3f}) + numeric {s2['ms_numeric_call']:.3f}")
    eng.free(d)
