#!/usr/bin/env python3
"""Per-phase timing of the SpGEMM pipeline on the synthetic workloads (development aid, GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
    best = None
    for it in range(6):
        t0 = time.perf_counter()
        nnz = eng.symbolic(d, d, 0, m.shape[0])
        eng.numeric_owned()
        wall = (time.perf_counter() - t0) * 1e3
        st = eng.stats()
        dev = st["ms_symbolic_call"] + st["ms_numeric_call"]
        if best is None or dev < best[0]:
            best = (dev, wall, st)
    dev, wall, st = best
    print(f"== {name}: rows {m.shape[0]} nnzA {m.nnz()} nprod {st['nprod']} nnzC {st['c_nnz']}")
    print(f"   device {dev:.3f} ms (wall {wall:.3f})  stats {st['ms_row_stats']:.3f} bin {st['ms_binning']:.3f} "
          f"sym {st['ms_symbolic']:.3f} scan {st['ms_scan']:.3f} num {st['ms_numeric']:.3f}")
    print(f"   {st['c_nnz'] / dev / 1e6:.2f} G nnzC/s   read {st['bytes_read'] / dev / 1e6:.1f} GB/s "
          f"({st['bytes_read'] / dev / 1e6 / 8000 * 100:.2f}% of 8 TB/s)")
    print(f"   sym bins {st['sym_bin_rows'][:9]}")
    print(f"   num bins {st['num_bin_rows'][:12]}  spill {st['spill_rows']}")
    eng.free(d)
