#!/usr/bin/env python3
"""Counts of the surrogate workloads next to the literature rows of SURVEY.md 8 (nnz(A), products, nnz(C)).

    python scripts/calibrate_surrogates.py [webbase-1M cop20k_A cage12 mc2depi]

Products and nnz(C) come from the CPU oracle (test infrastructure), so this script is a development aid, not product
code.  tests/test_host_cpu.py asserts the same counts within +-5 %.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import spada_sim_amd as S  # noqa: E402
from oracle import oracle  # noqa: E402

TARGETS = {   # rows, nnz(A), products, nnz(C)   
    "webbase-1M": (S.GEN_WEBBASE_LIKE, 12347, 1000005, 3105536, 69.5e6, 51.1e6),
    "cop20k_A": (S.GEN_COP20K_LIKE, 12346, 121192, 2624331, 79.9e6, 18.7e6),
    "cage12": (S.GEN_CAGE12_LIKE, 12348, 130228, 2032536, 34.6e6, 15.2e6),
    "mc2depi": (S.GEN_MC2DEPI_LIKE, 12349, 525825, 2100225, 8.4e6, 5.2e6),
}


def counts(a):
    ao = oracle.Csr(a.shape[0], a.shape[1], a.indptr, a.indices, a.data)
    lens = np.diff(a.indptr.astype(np.int64))
    prod_per_row = np.add.reduceat(np.append(lens[a.indices.astype(np.int64)], 0),
                                   np.minimum(a.indptr[:-1].astype(np.int64), a.nnz()))
    prod_per_row[lens == 0] = 0
    c = oracle.spgemm_spa(ao, ao, n_threads=oracle.num_threads())
    nc = np.diff(c.indptr.astype(np.int64))
    return lens, prod_per_row, nc


def main():
    names = sys.argv[1:] or list(TARGETS)
    for nm in names:
        kind, seed, rows, nnz_t, prod_t, nnzc_t = TARGETS[nm]
        t0 = time.time()
        a = S.generate(kind, 0, 0, seed)
        lens, ppr, nc = counts(a)
        nnz, prod, nnzc = int(lens.sum()), int(ppr.sum()), int(nc.sum())
        print(f"{nm}: rows {a.shape[0]} (target {rows})  nnz {nnz} ({nnz / nnz_t - 1:+.1%})  products {prod} "
              f"({prod / prod_t - 1:+.1%})  nnz(C) {nnzc} ({nnzc / nnzc_t - 1:+.1%})  compression {prod / max(nnzc, 1):.2f} "
              f"(target {prod_t / nnzc_t:.2f})  [{time.time() - t0:.1f} s]")
        edges = [0, 1, 64, 512, 1536, 3072, 8192, 24576, 1 << 62]
        h = np.histogram(ppr, bins=edges)[0]
        hp = [int(ppr[(ppr >= lo) & (ppr < hi)].sum()) for lo, hi in zip(edges[:-1], edges[1:])]
        print("   rows by products/row   " + "  ".join(f"<{e}: {c}" for e, c in zip(edges[1:-1] + ["inf"], h)))
        print("   products in those rows " + "  ".join(f"{p / max(prod, 1):.1%}" for p in hp))
        one = lens == 1
        print(f"   single-entry rows {int(one.sum())} with {int(ppr[one].sum())} products; max row len {int(lens.max())}, "
              f"max products/row {int(ppr.max())}, max nnz(C)/row {int(nc.max())}")


if __name__ == "__main__":
    main()
