"""Seeded random A, B pairs for the parity sweeps (tests/test_gpu_parity.py::test_random_parity_sweep and
scripts/fuzz_parity.py): sizes from 1 to 2^21 columns, empty / power-law / few-long / dense / clustered rows, signed
values in 30 % of the cases."""
import numpy as np

import spada_sim_amd as S


def random_case(seed):
    rng = np.random.default_rng(seed)
    m = int(rng.choice([1, 3, 17, 64, 257, 900, 2500, 6000]))
    k = int(rng.choice([1, 5, 64, 300, 1500, 5000]))
    n = int(rng.choice([1, 7, 100, 1024, 4097, 70000, 1 << 21]))
    style_a = rng.choice(["uniform", "powerlaw", "fewlong", "empty"])
    style_b = rng.choice(["uniform", "powerlaw", "clustered", "dense"])

    def lens(rows, cols, style, scale):
        if style == "empty":
            l = np.where(rng.random(rows) < 0.7, 0, rng.integers(0, scale + 1, rows))
        elif style == "powerlaw":
            l = np.minimum((rng.pareto(1.2, rows) * scale / 4).astype(np.int64), cols)
        elif style == "fewlong":
            l = np.where(rng.random(rows) < 0.05, min(cols, scale * 20), rng.integers(0, 3, rows))
        elif style == "dense":
            l = np.full(rows, min(cols, scale * 4))
        else:
            l = rng.integers(0, scale + 1, rows)
        return np.minimum(l, cols).astype(np.int64)

    def build(rows, cols, style, scale, signed):
        l = lens(rows, cols, style, scale)
        indptr = np.zeros(rows + 1, np.uint64)
        idx, val = [], []
        for r in range(rows):
            if style == "clustered" and l[r] > 0:
                base = int(rng.integers(0, max(1, cols - 64)))
                c = np.unique(np.concatenate([base + rng.integers(0, 64, l[r]), rng.integers(0, cols, max(1, l[r] // 6))]))
            else:
                c = np.unique(rng.integers(0, cols, l[r])) if l[r] < cols else np.arange(cols)
            c = c[c < cols]
            idx.append(c.astype(np.uint64))
            v = rng.uniform(0.1, 1.0, len(c))
            if signed:
                v *= rng.choice([-1.0, 1.0], len(c))
            val.append(v)
            indptr[r + 1] = indptr[r] + len(c)
        return S.CsMat((rows, cols), indptr, np.concatenate(idx) if idx else np.zeros(0, np.uint64),
                       np.concatenate(val) if val else np.zeros(0))

    signed = bool(rng.random() < 0.3)
    a = build(m, k, style_a, int(rng.choice([2, 8, 30])), signed)
    b = build(k, n, style_b, int(rng.choice([2, 8, 40, 200])), signed)
    return a, b, f"seed {seed}: A {m}x{k} {style_a} nnz {a.nnz()}, B {k}x{n} {style_b} nnz {b.nnz()}, signed {signed}"
