#!/usr/bin/env python3
"""Per-bin products / nnz(C) of a workload next to the row counts (development aid, GPU box).
usage: bin_profile.py <workload>   (prints the table; run under rocprofv3 with SPADA_SERIAL_BINS=1 for per-kernel times)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spada_sim_amd as S
from perf_probe import W

name = sys.argv[1] if len(sys.argv) > 1 else "webbase"
kind, p0, p1, seed = W[name]
m = S.generate(kind, p0, p1, seed)
eng = S.Engine()
d = eng.upload(m)
for it in range(3):
    nnz = eng.symbolic(d, d, 0, m.shape[0])
    p, i, v = eng.numeric_owned()
st = eng.stats()
c = eng.download(p, i, v, m.shape[0], nnz, m.shape[1])
ip = m.indptr.astype(np.int64)
L = np.diff(ip)
blen = L[m.indices.astype(np.int64)]
cs = np.concatenate([[0], np.cumsum(blen)])
P = cs[ip[1:]] - cs[ip[:-1]]
n = np.diff(c.indptr.astype(np.int64))
def symbin(P, L):
    b = np.full(P.shape, 8)
    for lim, k in ((24576, 7), (8192, 6), (2048, 5), (512, 4), (128, 3), (32, 2)):
        b[P <= lim] = k
    b[L == 1] = 1
    b[P == 0] = 0
    return b
sb = symbin(P, L)
print(f"{name}: rows {len(L)} nnzA {L.sum()} products {P.sum()} nnzC {n.sum()}")
print("symbolic bins: bin rows products")
for k in range(9):
    msk = sb == k
    print(f"  {k:2d} {msk.sum():8d} {P[msk].sum():12d}")
print("numeric bins (from stats rows):", st["num_bin_rows"])
# numeric bin replicate of num_bin_of
def numbin(n, P, L, vcap):
    bn = np.full(n.shape, 11)
    bn[n <= vcap] = 10
    for lim, k in ((6144, 9), (2048, 8), (1024, 7), (512, 6), (256, 5), (128, 4), (64, 3), (32, 2)):
        bn[n <= lim] = k
    bp = np.full(n.shape, 9)
    for lim, k in ((16384, 8), (8192, 7), (4096, 6), (2048, 4), (1024, 3), (512, 2)):
        bp[P <= lim] = k
    b = np.maximum(bn, bp)
    b[L == 1] = 1
    b[n == 0] = 0
    return b
for vcap in (0, 16384):
    nb = numbin(n, P, L, vcap)
    cnt = [int((nb == k).sum()) for k in range(12)]
    if cnt == list(st["num_bin_rows"]):
        print("numeric bins: bin rows products nnzC  (vcap %d)" % vcap)
        for k in range(12):
            msk = nb == k
            print(f"  {k:2d} {msk.sum():8d} {P[msk].sum():12d} {n[msk].sum():12d}")
        break
else:
    print("could not reproduce numeric binning", cnt)
print(f"device ms: sym {st['ms_symbolic_call']:.3f} num {st['ms_numeric_call']:.3f}")
