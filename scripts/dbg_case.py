import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import spada_sim_amd as S
from oracle import oracle
from test_oracle_golden import load_case
name = sys.argv[1]
a, b, exp = load_case(name)
ma = S.CsMat((a.rows, a.cols), a.indptr, a.indices, a.data)
mb = ma if name in ("rand_sq_300", "skewed_600", "explicit_zero") else S.CsMat((b.rows, b.cols), b.indptr, b.indices, b.data)
eng = S.Engine()
c = eng.spgemm(ma, mb)
ref = oracle.spgemm_sortmerge(a, b)
print("shape", ma.shape, mb.shape, "nnzC", ref.nnz, "indptr ok", np.array_equal(c.indptr, ref.indptr))
st = eng.stats(); print("sym", st["sym_bin_rows"], "num", st["num_bin_rows"])
ip = ref.indptr.astype(np.int64)
L = np.diff(a.indptr.astype(np.int64))
bad = 0
for r in range(a.rows):
    s, t = ip[r], ip[r + 1]
    if not np.array_equal(c.indices[s:t], ref.indices[s:t]):
        bad += 1
        if bad <= 6:
            print("row", r, "L", L[r], "n", t - s, "\n got", c.indices[s:t], "\n exp", ref.indices[s:t])
print("bad rows", bad, "of", a.rows)

import ctypes
from spada_sim_amd import _ffi
Lb = _ffi.lib()
def buf(which, n, dt):
    out = np.zeros(n, dt)
    Lb.spada_debug_buf.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64]
    Lb.spada_debug_buf(eng._ctx, which, out.ctypes.data_as(ctypes.c_void_p), out.nbytes)
    return out
print("num_rows list", buf(0, 8, np.uint32))
print("row_bin", buf(2, a.rows, np.uint8))
print("elen", buf(3, int(a.indptr[-1]), np.uint32)[:40])
print("batch_num", buf(5, 4, np.uint32))
print("cptr", buf(7, a.rows+1, np.uint64))
