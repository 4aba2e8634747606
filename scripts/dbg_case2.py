import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import spada_sim_amd as S
from oracle import oracle
from test_oracle_golden import load_case, CASES
eng = S.Engine()
for rep in range(6):
  for name in CASES:
    a, b, exp = load_case(name)
    ma = S.CsMat((a.rows, a.cols), a.indptr, a.indices, a.data)
    mb = ma if name in ("rand_sq_300", "skewed_600", "explicit_zero") else S.CsMat((b.rows, b.cols), b.indptr, b.indices, b.data)
    c = eng.spgemm(ma, mb)
    ref = oracle.spgemm_sortmerge(a, b)
    ok = np.array_equal(c.indptr, ref.indptr) and np.array_equal(c.indices, ref.indices)
    if not ok:
        st = eng.stats()
        ip = ref.indptr.astype(np.int64)
        L = np.diff(a.indptr.astype(np.int64))
        print(rep, name, "FAIL indptr ok", np.array_equal(c.indptr, ref.indptr), "sym", st["sym_bin_rows"][:6], "num", st["num_bin_rows"][:7])
        if np.array_equal(c.indptr, ref.indptr):
            k = 0
            for r in range(a.rows):
                s, t = ip[r], ip[r + 1]
                if not np.array_equal(c.indices[s:t], ref.indices[s:t]):
                    k += 1
                    if k <= 3: print("   row", r, "L", L[r], "n", t - s, "got", c.indices[s:t][:12], c.data[s:t][:4], "exp", ref.indices[s:t][:12], ref.data[s:t][:4])
            print("   bad rows", k)
    else:
        print(rep, name, "ok")
