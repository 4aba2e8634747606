import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import spada_sim_amd as S
from oracle import oracle
from test_oracle_golden import load_case
name = sys.argv[1]
a, b, exp = load_case(name)
ma = S.CsMat((a.rows, a.cols), a.indptr, a.indices, a.data)
mb = ma if name in ("rand_sq_300", "skewed_600", "explicit_zero") else S.CsMat((b.rows, b.cols), b.indptr, b.indices, b.data)
ref = oracle.spgemm_sortmerge(a, b)
eng = S.Engine()
da = eng.upload(ma); db = da if mb is ma else eng.upload(mb)
dev = torch.device("cuda", 0)
ip = ref.indptr.astype(np.int64)
for it in range(40):
    nnz = eng.symbolic(da, db, 0, ma.shape[0])
    cp = torch.empty(ma.shape[0] + 1, dtype=torch.int64, device=dev)
    ci = torch.full((max(nnz, 1),), 0x7FFFFFF0, dtype=torch.int32, device=dev)
    cv = torch.full((max(nnz, 1),), -7.0, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    eng.numeric(cp.data_ptr(), ci.data_ptr(), cv.data_ptr())
    idx = ci.cpu().numpy()[:nnz].astype(np.int64)
    if nnz != ref.nnz or not np.array_equal(idx, ref.indices.astype(np.int64)):
        badpos = np.nonzero(idx != ref.indices.astype(np.int64))[0] if nnz == ref.nnz else []
        rows = np.searchsorted(ip, badpos, side="right") - 1
        print("iter", it, "nnz", nnz, ref.nnz, "bad positions", badpos[:10], "rows", rows[:10], "got", idx[badpos[:10]], "exp", ref.indices[badpos[:10]])
print("done")
