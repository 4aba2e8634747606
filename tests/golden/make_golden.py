#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Run in the BUILD container only (needs /root/reference).

Two sources of truth are used, and the fixtures say which:

1. loader pins  -- produced by EXECUTING the reference's own embedded Python loader
   (``retrieve_mm_mat``), which is extracted at run time from /root/reference/src/py2rust.rs
   (lines 64-79, the raw string handed to PyModule::from_code).  Nothing of that source is
   stored in this repository; only its outputs (shape / indptr / indices / data) are.
2. product pins -- the reference's arithmetic is Rust and cannot be built here, so the expected
   products come from scipy/numpy: structure from the boolean product of the STORED patterns
   (explicit zeros and cancelled sums are kept, as in simulator.rs:199-230), values from a dense
   float64 matmul.  These pin the oracle (oracle/spgemm_ref.c) as a cross-check, not as a
   reference run.

Data files: ``cari.mtx.gz`` is a gzip of the reference's shipped workload
/root/reference/matrices/cari.mtx (SuiteSparse Meszaros/cari; data, not source).
"""
import gzip
import hashlib
import os
import re
import shutil
import sys
import tempfile

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def reference_loader():
    """Compile the Python snippet embedded in load_mm_mat (py2rust.rs:62-97) and return
    its retrieve_mm_mat function."""
    text = open(os.path.join(REF, "src", "py2rust.rs")).read()
    body = text[text.index("pub fn load_mm_mat"):]
    code = re.search(r'let code = r#"(.*?)"#;', body, re.S).group(1)
    ns = {}
    exec(compile(code, "retrieve_mm_mat.py", "exec"), ns)
    return ns["retrieve_mm_mat"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def expected_product(A, B):
    """Structure from stored patterns, values from dense matmul (see module docstring)."""
    A = A.tocsr()
    B = B.tocsr()
    Ao = sp.csr_matrix((np.ones(A.nnz), A.indices, A.indptr), shape=A.shape)
    Bo = sp.csr_matrix((np.ones(B.nnz), B.indices, B.indptr), shape=B.shape)
    P = (Ao @ Bo).tocsr()
    P.sort_indices()
    dense = A.toarray() @ B.toarray()
    rows = np.repeat(np.arange(A.shape[0]), np.diff(P.indptr))
    vals = dense[rows, P.indices]
    return P.indptr.astype(np.uint64), P.indices.astype(np.uint64), vals.astype(np.float64)


def csr_fields(m, prefix):
    m = m.tocsr()
    return {prefix + "_shape": np.array(m.shape, dtype=np.uint64),
            prefix + "_indptr": m.indptr.astype(np.uint64),
            prefix + "_indices": m.indices.astype(np.uint64),
            prefix + "_data": m.data.astype(np.float64)}


def rand_csr(rng, rows, cols, row_len_fn, val_fn):
    indptr = [0]
    indices = []
    for r in range(rows):
        n = min(cols, int(row_len_fn(r)))
        cs = np.sort(rng.choice(cols, size=n, replace=False)) if n else np.zeros(0, dtype=np.int64)
        indices.append(cs)
        indptr.append(indptr[-1] + n)
    indices = np.concatenate(indices) if indices else np.zeros(0, dtype=np.int64)
    data = val_fn(len(indices))
    return sp.csr_matrix((data, indices, np.array(indptr)), shape=(rows, cols))


def main():
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import oracle

    load = reference_loader()

    # ---- 1a. cari through the reference loader ------------------------------------------
    shape, indptr, indices, data = load(os.path.join(REF, "matrices"), "cari")
    A = sp.csr_matrix((data, indices, indptr), shape=shape)
    assert A.has_canonical_format
    with open(os.path.join(REF, "matrices", "cari.mtx"), "rb") as f, \
            gzip.GzipFile(os.path.join(HERE, "cari.mtx.gz"), "wb", compresslevel=9, mtime=0) as g:
        shutil.copyfileobj(f, g)
    np.savez_compressed(
        os.path.join(HERE, "cari_loader.npz"),
        shape=np.array(shape, dtype=np.uint64),
        nnz=np.uint64(A.nnz),
        indptr=np.asarray(indptr, dtype=np.uint64),
        indices_sha256=sha(np.asarray(indices, dtype=np.uint64)),
        data_sha256=sha(np.asarray(data, dtype=np.float64)),
        indices_head=np.asarray(indices[:16], dtype=np.uint64),
        data_head=np.asarray(data[:16], dtype=np.float64),
        row_sums=np.asarray(A.sum(axis=1)).ravel(),
        col_sums=np.asarray(A.sum(axis=0)).ravel(),
    )

    # ---- 1b. small .mtx variants through the reference loader ----------------------------
    out = {}
    for fn in sorted(os.listdir(os.path.join(HERE, "mtx"))):
        if not fn.endswith(".mtx"):
            continue
        name = fn[:-4]
        shape, indptr, indices, data = load(os.path.join(HERE, "mtx"), name)
        out[name + "_shape"] = np.array(shape, dtype=np.uint64)
        out[name + "_indptr"] = np.asarray(indptr, dtype=np.uint64)
        out[name + "_indices"] = np.asarray(indices, dtype=np.uint64)
        out[name + "_data"] = np.asarray(data, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "mtx_loader.npz"), **out)

    # ---- 2a. cari product A * A^T (gemm.rs:41-53: non-square => A * A^T) -------------------
    B = A.T.tocsr()
    B.sort_indices()
    c_indptr, c_indices, c_data = expected_product(A, B)
    Ao = oracle.Csr.from_scipy(A)
    _, Bo = oracle.from_mat(Ao)
    assert np.array_equal(Bo.indptr, B.indptr.astype(np.uint64))
    assert np.array_equal(Bo.indices, B.indices.astype(np.uint64))
    assert np.array_equal(Bo.data, B.data)
    Co = oracle.spgemm_sortmerge(Ao, Bo)
    assert np.array_equal(Co.indptr, c_indptr) and np.array_equal(Co.indices, c_indices)
    assert np.allclose(Co.data, c_data, rtol=1e-12, atol=0)
    Cs = oracle.spgemm_spa(Ao, Bo)
    assert np.array_equal(Cs.indices, Co.indices) and np.array_equal(Cs.data, Co.data)
    nnzc = len(c_indices)
    sample = np.arange(0, nnzc, 97)
    np.savez_compressed(
        os.path.join(HERE, "cari_product.npz"),
        shape=np.array([A.shape[0], B.shape[1]], dtype=np.uint64),
        nprod=np.uint64(oracle.count_products(Ao, Bo)),
        nnz=np.uint64(nnzc),
        indptr=c_indptr,
        indices_sha256=sha(c_indices),
        total=np.float64(c_data.sum()),
        vmax=np.float64(c_data.max()),
        vmin=np.float64(c_data.min()),
        row0_head=c_data[:5],
        row9_head=c_data[c_indptr[9]:c_indptr[9] + 5],
        row_sums=np.add.reduceat(c_data, c_indptr[:-1].astype(np.int64)),
        sample_pos=sample.astype(np.uint64),
        sample_val=c_data[sample],
    )

    # ---- 2b. small product cases: inputs + expected C -------------------------------------
    rng = np.random.default_rng(20261001)
    cases = {}

    def add_case(name, Am, Bm=None):
        Am = Am.tocsr()
        if Bm is None:  # from_mat rule
            Bm = Am if Am.shape[0] == Am.shape[1] else Am.T.tocsr()
        Bm = Bm.tocsr()
        Bm.sort_indices()
        ip, ix, dv = expected_product(Am, Bm)
        Cc = oracle.spgemm_sortmerge(oracle.Csr.from_scipy(Am), oracle.Csr.from_scipy(Bm))
        assert np.array_equal(Cc.indptr, ip) and np.array_equal(Cc.indices, ix), name
        scale = np.abs(Am).toarray() @ np.abs(Bm).toarray()
        rows = np.repeat(np.arange(Am.shape[0]), np.diff(ip.astype(np.int64)))
        bound = 1e-12 * scale[rows, ix.astype(np.int64)] if len(ix) else np.zeros(0)
        assert np.all(np.abs(Cc.data - dv) <= bound + 1e-300), name
        cases.update(csr_fields(Am, name + "_A"))
        cases.update(csr_fields(Bm, name + "_B"))
        cases[name + "_C_indptr"] = ip
        cases[name + "_C_indices"] = ix
        cases[name + "_C_data"] = dv

    # tiny hand-made: empty A rows, A nonzeros pointing at empty B rows, exact cancellation
    A1 = sp.csr_matrix(np.array([[1.0, 0, 2.0, 0],
                                 [0, 0, 0, 0],
                                 [0, 3.0, 0, 0],
                                 [0.5, 0, -0.5, 0]]))
    B1 = sp.csr_matrix(np.array([[4.0, 0, 1.0, 0, 0],
                                 [0, 0, 0, 0, 0],       # empty B row: A row 2 -> empty C row
                                 [4.0, 2.0, 0, 0, -1.0],
                                 [9.0, 9.0, 9.0, 9.0, 9.0]]))  # never referenced
    add_case("tiny_cancel", A1, B1)   # row 3: 0.5*4 - 0.5*4 = 0.0 stays stored
    # explicit stored zero in A still produces (zero-valued) products
    A2 = sp.csr_matrix((np.array([0.0, 2.0, 1.0]), np.array([0, 1, 1]), np.array([0, 2, 3])), shape=(2, 2))
    add_case("explicit_zero", A2)
    # rectangular => A * A^T
    add_case("rect_aat", rand_csr(rng, 37, 91, lambda r: rng.integers(0, 9), lambda n: rng.uniform(-1, 1, n)))
    # random square, signed values
    add_case("rand_sq_300", rand_csr(rng, 300, 300, lambda r: rng.integers(0, 12), lambda n: rng.uniform(-1, 1, n)))
    # skewed rows: a few long rows + many short, some empty
    lens = np.where(rng.random(600) < 0.02, rng.integers(100, 400, 600), rng.integers(0, 5, 600))
    add_case("skewed_600", rand_csr(rng, 600, 600, lambda r: lens[r], lambda n: rng.uniform(0.1, 1.0, n)))
    # one dense-ish block: every row hits most columns (dense-accumulator territory)
    add_case("denseish_64x200", rand_csr(rng, 64, 200, lambda r: 120, lambda n: rng.uniform(0.1, 1.0, n)))
    np.savez_compressed(os.path.join(HERE, "product_cases.npz"), **cases)

    for fn in sorted(os.listdir(HERE)):
        p = os.path.join(HERE, fn)
        if os.path.isfile(p):
            print(f"{os.path.getsize(p):9d}  {fn}")


if __name__ == "__main__":
    main()
