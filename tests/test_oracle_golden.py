"""CPU: the oracle (oracle/spgemm_ref.c) against the committed golden vectors.

Loader fixtures come from the reference's own embedded Python loader; product fixtures come from
scipy/numpy (see tests/golden/make_golden.py) -- the arithmetic half of the oracle is cross-checked,
not reference-pinned (the Rust reference cannot be built)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle

CASES = ["tiny_cancel", "explicit_zero", "rect_aat", "rand_sq_300", "skewed_600", "denseish_64x200"]


def load_case(name):
    g = np.load(os.path.join(GOLDEN, "product_cases.npz"))
    def csr(p):
        sh = g[f"{name}_{p}_shape"]
        return oracle.Csr(sh[0], sh[1], g[f"{name}_{p}_indptr"], g[f"{name}_{p}_indices"], g[f"{name}_{p}_data"])
    a, b = csr("A"), csr("B")
    exp = oracle.Csr(a.rows, b.cols, g[f"{name}_C_indptr"], g[f"{name}_C_indices"], g[f"{name}_C_data"])
    return a, b, exp


@pytest.mark.parametrize("name", CASES)
def test_sortmerge_matches_golden(name):
    a, b, exp = load_case(name)
    c = oracle.spgemm_sortmerge(a, b)
    assert np.array_equal(c.indptr, exp.indptr)
    assert np.array_equal(c.indices, exp.indices)
    aa = oracle.Csr(a.rows, a.cols, a.indptr, a.indices, np.abs(a.data))
    bb = oracle.Csr(b.rows, b.cols, b.indptr, b.indices, np.abs(b.data))
    scale = oracle.spgemm_sortmerge(aa, bb).data
    assert np.all(np.abs(c.data - exp.data) <= 1e-12 * scale + 1e-300)


@pytest.mark.parametrize("name", CASES)
def test_spa_variant_is_bit_identical(name):
    a, b, _ = load_case(name)
    c1 = oracle.spgemm_sortmerge(a, b)
    for nt in (1, 3):
        c2 = oracle.spgemm_spa(a, b, n_threads=nt)
        assert np.array_equal(c1.indptr, c2.indptr)
        assert np.array_equal(c1.indices, c2.indices)
        assert np.array_equal(c1.data, c2.data)


def test_cancelled_and_explicit_zeros_are_kept():
    a, b, exp = load_case("tiny_cancel")
    c = oracle.spgemm_sortmerge(a, b)
    # row 1: empty A row; row 2: A nonzero pointing at an empty B row; row 3: 0.5*4 - 0.5*4 == 0.0 stored
    assert list(np.diff(c.indptr.astype(np.int64))) == [4, 0, 0, 4]
    r3 = slice(int(c.indptr[3]), int(c.indptr[4]))
    assert list(c.indices[r3]) == [0, 1, 2, 4] and c.data[r3][0] == 0.0
    a, b, _ = load_case("explicit_zero")
    c = oracle.spgemm_sortmerge(a, b)
    assert c.nnz == 3 and c.data[0] == 0.0   # 0.0*0.0 product of the stored zero is kept


def test_transpose_and_from_mat_rule():
    a, b, _ = load_case("rect_aat")          # 37 x 91 -> B = A^T
    a2, b2 = oracle.from_mat(a)
    assert (b2.rows, b2.cols) == (a.cols, a.rows)
    assert np.array_equal(b2.indptr, b.indptr) and np.array_equal(b2.indices, b.indices) and np.array_equal(b2.data, b.data)
    sq, _, _ = load_case("rand_sq_300")
    assert oracle.from_mat(sq)[1] is sq


def test_cari_product_pins(matrices_dir):
    import spada_sim_amd as S
    m = S.load_mm_mat(matrices_dir, "cari")
    a = oracle.Csr(m.shape[0], m.shape[1], m.indptr, m.indices, m.data)
    a, b = oracle.from_mat(a)
    g = np.load(os.path.join(GOLDEN, "cari_product.npz"))
    assert oracle.count_products(a, b) == int(g["nprod"]) == 57760800
    c = oracle.spgemm_spa(a, b)
    assert c.nnz == int(g["nnz"]) == 160000
    assert np.array_equal(c.indptr, g["indptr"])
    assert hashlib.sha256(c.indices.tobytes()).hexdigest() == str(g["indices_sha256"])
    assert np.allclose(c.data[:5], g["row0_head"], rtol=1e-12, atol=0)
    assert np.allclose(c.data[int(c.indptr[9]):int(c.indptr[9]) + 5], g["row9_head"], rtol=1e-12, atol=0)
    assert np.allclose(c.data[g["sample_pos"].astype(np.int64)], g["sample_val"], rtol=1e-12, atol=0)
    assert np.allclose(np.add.reduceat(c.data, c.indptr[:-1].astype(np.int64)), g["row_sums"], rtol=1e-12, atol=0)
    assert abs(c.data.sum() - float(g["total"])) < 1e-9 and c.data.max() == pytest.approx(float(g["vmax"]), rel=1e-12)
