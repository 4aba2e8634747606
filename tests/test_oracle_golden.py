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


def test_order_faithful_variant_bounds_the_summation_order_effect(matrices_dir):
    """The arithmetic half of the oracle cannot be pinned to the reference binary (no Rust toolchain, no golden vectors), so the
    one thing that could make the reference's values differ from the oracle's -- the order in which the products of an output
    entry are added -- is bounded instead: oracle.spgemm_windowed adds them the way the reference's dataflow does (K-windows of
    lane_num = 8 A scalars, partial fibers merged two at a time, oldest first: scheduler.rs:482-606, :381-480,
    adder_tree.rs:73-83) and must give the same structure and values within a few ulps of the ascending-k sum.

    Recorded maxima (this test prints them with -s):  cari A*A^T (382 products per entry, positive values): 3.6e-15 relative;
    signed random cases: 6.6e-16 of sum|a_ik b_kj| (the cancellation scale) -- six orders of magnitude inside the 1e-9 tolerance."""
    import spada_sim_amd as S
    from fuzz_cases import random_case
    m = S.load_mm_mat(matrices_dir, "cari")
    a = oracle.Csr(m.shape[0], m.shape[1], m.indptr, m.indices, m.data)
    a, b = oracle.from_mat(a)
    ref = oracle.spgemm_sortmerge(a, b)
    win = oracle.spgemm_windowed(a, b, 8)
    assert np.array_equal(win.indptr, ref.indptr) and np.array_equal(win.indices, ref.indices)
    rel = float(np.max(np.abs(win.data - ref.data) / np.abs(ref.data)))
    print(f"cari: max relative deviation of the windowed order from ascending k = {rel:.3e}")
    assert 0 < rel < 5e-15            # different order (not the same sum by accident), far inside 1e-9
    for lane in (1, 2, 16):           # lane_num = 1: every product its own fiber, pure pairwise tree
        w2 = oracle.spgemm_windowed(a, b, lane)
        assert np.array_equal(w2.indices, ref.indices) and np.max(np.abs(w2.data - ref.data) / np.abs(ref.data)) < 5e-15
    worst = 0.0
    for seed in range(40):
        x, y, desc = random_case(seed)
        if x.nnz() * y.nnz() == 0 or y.shape[1] > 100000:
            continue
        xo = oracle.Csr(x.shape[0], x.shape[1], x.indptr, x.indices, x.data)
        yo = oracle.Csr(y.shape[0], y.shape[1], y.indptr, y.indices, y.data)
        r = oracle.spgemm_sortmerge(xo, yo)
        w = oracle.spgemm_windowed(xo, yo, 8)
        assert np.array_equal(w.indptr, r.indptr) and np.array_equal(w.indices, r.indices), desc
        xa = oracle.Csr(xo.rows, xo.cols, xo.indptr, xo.indices, np.abs(xo.data))
        ya = oracle.Csr(yo.rows, yo.cols, yo.indptr, yo.indices, np.abs(yo.data))
        scale = oracle.spgemm_sortmerge(xa, ya).data
        if len(scale):
            worst = max(worst, float(np.max(np.abs(w.data - r.data) / scale)))
    print(f"random cases: max deviation relative to sum|a b| = {worst:.3e}")
    assert worst < 5e-15
    # kept zeros survive the merges too (nothing filters value == 0: simulator.rs:199-230)
    a, b, _ = load_case("tiny_cancel")
    w = oracle.spgemm_windowed(a, b, 8)
    r3 = slice(int(w.indptr[3]), int(w.indptr[4]))
    assert list(w.indices[r3]) == [0, 1, 2, 4] and w.data[r3][0] == 0.0
