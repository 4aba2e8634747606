"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the golden vectors.

Bar: C indptr / column indices bit-exact; values within 1e-9 relative (BASELINE.json), with the
documented cancellation fallback |x - y| <= 1e-9 * sum|a_ik * b_kj| (counted, see conftest.assert_parity)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_parity, to_oracle
from oracle import oracle
from test_oracle_golden import CASES, load_case

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def as_csmat(o):
    import spada_sim_amd as S
    return S.CsMat((o.rows, o.cols), o.indptr, o.indices, o.data)


@pytest.mark.parametrize("name", CASES)
def test_golden_product_cases(engine, name):
    a, b, exp = load_case(name)
    ma = as_csmat(a)
    mb = ma if name in ("rand_sq_300", "skewed_600", "explicit_zero") else as_csmat(b)
    c = engine.spgemm(ma, mb)
    ref = oracle.spgemm_sortmerge(a, b)
    assert_parity(c, ref, a, b, RTOL)
    assert np.array_equal(c.indices, exp.indices) and np.array_equal(c.indptr, exp.indptr)


def test_cari_a_at_through_simulator_interface(engine, matrices_dir):
    """configs[0]: `accuratesimu spada ss cari config_1mb_row1.json` -> A * A^T (gemm.rs:41-53)."""
    import spada_sim_amd as S
    mat = S.load_mm_mat(matrices_dir, "cari")
    gemm = S.GEMM.from_mat("cari", mat)
    assert gemm.b.shape == (1200, 400)
    dram_a, dram_b = S.CsrMatStorage.init_with_gemm(gemm)
    sim = S.Simulator(2, 16, 8, 1572864, 8, len(dram_b.indptr), [1, 10000000], dram_a, dram_b, None, "Spada",
                      30, 0, 1.0, 16, 8.0, engine=engine)
    sim.execute()
    rows = sim.get_exec_result()
    g = np.load(os.path.join(GOLDEN, "cari_product.npz"))
    c = sim.result_matrix()
    assert len(rows) == 400 and c.nnz() == int(g["nnz"]) == 160000
    assert np.array_equal(c.indptr, g["indptr"])
    assert hashlib.sha256(c.indices.tobytes()).hexdigest() == str(g["indices_sha256"])
    assert np.allclose(rows[0].data[:5], g["row0_head"], rtol=RTOL, atol=0)
    assert np.allclose(rows[9].data[:5], g["row9_head"], rtol=RTOL, atol=0)
    assert np.allclose(c.data[g["sample_pos"].astype(np.int64)], g["sample_val"], rtol=RTOL, atol=0)
    a = to_oracle(gemm.a)
    ref = oracle.spgemm_spa(a, to_oracle(gemm.b))
    assert_parity(c, ref, rtol=RTOL)
    st = engine.stats()
    assert st["nprod"] == 57760800 and st["c_nnz"] == 160000


GEN = [
    ("uniform_small", 5, 2000, 6, 1),
    ("rmat_s12", 0, 12, 8, 2),          # power-law rows: hits most LDS bins
    ("rmat_s14", 0, 14, 16, 3),         # long rows: largest LDS bins + HBM spill path
    ("webbase_like_50k", 1, 50000, 155000, 4),
    ("cop20k_like_20k", 2, 20000, 0, 5),
    ("cage12_like_20k", 3, 20000, 0, 6),
    ("mc2depi_like", 4, 779 * 40, 0, 7),
]


@pytest.mark.parametrize("name,kind,p0,p1,seed", GEN)
def test_generated_workloads_match_oracle(engine, name, kind, p0, p1, seed):
    import spada_sim_amd as S
    m = S.generate(kind, p0, p1, seed)
    c = engine.spgemm(m, m)
    a = to_oracle(m)
    ref = oracle.spgemm_spa(a, a)
    nfallback = assert_parity(c, ref, a, a, RTOL)
    assert nfallback == 0     # positive values: no cancellation
    st = engine.stats()
    assert st["nprod"] == oracle.count_products(a, a) and st["c_nnz"] == ref.nnz


def test_spill_path_is_exercised(engine):
    """R-MAT scale 14, degree 16: hub rows whose accumulator does not fit LDS (SURVEY 7 hard part 3) are BIG rows: they
    become column-range tasks, and the products of the largest ones are spilled to HBM scratch range by range; the counters say
    how many."""
    import spada_sim_amd as S
    m = S.generate(S.GEN_RMAT, 14, 16, 3)
    engine.spgemm(m, m)
    st = engine.stats()
    ao = to_oracle(m)
    lens = np.diff(m.indptr.astype(np.int64))
    prod = np.array([lens[m.indices[int(m.indptr[r]):int(m.indptr[r + 1])].astype(np.int64)].sum() for r in range(m.shape[0])])
    assert st["task_product_limit"] == 2040
    # BIG: more products than a task hashes, or a shape no batch takes (more than 512 entries; one entry selecting more than 2048 products)
    big = ((prod > st["task_product_limit"]) & (lens > 1)) | ((lens > 512) & (prod > 0)) | ((lens == 1) & (prod > 2048))
    assert st["cls_rows"][4] == int(big.sum()) > 0 and st["cls_prod"][4] == int(prod[big].sum())
    # hubs (thousands of entries, hundreds of ranges) are spilled; BIG rows with few ranges are walked from B directly
    assert 0 < st["spill_rows"] < st["cls_rows"][4]
    assert 0 < st["scratch_products"] < st["cls_prod"][4]
    assert st["n_tasks"] > st["cls_rows"][4]


def test_empty_and_ragged_inputs(engine):
    import spada_sim_amd as S
    # all-empty matrix
    z = S.CsMat((5, 5), np.zeros(6, np.uint64), np.zeros(0, np.uint64), np.zeros(0))
    c = engine.spgemm(z, z)
    assert c.nnz() == 0 and np.array_equal(c.indptr, np.zeros(6, np.uint64))
    # one dense row against a diagonal, the rest empty
    n = 300
    indptr = np.zeros(n + 1, np.uint64)
    indptr[1:] = n
    a = S.CsMat((n, n), indptr, np.arange(n, dtype=np.uint64), np.linspace(0.5, 1.5, n))
    d = S.CsMat((n, n), np.arange(n + 1, dtype=np.uint64), np.arange(n, dtype=np.uint64), np.full(n, 2.0))
    c = engine.spgemm(a, d)
    ref = oracle.spgemm_sortmerge(to_oracle(a), to_oracle(d))
    assert_parity(c, ref, rtol=RTOL)
    # signed values with cancellation: structure exact, values within the cancellation bound
    rng = np.random.default_rng(5)
    m = S.generate(S.GEN_UNIFORM, 1500, 12, 9)
    m.data[:] = rng.uniform(-1, 1, m.nnz())
    ao = to_oracle(m)
    assert_parity(engine.spgemm(m, m), oracle.spgemm_spa(ao, ao), ao, ao, RTOL)


def test_device_resident_row_blocks_concatenate(engine):
    """A-row blocks (scheduler.rs:296-379) computed separately concatenate to the full product."""
    import spada_sim_amd as S
    m = S.generate(S.GEN_RMAT, 12, 8, 11)
    ao = to_oracle(m)
    ref = oracle.spgemm_spa(ao, ao)
    bounds = S.partition_rows(m, m, 3)
    d = engine.upload(m)
    parts = []
    for r0, r1 in zip(bounds[:-1], bounds[1:]):
        nnz = engine.symbolic(d, d, r0, r1)
        p, i, v = engine.numeric_owned()
        parts.append(engine.download(p, i, v, r1 - r0, nnz, m.shape[1]))
    engine.free(d)
    indices = np.concatenate([p.indices for p in parts])
    data = np.concatenate([p.data for p in parts])
    lens = np.concatenate([np.diff(p.indptr.astype(np.int64)) for p in parts])
    assert np.array_equal(np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64), ref.indptr)
    assert np.array_equal(indices, ref.indices)
    assert np.all(np.abs(data - ref.data) <= RTOL * np.abs(ref.data))


def test_repeatability(engine):
    """Run twice: identical structure, values within tolerance (LDS atomics add in arbitrary order)."""
    import spada_sim_amd as S
    m = S.generate(S.GEN_RMAT, 11, 8, 21)
    c1 = engine.spgemm(m, m)
    c2 = engine.spgemm(m, m)
    assert np.array_equal(c1.indptr, c2.indptr) and np.array_equal(c1.indices, c2.indices)
    assert np.all(np.abs(c1.data - c2.data) <= 1e-12 * np.abs(c1.data))
    # one symbolic call, two numeric calls (the ABI does not forbid it): the second fills its buffers like the first
    m = S.generate(S.GEN_RMAT, 13, 12, 23)          # BIG rows, spilled and direct ranges
    d = engine.upload(m)
    nnz = engine.symbolic(d, d, 0, m.shape[0])
    first = engine.download(*engine.numeric_owned(), m.shape[0], nnz, m.shape[1])
    second = engine.download(*engine.numeric_owned(), m.shape[0], nnz, m.shape[1])
    engine.free(d)
    assert np.array_equal(first.indptr, second.indptr) and np.array_equal(first.indices, second.indices)
    assert np.all(np.abs(first.data - second.data) <= 1e-12 * np.abs(first.data))
    ao = to_oracle(m)
    assert_parity(second, oracle.spgemm_spa(ao, ao), ao, ao, RTOL)


def test_error_paths(engine):
    import spada_sim_amd as S
    a = S.generate(S.GEN_UNIFORM, 50, 3, 1)
    b = S.CsMat((40, 40), np.zeros(41, np.uint64), np.zeros(0, np.uint64), np.zeros(0))
    with pytest.raises(S.SpadaError) as e:
        engine.spgemm(a, b)          # inner dimensions differ
    assert e.value.code == 1
    bad = S.CsMat((2, 2), np.array([0, 2, 2], np.uint64), np.array([1, 0], np.uint64), np.array([1.0, 2.0]))
    with pytest.raises(S.SpadaError):
        engine.spgemm(bad, bad)      # columns not ascending


# ---- sort-merge accumulator variant (BASELINE.json configs[2]: "LDS-hash vs sort-merge accumulator variants") ----
@pytest.fixture(scope="module")
def engine_sm():
    import spada_sim_amd as S
    e = S.Engine(accumulator=S.ACC_SORT_MERGE)
    yield e
    e.close()


@pytest.mark.parametrize("name", CASES)
def test_sort_merge_golden_product_cases(engine_sm, name):
    a, b, exp = load_case(name)
    ma = as_csmat(a)
    mb = ma if name in ("rand_sq_300", "skewed_600", "explicit_zero") else as_csmat(b)
    c = engine_sm.spgemm(ma, mb)
    ref = oracle.spgemm_sortmerge(a, b)
    assert_parity(c, ref, a, b, RTOL)


@pytest.mark.parametrize("name,kind,p0,p1,seed", [g for g in GEN if g[0] in ("uniform_small", "rmat_s12", "webbase_like_50k",
                                                                            "mc2depi_like")])
def test_sort_merge_generated_workloads(engine_sm, name, kind, p0, p1, seed):
    import spada_sim_amd as S
    m = S.generate(kind, p0, p1, seed)
    c = engine_sm.spgemm(m, m)
    a = to_oracle(m)
    ref = oracle.spgemm_sortmerge(a, a)
    assert assert_parity(c, ref, a, a, RTOL) == 0
    # every row -- batches and the column ranges of BIG rows alike -- is added in ascending k like the CPU restatement
    # (simulator.rs:209-220 adds left to right): values are bit-identical, not just within 1e-9
    assert np.array_equal(c.data, ref.data)
    st = engine_sm.stats()
    if name == "rmat_s12":
        assert st["cls_rows"][4] > 0      # BIG rows took the same sort-merge accumulator


# ---- flat-batch pipeline: edge cases of the batch cut, the composite keys and the bucket order ---------------------
def _random_csr(rng, rows, cols, row_lens, col_sampler):
    import spada_sim_amd as S
    indptr = np.zeros(rows + 1, np.uint64)
    idx, val = [], []
    for r in range(rows):
        c = np.unique(col_sampler(r, int(row_lens[r])))
        c = c[(c >= 0) & (c < cols)]
        idx.append(c.astype(np.uint64))
        val.append(rng.uniform(0.1, 1.0, len(c)))
        indptr[r + 1] = indptr[r] + len(c)
    return S.CsMat((rows, cols), indptr, np.concatenate(idx) if idx else np.zeros(0, np.uint64),
                   np.concatenate(val) if val else np.zeros(0))


def test_clustered_columns_with_far_outliers(engine):
    """Rows whose columns are tight clusters plus a few far outliers: the value-proportional buckets of the ordered
    emission degenerate (most entries in one bucket) -- the result must still be exact."""
    rng = np.random.default_rng(7)
    n = 3000

    def cols(r, k):
        base = (r * 37) % (n - 200)
        local = base + rng.integers(0, 40, size=k)             # dense cluster next to `base`
        far = rng.integers(0, n, size=max(1, k // 8))           # a few far links
        return np.concatenate([local, far])

    m = _random_csr(rng, n, n, rng.integers(2, 40, size=n), cols)
    ao = to_oracle(m)
    assert assert_parity(engine.spgemm(m, m), oracle.spgemm_spa(ao, ao), ao, ao, RTOL) == 0


@pytest.mark.parametrize("cols_log2", [24, 27, 30, 31])
def test_wide_column_spaces(engine, cols_log2):
    """cols up to 2^31: the (local row, column) composite key leaves 8, 5, 2 and 1 bits for the local row, i.e. at most
    255 / 31 / 3 rows per batch and finally the per-row kernels only (flat batches off)."""
    import spada_sim_amd as S
    rng = np.random.default_rng(cols_log2)
    rows, inner, cols = 400, 300, 1 << cols_log2
    a = _random_csr(rng, rows, inner, rng.integers(0, 12, size=rows), lambda r, k: rng.integers(0, inner, size=k))
    b = _random_csr(rng, inner, cols, rng.integers(0, 30, size=inner),
                    lambda r, k: rng.integers(0, cols, size=k, dtype=np.int64))
    c = engine.spgemm(a, b)
    ref = oracle.spgemm_sortmerge(to_oracle(a), to_oracle(b))
    assert_parity(c, ref, to_oracle(a), to_oracle(b), RTOL)


def test_row_chunks_stream_the_same_product(engine):
    """Chunked execution (what bench.py uses for R-MAT scale 22) concatenates to the one-shot product."""
    import spada_sim_amd as S
    import torch
    m = S.generate(S.GEN_RMAT, 12, 8, 5)
    ao = to_oracle(m)
    ref = oracle.spgemm_spa(ao, ao)
    bounds = S.partition_rows(m, m, 7)
    d = engine.upload(m)
    dev = torch.device("cuda", 0)
    got = {"idx": [], "val": [], "len": []}
    bufs = {}

    def alloc(nrows, nnz):
        bufs["p"] = torch.empty(nrows + 1, dtype=torch.int64, device=dev)
        bufs["i"] = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        bufs["v"] = torch.empty(max(nnz, 1), dtype=torch.float64, device=dev)
        return bufs["p"].data_ptr(), bufs["i"].data_ptr(), bufs["v"].data_ptr()

    def consume(b0, b1, nnz, st):
        got["idx"].append(bufs["i"][:nnz].cpu().numpy().astype(np.uint64))
        got["val"].append(bufs["v"][:nnz].cpu().numpy())
        got["len"].append(np.diff(bufs["p"].cpu().numpy()))

    total = engine.spgemm_row_chunks(d, d, bounds, alloc, consume)
    engine.free(d)
    assert total == ref.nnz
    assert np.array_equal(np.concatenate(got["idx"]), ref.indices)
    assert np.array_equal(np.concatenate([[0], np.cumsum(np.concatenate(got["len"]))]).astype(np.uint64), ref.indptr)
    assert np.all(np.abs(np.concatenate(got["val"]) - ref.data) <= RTOL * np.abs(ref.data))


def test_stats_account_for_every_product(engine):
    import spada_sim_amd as S
    m = S.generate(S.GEN_RMAT, 13, 12, 9)
    c = engine.spgemm(m, m)
    st = engine.stats()
    ao = to_oracle(m)
    assert st["nprod"] == oracle.count_products(ao, ao)
    assert sum(st["cls_rows"]) == m.shape[0] and sum(st["cls_prod"]) == st["nprod"]
    assert c.nnz() == st["c_nnz"] and st["a_nnz"] == m.nnz()
    assert st["bytes_read"] == (m.shape[0] + 1) * 8 + m.nnz() * 28 + st["nprod"] * 12
    assert st["bytes_write"] == (m.shape[0] + 1) * 8 + st["c_nnz"] * 12


def test_sort_merge_full_size_webbase(engine_sm):
    """configs[2]: the sort-merge accumulator over the SAME rows as the hash variant, at the full size of the webbase-1M
    surrogate (51.7 M nnz(C)): structure bit-exact, values bit-identical to the sequential sort-merge."""
    import spada_sim_amd as S
    m = S.generate(S.GEN_WEBBASE_LIKE, 0, 0, 12347)
    d = engine_sm.upload(m)
    nnz = engine_sm.symbolic(d, d, 0, m.shape[0])
    p, i, v = engine_sm.numeric_owned()
    c = engine_sm.download(p, i, v, m.shape[0], nnz, m.shape[1])
    st = engine_sm.stats()
    engine_sm.free(d)
    ao = to_oracle(m)
    ref = oracle.spgemm_spa(ao, ao)       # bit-identical to oracle.spgemm_sortmerge (test_oracle_golden.py)
    assert np.array_equal(c.indptr, ref.indptr) and np.array_equal(c.indices, ref.indices)
    assert np.array_equal(c.data, ref.data)
    assert st["cls_rows"][4] > 4000 and sum(st["cls_rows"]) == m.shape[0]


@pytest.mark.parametrize("seed", range(24))
def test_random_parity_sweep(engine, engine_sm, seed):
    """Random shapes and row-length profiles (tests/fuzz_cases.py), both accumulators."""
    from fuzz_cases import random_case
    a, b, desc = random_case(seed)
    ao, bo = to_oracle(a), to_oracle(b)
    ref = oracle.spgemm_sortmerge(ao, bo)
    for e in (engine, engine_sm):
        assert_parity(e.spgemm(a, b), ref, ao, bo, RTOL)


@pytest.fixture(scope="module")
def full_size_refs():
    """oracle products of the full-size bench workloads, computed once per (kind) and shared by the entry-point cases"""
    return {}


@pytest.mark.parametrize("entry", ["fused", "two_phase"])
@pytest.mark.parametrize("kind,name", [(1, "webbase-1M surrogate"), (2, "cop20k_A surrogate"), (3, "cage12 surrogate"),
                                       (4, "mc2depi surrogate")])
def test_bench_workloads_at_full_size(engine, full_size_refs, kind, name, entry):
    """Every workload bench.py times, at BASELINE.json's full sizes, through BOTH entry points -- `fused` is
    spada_dev_spgemm_fused, the one-pass call bench.py times by default; `two_phase` the symbolic + numeric contract
    (`--two-phase`) -- against the oracle (structure bit-exact, values within 1e-9): 51.7 M / 18.5 M / 15.5 M / 5.2 M nnz(C).
    The oracle's OpenMP SPA variant finishes them in seconds."""
    import spada_sim_amd as S
    seeds = {1: 12347, 2: 12346, 3: 12348, 4: 12349}
    m = S.generate(kind, 0, 0, seeds[kind])
    if kind not in full_size_refs:
        full_size_refs.clear()          # one reference product in memory at a time
        ao = to_oracle(m)
        full_size_refs[kind] = oracle.spgemm_spa(ao, ao)
    ref = full_size_refs[kind]
    d = engine.upload(m)
    try:
        if entry == "fused":
            cap = S.count_products(m, m, 0, m.shape[0])
            p, i, v, nnz = engine.fused_owned(d, d, 0, m.shape[0], cap)
        else:
            nnz = engine.symbolic(d, d, 0, m.shape[0])
            p, i, v = engine.numeric_owned()
        c = engine.download(p, i, v, m.shape[0], nnz, m.shape[1])
    finally:
        engine.free(d)
    assert nnz == ref.nnz
    assert np.array_equal(c.indptr, ref.indptr)
    assert np.array_equal(c.indices, ref.indices)
    assert np.all(np.abs(c.data - ref.data) <= RTOL * np.abs(ref.data))


@pytest.mark.parametrize("name", ["cop20k_A", "webbase-1M", "cage12", "mc2depi"])
def test_suitesparse_file_when_present(engine, name):
    """Real SuiteSparse inputs: the image ships none and there is no network, so this runs only where $SPADA_MTX_DIR/<name>.mtx
    exists (as bench.py does); both entry points against the oracle."""
    import spada_sim_amd as S
    d = os.environ.get("SPADA_MTX_DIR")
    if not d or not os.path.exists(os.path.join(d, name + ".mtx")):
        pytest.skip(f"$SPADA_MTX_DIR/{name}.mtx not present (no SuiteSparse file in the image)")
    m = S.load_mm_mat(d, name)
    gemm = S.GEMM.from_mat(name, m)
    a, b = gemm.a, gemm.b
    ao, bo = to_oracle(a), (to_oracle(a) if b is a else to_oracle(b))
    ref = oracle.spgemm_spa(ao, bo)
    assert_parity(engine.spgemm(a, b), ref, ao, bo, RTOL)
    assert_parity(engine.spgemm_fused(a, b), ref, ao, bo, RTOL)


def test_rmat22_row_ranges_against_oracle(engine):
    """BASELINE.json configs[4] at its full size: R-MAT scale 22, degree 16 (4.19 M rows and columns, 65 M entries, 146 G products
    -- bench.py streams C in 69 row chunks).  Three row ranges of about 0.2 G products each -- the hubs at the top, the middle, the
    tail -- through the same two-phase calls the chunks use, against the oracle: structure bit-exact, values within 1e-9.  The hub
    rows are spilled to HBM scratch and their histogram buckets (4096 columns) are wider than the table: column sub-range tasks."""
    import spada_sim_amd as S
    m = S.generate(S.GEN_RMAT, 22, 16, 22)
    assert m.shape[0] == 1 << 22 and m.nnz() > 60_000_000
    fine = S.partition_rows(m, m, 690)
    d = engine.upload(m)
    ao = to_oracle(m)
    try:
        for piece in (0, 345, 689):
            r0, r1 = int(fine[piece]), int(fine[piece + 1])
            nnz = engine.symbolic(d, d, r0, r1)
            st = engine.stats()
            c = engine.download(*engine.numeric_owned(), r1 - r0, nnz, m.shape[1])
            lo, hi = int(ao.indptr[r0]), int(ao.indptr[r1])
            sub = oracle.Csr(r1 - r0, ao.cols, (ao.indptr[r0:r1 + 1] - ao.indptr[r0]).astype(np.uint64), ao.indices[lo:hi], ao.data[lo:hi])
            ref = oracle.spgemm_spa(sub, ao)
            assert nnz == ref.nnz and st["nprod"] == oracle.count_products(sub, ao)
            assert np.array_equal(c.indptr, ref.indptr)
            assert np.array_equal(c.indices, ref.indices)
            assert np.all(np.abs(c.data - ref.data) <= RTOL * np.abs(ref.data))
            if piece == 0:
                assert st["spill_rows"] > 0 and st["multi_pass_tasks"] == 0 and st["cls_prod"][4] > 0.99 * st["nprod"]
            del c, ref, sub
    finally:
        engine.free(d)
