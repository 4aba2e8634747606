"""CPU: the cycle-level Spada model (include/spada_cycle.h) through the C ABI and the CLI.

What can be checked without the Rust tool (parity with it is UNPINNED, see the header of spada_cycle.h): the product the
model assembles equals the CPU oracle's (structure bit-exact, values within 1e-12 relative -- the model adds the products in
the accelerator's order, the oracle in ascending k), the counters obey the invariants the reference's definitions imply,
runs are reproducible, and the counters of the shipped workload (cari, shipped configuration) are pinned to the values this
restatement produced when it was written, so that any later change of its behaviour shows."""
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import assert_parity, to_oracle
from fuzz_cases import random_case
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIG = os.path.join(ROOT, "config", "config_1mb_row1.json")
RTOL = 1e-12


def config(**over):
    import spada_sim_amd as S
    return dict(S.parse_config(CONFIG), **over)


def simulate(a, b, accelerator="spada", row_remap=None, **over):
    import spada_sim_amd as S
    m = S.CycleModel(a, b, config(**over), accelerator=accelerator, row_remap=row_remap)
    try:
        m.execute(max_cycles=500_000_000)
        return m.result(), m.counts()
    finally:
        m.close()


def check_product_and_counters(a, b, c, k, cfg):
    ao = to_oracle(a)
    bo = ao if b is a else to_oracle(b)
    ref = oracle.spgemm_sortmerge(ao, bo)
    assert_parity(c, ref, ao, bo, RTOL)
    nprod = oracle.count_products(ao, bo)
    assert k["c_nnz"] == ref.nnz
    assert k["a_read"] == 2 * a.nnz() and k["a_write"] == 0       # every A scalar is fetched once: two words (storage.rs:312)
    assert k["b_write"] == 0
    assert k["c_write"] >= 2 * ref.nnz                              # every finished fiber is swapped out to the psum DRAM
    assert k["exec_cycles"] <= k["raw_cycles"]
    assert k["raw_cycles"] * cfg["pe_num"] * cfg["lane_num"] >= nprod   # one product per lane and cycle at most
    assert k["cache_read"] + k["cache_miss"] >= 2 * nprod           # every product takes its B element from the cache (the
                                                                    # request that misses is counted as a miss, not as a read)
    assert k["windows"] >= (a.nnz() + cfg["lane_num"] - 1) // cfg["lane_num"]


@pytest.mark.parametrize("accelerator", ["spada", "ip", "op", "multirow"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_small_square_matrices_all_accelerators(accelerator, seed):
    import spada_sim_amd as S
    a = S.generate(S.GEN_UNIFORM, 90, 7, seed)
    c, k = simulate(a, a, accelerator)
    check_product_and_counters(a, a, c, k, config())


@pytest.mark.parametrize("seed", [3, 5, 11, 14, 17, 21, 26, 33])
def test_random_cases_with_empty_rows_and_signed_values(seed):
    """A x B with empty A rows, A entries that select empty B rows, rectangular shapes, signed values (cancellation: the
    stored zeros are kept, simulator.rs:199-230)."""
    a, b, desc = random_case(seed)
    if a.nnz() > 20000 or oracle.count_products(to_oracle(a), to_oracle(b)) > 400000:
        pytest.skip("too large for the cycle model in a unit test: " + desc)
    c, k = simulate(a, b)
    check_product_and_counters(a, b, c, k, config())


def test_exact_cancellation_keeps_the_zero():
    import spada_sim_amd as S
    # row 0 of A = [1, 1] over rows 0 and 1 of B, which hold +x and -x in column 2
    a = S.CsMat((2, 2), np.array([0, 2, 2], np.uint64), np.array([0, 1], np.uint64), np.array([1.0, 1.0]))
    b = S.CsMat((2, 4), np.array([0, 1, 2], np.uint64), np.array([2, 2], np.uint64), np.array([0.75, -0.75]))
    c, k = simulate(a, b)
    assert c.nnz() == 1 and int(c.indices[0]) == 2 and c.data[0] == 0.0
    assert list(c.indptr) == [0, 1, 1]


@pytest.mark.parametrize("accelerator", ["spada", "ip", "op", "multirow"])
def test_fewer_rows_than_lanes(accelerator):
    """One to three rows of A (the reference's Op run would index past the end of A here: the model fits the first block)."""
    import spada_sim_amd as S
    for rows in (1, 3):
        rng = np.random.default_rng(rows)
        lens = rng.integers(0, 6, rows)
        indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        idx = np.concatenate([np.sort(rng.choice(5, int(l), replace=False)) for l in lens]).astype(np.uint64) if lens.sum() else np.zeros(0, np.uint64)
        a = S.CsMat((rows, 5), indptr, idx, rng.uniform(0.1, 1.0, int(lens.sum())))
        b = S.generate(S.GEN_UNIFORM, 5, 3, 7)
        c, k = simulate(a, b, accelerator)
        ao, bo = to_oracle(a), to_oracle(b)
        assert_parity(c, oracle.spgemm_sortmerge(ao, bo), ao, bo, RTOL)


def test_a_fiber_larger_than_the_cache_is_an_error_not_a_crash():
    """The reference panics when a single B row or partial fiber does not fit the fiber cache (storage.rs:604-609, :643-648); the
    model reports it."""
    import spada_sim_amd as S
    a = S.generate(S.GEN_UNIFORM, 40, 12, 3)
    with pytest.raises(S.SpadaError):
        simulate(a, a, cache_size=64)


def test_power_law_rows_merge_tasks_and_policy():
    """R-MAT rows: long rows are cut into many windows whose partial fibers are merged by the adder trees (and by PE pairs once
    A is exhausted); the Spada policy samples block heights 1, 2, 4, 8."""
    import spada_sim_amd as S
    a = S.generate(S.GEN_RMAT, 9, 8, 3)
    c, k = simulate(a, a)
    check_product_and_counters(a, a, c, k, config())
    assert k["tree_merge_tasks"] > 0
    assert k["blocks"] < a.shape[0]          # blocks taller than one row were issued


@pytest.mark.parametrize("over", [dict(lane_num=4), dict(pe_num=4, at_num=4), dict(at_num=1), dict(cache_size=4096),
                                  dict(cache_size=2048, at_num=2)])
def test_other_configurations(over):
    """Fewer lanes, more PEs, one adder tree, and caches so small that B rows and partial fibers are evicted all the time
    (psum fibers spill to the psum DRAM and are read back: storage.rs:591-649)."""
    import spada_sim_amd as S
    a = S.generate(S.GEN_UNIFORM, 120, 10, 4)
    c, k = simulate(a, a, **over)
    check_product_and_counters(a, a, c, k, config(**over))
    if over.get("cache_size", 1 << 20) <= 4096:
        assert k["b_evict"] > 0 and k["b_read"] > 2 * a.nnz()      # B rows fetched more than once


def test_runs_are_reproducible_and_row_remap_changes_nothing_in_c():
    import spada_sim_amd as S
    a = S.generate(S.GEN_RMAT, 8, 6, 9)
    c1, k1 = simulate(a, a)
    c2, k2 = simulate(a, a)
    assert k1 == k2 and np.array_equal(c1.data, c2.data) and np.array_equal(c1.indices, c2.indices)
    # -p: rows by ascending length, stable (preprocessing.rs:76-89); the product comes back under the original row numbers
    lens = np.diff(a.indptr.astype(np.int64))
    remap = np.argsort(lens, kind="stable").astype(np.uint64)
    c3, k3 = simulate(a, a, row_remap=remap)
    ao = to_oracle(a)
    assert_parity(c3, oracle.spgemm_sortmerge(ao, ao), ao, ao, RTOL)
    assert k3["a_read"] == k1["a_read"] and k3["c_nnz"] == k1["c_nnz"]


def test_argument_errors():
    import spada_sim_amd as S
    a = S.generate(S.GEN_UNIFORM, 10, 3, 1)
    b = S.CsMat((7, 7), np.zeros(8, np.uint64), np.zeros(0, np.uint64), np.zeros(0))
    with pytest.raises(S.SpadaError):
        S.CycleModel(a, b, config())                       # A.cols != B.rows
    with pytest.raises(S.SpadaError):
        S.CycleModel(a, a, config(lane_num=6))             # not a power of two
    with pytest.raises(S.SpadaError):
        S.CycleModel(a, a, config(pe_num=0))
    m = S.CycleModel(a, a, config())
    with pytest.raises(S.SpadaError):
        m.counts()                                          # before execute
    m.execute()
    with pytest.raises(S.SpadaError):
        m.execute()                                         # twice
    m.close()


@pytest.fixture(scope="module")
def cari_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("cari")
    os.makedirs(d / "matrices")
    with gzip.open(os.path.join(ROOT, "tests", "golden", "cari.mtx.gz"), "rb") as f, open(d / "matrices" / "cari.mtx", "wb") as g:
        shutil.copyfileobj(f, g)
    os.makedirs(d / "config")
    shutil.copy(CONFIG, d / "config" / "config_1mb_row1.json")
    return d


# counters of `spada-sim accuratesimu spada ss cari config/config_1mb_row1.json` as THIS restatement produces them (not the
# Rust tool's: see the module docstring); 57.76 M products on 16 lanes need at least 3.61 M cycles
CARI_PINS = {"exec_cycles": 4267108, "raw_cycles": 4361281, "a_read": 305600, "a_write": 0, "b_read": 31737680, "b_write": 0,
             "c_read": 800, "c_write": 320800, "cache_read": 241391498, "cache_write": 158177604, "windows": 19104,
             "c_nnz": 160000}


def test_cli_cycle_model_on_cari(cari_dir):
    """The shipped workload through the command line (no GPU involved): stdout skeleton of main.rs:44-116 with the simulated
    counters, the first ten rows of the product against the committed product pin, the checksum against the oracle."""
    exe = os.environ.get("SPADA_BIN_PATH") or os.path.join(ROOT, "spada_sim_amd", "bin", "spada-sim")
    out = subprocess.run([exe, "accuratesimu", "spada", "ss", "cari", "config/config_1mb_row1.json", "--cycle-model",
                          "--output", "C.bin"], cwd=cari_dir, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert "Get GEMM cari" in lines and "-----Result-----" in lines and "-----Output product matrix" in lines
    got = {}
    for ln in lines:
        if ln.startswith("Execution count:"):
            got["exec_cycles"] = int(ln.split(":")[1])
        for name, key in (("A matrix", "a"), ("B matrix", "b"), ("C matrix", "c"), ("Cache", "cache")):
            if ln.startswith(name + " count:"):
                w = ln.split()
                got[key + "_read"], got[key + "_write"] = int(w[w.index("read") + 1]), int(w[w.index("write") + 1])
    for k, v in got.items():
        assert v == CARI_PINS[k], (k, v, CARI_PINS[k])
    assert len(got) == 9
    rows = [ln for ln in lines if ln.startswith("rowptr: ")]
    assert len(rows) == 10 and rows[0].startswith("rowptr: 0 indptr: [0, 1, 2, 3, 4] data: [2.14086603168")
    pin = np.load(os.path.join(ROOT, "tests", "golden", "cari_product.npz"))
    for r, key in ((0, "row0_head"), (9, "row9_head")):
        vals = [float(x) for x in rows[r].split("data: [")[1].rstrip("]").split(", ")]
        assert np.allclose(vals, pin[key], rtol=1e-9, atol=0)
    # the whole product, read back from the binary dump, against the oracle
    import spada_sim_amd as S
    c = S.read_bin(str(cari_dir / "C.bin"))
    a = S.load_mm_mat(str(cari_dir / "matrices"), "cari")
    g = S.GEMM.from_mat("cari", a)
    ao, bo = to_oracle(g.a), to_oracle(g.b)
    assert_parity(c, oracle.spgemm_sortmerge(ao, bo), ao, bo, 1e-9)
    assert c.nnz() == CARI_PINS["c_nnz"]


def test_cli_cycle_model_preprocess_and_accelerators(tmp_path):
    """-p and the other accelerator spellings through the command line on a small matrix: same product (checksum of the
    structure identical, values within rounding), different schedules."""
    import spada_sim_amd as S
    os.makedirs(tmp_path / "matrices")
    a = S.generate(S.GEN_RMAT, 7, 6, 2)
    S.write_mm_mat(str(tmp_path / "matrices" / "small.mtx"), a)
    exe = os.environ.get("SPADA_BIN_PATH") or os.path.join(ROOT, "spada_sim_amd", "bin", "spada-sim")
    sums, cycles = [], []
    for extra in ([], ["-p"], ["--preprocess-by", "products"]):
        for acc in ("Spada", "IP", "op", "MultiRow"):
            out = subprocess.run([exe, "AccurateSimu", acc, "SS", "small", CONFIG, "--cycle-model", "--checksum"] + extra,
                                 cwd=tmp_path, capture_output=True, text=True, timeout=120)
            assert out.returncode == 0, out.stderr
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("rows ")][0].split()
            sums.append((line[line.index("structure") + 1], float(line[line.index("sum") + 1])))
            cycles.append(int([ln for ln in out.stdout.splitlines() if ln.startswith("Execution count:")][0].split(":")[1]))
    assert len({s for s, _ in sums}) == 1
    assert max(v for _, v in sums) - min(v for _, v in sums) <= 1e-9 * abs(sums[0][1])
    assert len(set(cycles)) > 1
