"""GPU (-m gpu): the drop-in `spada-sim` binary (C++ host mirror over the C ABI) run as the reference is run
(BASELINE.json configs[0]: `accuratesimu spada ss cari config/config_1mb_row1.json`), its stdout skeleton
(main.rs:44-116, storage.rs:115-126, frontend.rs:52-79), its writers, and -p / --preprocess (main.rs:60-63)."""
import gzip
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_parity, to_oracle
from oracle import oracle

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "spada_sim_amd", "bin", "spada-sim")
CFG = os.path.join(ROOT, "config", "config_1mb_row1.json")
RTOL = 1e-9


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    """A working directory laid out like the reference's: ./matrices/cari.mtx (config ss_filepath = ./matrices)."""
    d = tmp_path_factory.mktemp("run")
    os.makedirs(d / "matrices")
    with gzip.open(os.path.join(GOLDEN, "cari.mtx.gz"), "rb") as f, open(d / "matrices" / "cari.mtx", "wb") as g:
        shutil.copyfileobj(f, g)
    return str(d)


def run(args, cwd):
    r = subprocess.run([BIN] + args, cwd=cwd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_cari_run_prints_the_reference_skeleton_and_the_right_rows(workdir):
    import spada_sim_amd as S
    out = run(["accuratesimu", "spada", "ss", "cari", CFG, "--output", "C.mtx"], workdir)
    lines = out.splitlines()
    assert lines[0] == CFG and lines[1] == "---- Python Interface ----" and lines[2] == "% Load cari from ./matrices"
    assert lines[3] == "Get GEMM cari" and lines[4] == "---- cari ----" and lines[5] == "--A: (400, 1200)"
    assert "--B: (1200, 400)" in lines and "Avg row len of A: 382, Avg row len of B: 127" in lines
    k = lines.index("-----Result-----")
    assert lines[k + 1] == "-----Access count" and lines[k + 2].startswith("Execution count: ")
    for j, name in enumerate(["A matrix", "B matrix", "C matrix", "Cache"]):
        assert re.fullmatch(name + r" count: read \d+ write \d+", lines[k + 3 + j]), lines[k + 3 + j]
    assert lines[k + 7] == "-----Output product matrix"
    rows = lines[k + 8:k + 18]
    g = np.load(os.path.join(GOLDEN, "cari_product.npz"))
    for r, line in enumerate(rows):           # storage.rs:115-126: rowptr, first five column indices, first five values
        mo = re.fullmatch(r"rowptr: (\d+) indptr: \[(.*)\] data: \[(.*)\]", line)
        assert mo and int(mo.group(1)) == r and mo.group(2) == "0, 1, 2, 3, 4"
        vals = np.array([float(x) for x in mo.group(3).split(", ")])
        if r in (0, 9):
            assert np.allclose(vals, g["row0_head" if r == 0 else "row9_head"], rtol=RTOL, atol=0)
    # the product on disk against the oracle (A * A^T: gemm.rs:41-53)
    c = S.load_mm_mat(workdir, "C")
    a = S.load_mm_mat(os.path.join(workdir, "matrices"), "cari")
    ao = to_oracle(a)
    bo = to_oracle(a.transpose())
    ref = oracle.spgemm_spa(ao, bo)
    assert_parity(c, ref, ao, bo, RTOL)
    # checksum line: printed, embedded in the file, equal to the library's own
    cs_line = lines[lines.index("-----Checksum of the product matrix") + 1]
    assert open(os.path.join(workdir, "C.mtx")).read().splitlines()[1] == "% spada-sim checksum: " + cs_line
    assert S.checksum(c)[1] == cs_line and "rows 400 cols 400 nnz 160000 " in cs_line


def test_preprocess_and_accumulators_give_the_same_product(workdir):
    """-p (rows of A sorted by length on the GPU, result mapped back: main.rs:60-63, simulator.rs:1039-1055), its product-aware
    variant and the binary dump.  With the sort-merge accumulator every run adds in ascending k, so the checksum lines --
    structure AND value hashes -- are identical and equal to the oracle's."""
    import spada_sim_amd as S
    a = S.load_mm_mat(os.path.join(workdir, "matrices"), "cari")
    ref = oracle.spgemm_sortmerge(to_oracle(a), to_oracle(a.transpose()))
    ref_line = S.checksum(S.CsMat((ref.rows, ref.cols), ref.indptr, ref.indices, ref.data))[1]
    base = ["accuratesimu", "spada", "ss", "cari", CFG, "--accumulator", "sort_merge"]
    seen = []
    for extra in ([], ["-p"], ["--preprocess-by", "products"], ["--preprocess", "--output", "Cp.bin"]):
        out = run(base + extra + (["--checksum"] if "--output" not in extra else []), workdir).splitlines()
        seen.append(out[out.index("-----Checksum of the product matrix") + 1])
    assert seen == [ref_line] * 4
    c = S.read_bin(os.path.join(workdir, "Cp.bin"))
    assert np.array_equal(c.indptr, ref.indptr) and np.array_equal(c.indices, ref.indices) and np.array_equal(c.data, ref.data)
    # hash accumulator + -p: same structure hash, values within tolerance
    out = run(["accuratesimu", "spada", "ss", "cari", CFG, "-p", "--checksum"], workdir).splitlines()
    line = out[out.index("-----Checksum of the product matrix") + 1]
    assert line.split(" values ")[0] == ref_line.split(" values ")[0]


def test_reorder_api_leaves_c_unchanged(engine):
    """spada_dev_csr_reorder / spada_dev_unpermute_c through the host-pointer calls: the row map is the stable ascending order of
    the key (sort_by_length, preprocessing.rs:76-89) and C is the same matrix with and without the pre-pass."""
    import spada_sim_amd as S
    m = S.generate(S.GEN_WEBBASE_LIKE, 30000, 95000, 12)
    ao = to_oracle(m)
    ref = oracle.spgemm_spa(ao, ao)
    lens = np.diff(m.indptr.astype(np.int64))
    prods = np.array([lens[m.indices[int(m.indptr[r]):int(m.indptr[r + 1])].astype(np.int64)].sum() for r in range(m.shape[0])])
    for key, k in ((S.REORDER_BY_LENGTH, lens), (S.REORDER_BY_PRODUCTS, prods)):
        c = engine.spgemm(m, m, reorder=key)
        assert np.array_equal(engine.last_rowmap.astype(np.int64), np.argsort(k, kind="stable"))
        assert_parity(c, ref, ao, ao, RTOL)
    eng = S.Engine(accumulator=S.ACC_SORT_MERGE)
    try:
        c0 = eng.spgemm(m, m)
        c1 = eng.spgemm(m, m, reorder=S.REORDER_BY_LENGTH)
        assert np.array_equal(c0.indptr, c1.indptr) and np.array_equal(c0.indices, c1.indices) and np.array_equal(c0.data, c1.data)
    finally:
        eng.close()
    # the Simulator mirror: reorder_row(sort_by_length(..)) as main.rs:60-63 does
    dram_a, dram_b = S.CsrMatStorage.init_with_gemm(S.GEMM.from_mat("w", m))
    dram_a.reorder_row(S.sort_by_length(dram_a))
    sim = S.Simulator(2, 16, 8, 1572864, 8, len(dram_b.indptr), [1, 10000000], dram_a, dram_b, None, "Spada", 30, 0, 1.0, 16, 8.0,
                      engine=engine)
    sim.execute()
    assert_parity(sim.result_matrix(), ref, ao, ao, RTOL)
    assert dram_a.remapped and [dram_a.row_remap[i] for i in range(5)] == list(np.argsort(lens, kind="stable")[:5])


def test_bench_exercises_the_native_exchange_on_one_rank():
    """`bench.py --exercise-exchange`: both forms of libspada_comm.so's RCCL exchange on a one-rank communicator, checked against the
    plain one-pass product -- the code path the driver's multi-GPU runs take.  In the file that runs first, as a process of its own:
    the native exchange is exercised on the box even when a later test file stops the run."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--exercise-exchange", "--workload", "mc2depi", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    found = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and "exercise_exchange" in ln]
    assert found, (r.stdout[-1000:], r.stderr[-1000:])
    line = json.loads(found[-1])
    assert line["exercise_exchange"] == "ok" and line["nnz_c"] > 0 and line["modes"] == ["overlap", "after"]
