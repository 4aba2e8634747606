"""CPU: the host half of libspada_spgemm.so (no GPU needed) -- MatrixMarket ingest against the
reference loader's outputs, GEMM::from_mat, partitioning, configuration, C-ABI surface."""
import ctypes
import hashlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, to_oracle
from oracle import oracle

import spada_sim_amd as S
from spada_sim_amd import _ffi

MTX = ["general_unsorted_dups", "symmetric", "skew", "pattern_sym", "integer_rect", "pattern_general"]


def test_library_exports_every_symbol_of_the_header():
    """Every function include/spada_ffi.h and include/spada_cycle.h declare is exported, and nothing is missing from the binding."""
    declared = set()
    for name in ("spada_ffi.h", "spada_cycle.h"):
        hdr = open(os.path.join(ROOT, "include", name)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        declared |= set(re.findall(r"\b(spada_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"spada_options", "spada_stats", "spada_config"}
    assert len(declared) >= 30 and {"spada_cycle_create", "spada_cycle_execute", "spada_cycle_get_counts", "spada_cycle_get_result",
                                    "spada_cycle_destroy"} <= declared
    L = ctypes.CDLL(_ffi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/*.h but not exported"
    assert declared == set(_ffi.SIGNATURES), declared ^ set(_ffi.SIGNATURES)
    assert _ffi.lib().spada_abi_version() == 5


def test_probe_header_symbols_are_exported():
    """include/spada_probe.h (measurement entry points: not part of the drop-in boundary, not in the binding) against libspada_spgemm.so."""
    hdr = open(os.path.join(ROOT, "include", "spada_probe.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(spada_[a-z0-9_]+)\s*\(", hdr))
    assert declared == {"spada_dev_probe_floor", "spada_dev_csr_aux_cost", "spada_dev_scratch_placement"}, declared
    L = ctypes.CDLL(_ffi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/spada_probe.h but not exported"
    assert not (declared & set(_ffi.SIGNATURES))      # (the product binding does not carry them)


def test_comm_library_exports_every_symbol_of_its_header():
    """include/spada_comm.h (RCCL exchange) against libspada_comm.so and the binding; no collective is called here."""
    hdr = open(os.path.join(ROOT, "include", "spada_comm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(spada_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) == 10 and "spada_comm_plan" in declared
    L = ctypes.CDLL(_ffi.COMM_LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in spada_comm.h but not exported"
    assert declared == set(_ffi.COMM_SIGNATURES), declared ^ set(_ffi.COMM_SIGNATURES)


@pytest.mark.parametrize("nranks", [2, 3, 8])
@pytest.mark.parametrize("chunks", [0, 1, 4])
def test_comm_plan_places_every_segment(nranks, chunks):
    """spada_comm_plan -- the offset arithmetic both native exchange forms (spada_comm_allgatherv_c, spada_dist_spgemm_numeric) use,
    as a pure host function: for N = 2, 3, 8 with empty row blocks, blocks without entries and empty pieces, an exchange simulated with
    its offsets (every rank copies the root's segment at the planned position: what the in-place ncclBroadcast does) leaves the same
    whole C on every rank as the torch.distributed path builds (spada_sim_amd/parallel.py: concatenation + cumulative row lengths).
    No N > 1 RCCL run exists yet; this is the part of it that needs no GPU."""
    rng = np.random.default_rng(100 * nranks + chunks)
    for trial in range(20):
        rows = rng.integers(0, 40, nranks).astype(np.uint64)
        rows[rng.integers(0, nranks)] = 0                       # an empty row block
        lens = [rng.integers(0, 9, int(r)) for r in rows]
        if trial % 3 == 0:
            lens[int(rng.integers(0, nranks))][:] = 0            # a block whose rows are all empty
        nnz = np.array([int(l.sum()) for l in lens], np.uint64)
        blocks = []
        for r in range(nranks):
            ptr = np.concatenate([[0], np.cumsum(lens[r])]).astype(np.uint64)
            blocks.append((ptr, rng.integers(0, 1000, int(nnz[r])).astype(np.uint32), rng.uniform(0, 1, int(nnz[r]))))
        pos = None
        if chunks:
            pos = np.zeros((nranks, chunks + 1), np.uint64)
            for r in range(nranks):
                cuts = np.sort(rng.integers(0, int(nnz[r]) + 1, chunks - 1)) if chunks > 1 else np.zeros(0, np.int64)
                if trial % 2 == 0 and chunks > 1:
                    cuts[0] = 0                                  # an empty first piece
                pos[r] = np.concatenate([[0], np.sort(cuts), [int(nnz[r])]])
        row_off, nnz_off, pb, pc = S.comm_plan(rows, nnz, pos)
        # the torch.distributed path's layout (parallel.allgatherv_c): concatenation, indptr = cumulative sum of the row lengths
        exp_idx = np.concatenate([b[1] for b in blocks])
        exp_val = np.concatenate([b[2] for b in blocks])
        exp_ptr = np.concatenate([[0], np.cumsum(np.concatenate(lens))]).astype(np.uint64)
        assert row_off[-1] == rows.sum() and nnz_off[-1] == nnz.sum()
        assert np.array_equal(nnz_off[:-1], np.concatenate([[0], np.cumsum(nnz)[:-1]]).astype(np.uint64))
        # simulated exchange on every rank
        for me in range(nranks):
            fi = np.full(int(nnz_off[-1]), 0xFFFFFFFF, np.uint32)
            fv = np.full(int(nnz_off[-1]), np.nan)
            fp = np.full(int(row_off[-1]) + 1, 0xFFFFFFFFFFFFFFFF, np.uint64)
            if chunks:
                for k in range(chunks):
                    for r in range(nranks):
                        b, n = int(pb[r, k]), int(pc[r, k])
                        lo = int(pos[r, k])
                        fi[b:b + n] = blocks[r][1][lo:lo + n]
                        fv[b:b + n] = blocks[r][2][lo:lo + n]
            else:
                for r in range(nranks):
                    o, n = int(nnz_off[r]), int(nnz[r])
                    fi[o:o + n] = blocks[r][1]
                    fv[o:o + n] = blocks[r][2]
            fp[0] = 0                                            # gather_indptr: entries 1 .. rows_r of every block, shifted
            for r in range(nranks):
                o, n = int(row_off[r]), int(rows[r])
                fp[o + 1:o + 1 + n] = blocks[r][0][1:] + nnz_off[r]
            assert np.array_equal(fi, exp_idx) and np.array_equal(fv, exp_val) and np.array_equal(fp, exp_ptr), (me, trial)
    # malformed piece positions are refused, not trusted
    bad = np.array([[0, 5, 3, 7]], np.uint64)
    with pytest.raises(S.SpadaError):
        S.comm_plan(np.array([4], np.uint64), np.array([7], np.uint64), bad)
    with pytest.raises(S.SpadaError):
        S.comm_plan(np.array([4], np.uint64), np.array([9], np.uint64), np.array([[0, 2, 7]], np.uint64))


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(_ffi.CsrView) == 48
    assert ctypes.sizeof(_ffi.Options) == 16
    assert ctypes.sizeof(_ffi.Stats) == 7 * 8 + 7 * 8 + 2 * 8 * 8 + 7 * 8 + 3 * 8
    # and against the C compiler's view of include/spada_ffi.h
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "sz.c")
        with open(src, "w") as f:
            f.write('#include <stdio.h>\n#include "spada_ffi.h"\n#include "spada_cycle.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n", '
                    'sizeof(spada_csr_view), sizeof(spada_options), sizeof(spada_stats), sizeof(spada_config), '
                    'sizeof(spada_cycle_config), sizeof(spada_cycle_counts));return 0;}\n')
        exe = os.path.join(d, "sz")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(_ffi.CsrView), ctypes.sizeof(_ffi.Options), ctypes.sizeof(_ffi.Stats),
                     ctypes.sizeof(_ffi.Config), ctypes.sizeof(_ffi.CycleConfig), ctypes.sizeof(_ffi.CycleCounts)]


def test_cari_loader_pins(matrices_dir):
    """Outputs of the reference's embedded Python loader on cari.mtx (py2rust.rs:64-79)."""
    g = np.load(os.path.join(GOLDEN, "cari_loader.npz"))
    m = S.load_mm_mat(matrices_dir, "cari")
    assert m.shape == tuple(int(x) for x in g["shape"]) == (400, 1200)
    assert m.nnz() == int(g["nnz"]) == 152800
    assert np.array_equal(m.indptr, g["indptr"])
    assert hashlib.sha256(m.indices.tobytes()).hexdigest() == str(g["indices_sha256"])
    assert hashlib.sha256(m.data.tobytes()).hexdigest() == str(g["data_sha256"])   # bit-exact value parsing
    assert np.array_equal(m.indices[:16], g["indices_head"]) and np.array_equal(m.data[:16], g["data_head"])
    m.validate()


@pytest.mark.parametrize("name", MTX)
def test_mtx_variants_match_reference_loader(name):
    g = np.load(os.path.join(GOLDEN, "mtx_loader.npz"))
    m = S.load_mm_mat(os.path.join(GOLDEN, "mtx"), name)
    assert m.shape == tuple(int(x) for x in g[name + "_shape"])
    assert np.array_equal(m.indptr, g[name + "_indptr"])
    assert np.array_equal(m.indices, g[name + "_indices"])
    assert np.array_equal(m.data, g[name + "_data"])


def test_mtx_errors(tmp_path):
    with pytest.raises(S.SpadaError) as e:
        S.load_mm_mat(str(tmp_path), "missing")
    assert e.value.code == 5
    (tmp_path / "dense.mtx").write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    with pytest.raises(S.SpadaError) as e:
        S.load_mm_mat(str(tmp_path), "dense")
    assert e.value.code == 8
    (tmp_path / "short.mtx").write_text("%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1.0\n")
    with pytest.raises(S.SpadaError) as e:
        S.load_mm_mat(str(tmp_path), "short")
    assert e.value.code == 6
    (tmp_path / "oob.mtx").write_text("%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n")
    with pytest.raises(S.SpadaError):
        S.load_mm_mat(str(tmp_path), "oob")


def test_mtx_write_read_round_trip(tmp_path):
    m = S.generate(S.GEN_UNIFORM, 200, 5, 3)
    S.write_mm_mat(tmp_path / "rt.mtx", m)
    r = S.load_mm_mat(str(tmp_path), "rt")
    assert r.shape == m.shape and np.array_equal(r.indptr, m.indptr) and np.array_equal(r.indices, m.indices)
    assert np.array_equal(r.data, m.data)   # %.17g round-trips f64


def test_from_mat_rule_and_transpose(matrices_dir):
    m = S.load_mm_mat(matrices_dir, "cari")
    g = S.GEMM.from_mat("cari", m)
    assert g.a is m and g.b.shape == (1200, 400)          # non-square => A * A^T (gemm.rs:45-47)
    ref = oracle.transpose(to_oracle(m))
    assert np.array_equal(g.b.indptr, ref.indptr) and np.array_equal(g.b.indices, ref.indices)
    assert np.array_equal(g.b.data, ref.data)
    sq = S.generate(S.GEN_UNIFORM, 64, 4, 1)
    assert S.GEMM.from_mat("sq", sq).b is sq              # square => A * A (gemm.rs:43-44)
    a, b = S.CsrMatStorage.init_with_gemm(g)
    assert a.mat_shape == [1200, 400] and b.mat_shape == [400, 1200]   # [cols, rows], storage.rs:225
    assert a.row_num() == 400 and a.read_row(3).len() == 382


def test_validate_rejects_malformed_csr():
    bad = S.CsMat((2, 2), np.array([0, 2, 2], np.uint64), np.array([1, 0], np.uint64), np.array([1.0, 2.0]))
    with pytest.raises(S.SpadaError):
        bad.validate()
    bad = S.CsMat((2, 2), np.array([0, 1, 2], np.uint64), np.array([0, 5], np.uint64), np.array([1.0, 2.0]))
    with pytest.raises(S.SpadaError):
        bad.validate()


def test_partition_rows_balances_products():
    m = S.generate(S.GEN_RMAT, 12, 8, 5)
    total = S.count_products(m, m)
    assert total == oracle.count_products(to_oracle(m), to_oracle(m))
    for nparts in (1, 2, 3, 8):
        b = S.partition_rows(m, m, nparts)
        assert b[0] == 0 and b[-1] == m.shape[0] and all(x <= y for x, y in zip(b, b[1:]))
        parts = [S.count_products(m, m, r0, r1) for r0, r1 in zip(b[:-1], b[1:])]
        assert sum(parts) == total
        if nparts > 1:
            heaviest_row = max(S.count_products(m, m, r, r + 1) for r in range(0, m.shape[0], 97))
            assert max(parts) <= total / nparts + max(heaviest_row, total // 50) + m.shape[0]


def test_generators_are_deterministic_and_canonical():
    for kind, p0, p1 in [(S.GEN_RMAT, 10, 8), (S.GEN_WEBBASE_LIKE, 5000, 15000), (S.GEN_COP20K_LIKE, 3000, 0),
                         (S.GEN_CAGE12_LIKE, 3000, 0), (S.GEN_MC2DEPI_LIKE, 779 * 4, 0), (S.GEN_UNIFORM, 500, 7)]:
        a = S.generate(kind, p0, p1, 99)
        b = S.generate(kind, p0, p1, 99)
        a.validate()
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data)
        assert a.data.min() >= 0.1          # uniform(0.1, 1.0), duplicates summed
        c = S.generate(kind, p0, p1, 100)
        if kind != S.GEN_MC2DEPI_LIKE:      # the stencil's pattern does not depend on the seed
            assert not np.array_equal(a.indices, c.indices) or not np.array_equal(a.indptr, c.indptr)


# literature counts of the SuiteSparse inputs (rows, nnz(A), products, nnz(C))
LITERATURE = {
    "webbase-1M": (S.GEN_WEBBASE_LIKE, 12347, 1000005, 3105536, 69.5e6, 51.1e6),
    "cop20k_A": (S.GEN_COP20K_LIKE, 12346, 121192, 2624331, 79.9e6, 18.7e6),
    "cage12": (S.GEN_CAGE12_LIKE, 12348, 130228, 2032536, 34.6e6, 15.2e6),
    "mc2depi": (S.GEN_MC2DEPI_LIKE, 12349, 525825, 2100225, 8.4e6, 5.2e6),
}


@pytest.mark.parametrize("name", list(LITERATURE))
def test_surrogates_match_the_literature_counts(name):
    """The bench surrogates (same kind / seed as bench.py) reproduce rows exactly and nnz(A), products and nnz(C) of
    A*A within 5 % of the literature rows of SURVEY.md section 8 -- counted by the oracle."""
    kind, seed, rows, nnz_t, prod_t, nnzc_t = LITERATURE[name]
    a = S.generate(kind, 0, 0, seed)
    assert a.shape == (rows, rows)
    ao = to_oracle(a)
    prod = oracle.count_products(ao, ao)
    nnzc = oracle.spgemm_spa(ao, ao, n_threads=oracle.num_threads()).nnz
    for got, want, what in [(a.nnz(), nnz_t, "nnz(A)"), (prod, prod_t, "products"), (nnzc, nnzc_t, "nnz(C)")]:
        assert abs(got / want - 1.0) <= 0.05, f"{name}: {what} = {got}, literature {want:.0f}"


def test_parse_config(tmp_path):
    cfg = S.parse_config(os.path.join(ROOT, "config", "config_1mb_row1.json"))
    assert cfg["ss_filepath"] == "./matrices" and cfg["pe_num"] == 2 and cfg["at_num"] == 16 and cfg["lane_num"] == 8
    assert cfg["cache_size"] == 1572864 and cfg["word_byte"] == 8 and cfg["block_shape"] == [1, 10000000]
    assert cfg["mem_latency"] == 30 and cfg["channel"] == 16 and cfg["bandwidth_per_channel"] == 8.0 and cfg["freq"] == 1.0
    base = json.load(open(os.path.join(ROOT, "config", "config_1mb_row1.json")))
    assert "gpus" not in base          # the binary drives one GPU: the shipped configuration does not pretend otherwise
    for k in ("accumulator", "repeat"):
        base.pop(k)
    p = tmp_path / "min.json"
    p.write_text(json.dumps(base))
    assert S.parse_config(p)["accumulator"] == 0      # optional engine keys default
    for missing in ("lane_num", "block_shape", "ss_filepath"):
        d = dict(base)
        d.pop(missing)
        p.write_text(json.dumps(d))
        with pytest.raises(S.SpadaError) as e:       # serde: "missing field `x`"
            S.parse_config(p)
        assert e.value.code == 6 and missing in str(e.value)
    p.write_text("{ not json")
    with pytest.raises(S.SpadaError):
        S.parse_config(p)


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a gfx950 device the engine refuses to exist; nothing computes on the CPU."""
    if S.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(S.SpadaError) as e:
        S.Engine()
    assert e.value.code == 2
    L = _ffi.lib()
    nnz = ctypes.c_uint64(0)
    m = S.generate(S.GEN_UNIFORM, 10, 2, 1)
    rc = L.spada_spgemm_symbolic(None, ctypes.byref(m.view()), ctypes.byref(m.view()), ctypes.byref(nnz))
    assert rc == 7    # SPADA_ERR_STATE


def test_product_path_does_not_touch_the_oracle():
    """The package and the native sources never import / link the oracle."""
    for d, _, files in os.walk(os.path.join(ROOT, "spada_sim_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", "Makefile")):
                text = open(os.path.join(d, f), errors="replace").read()
                assert "oracle" not in text.replace("oracle_rowwise", ""), f"{f} mentions the oracle"


def test_checksum_and_binary_dump_round_trip(tmp_path):
    """C writer: binary CSR dump + checksum line; the .mtx writer embeds the same line as a comment."""
    m = S.generate(S.GEN_RMAT, 9, 6, 3)
    cs, line = S.checksum(m)
    assert cs["rows"] == m.shape[0] and cs["nnz"] == m.nnz() and f"nnz {m.nnz()} " in line
    assert abs(cs["value_sum"] - float(np.sum(m.data))) <= 1e-9 * cs["value_abs_sum"]
    p = tmp_path / "m.bin"
    S.write_bin(p, m)
    assert p.stat().st_size == 8 + 4 * 8 + (m.shape[0] + 1) * 8 + m.nnz() * 16 + 16
    r = S.read_bin(p)
    assert r.shape == m.shape and np.array_equal(r.indptr, m.indptr) and np.array_equal(r.indices, m.indices)
    assert np.array_equal(r.data, m.data) and S.checksum(r)[1] == line
    raw = bytearray(p.read_bytes())
    raw[8 + 32 + (m.shape[0] + 1) * 8 + 3] ^= 1          # flip one bit of a column index
    p.write_bytes(bytes(raw))
    with pytest.raises(S.SpadaError) as e:
        S.read_bin(p)
    assert e.value.code == 6
    raw[8 + 32 + (m.shape[0] + 1) * 8 + 3] ^= 1
    raw[-16 - 5] ^= 4                                      # and one bit of the last value: structure fine, checksum not
    p.write_bytes(bytes(raw))
    with pytest.raises(S.SpadaError) as e:
        S.read_bin(p)
    assert e.value.code == 6 and "checksum mismatch" in str(e.value)
    (tmp_path / "short.bin").write_bytes(bytes(raw[:100]))
    with pytest.raises(S.SpadaError):
        S.read_bin(tmp_path / "short.bin")
    q = tmp_path / "m.mtx"
    S.write_mm_mat(q, m)
    text = q.read_text().splitlines()
    assert text[1] == "% spada-sim checksum: " + line
    back = S.load_mm_mat(str(tmp_path), "m")
    assert S.checksum(back)[1] == line                   # %.17g round-trips every value
    m2 = S.CsMat(m.shape, m.indptr, m.indices, m.data.copy())
    m2.data[5] = np.nextafter(m2.data[5], 2.0)
    assert S.checksum(m2)[0]["value_hash"] != cs["value_hash"] and S.checksum(m2)[0]["structure_hash"] == cs["structure_hash"]


def test_hostile_sizes_are_refused_not_allocated(tmp_path):
    """Sizes read from files are not trusted: a 40-byte binary dump that announces 2^40 rows and 2^44 entries, and a MatrixMarket size
    line that announces 2^60 entries, come back as error codes -- no allocation is attempted from them and no C++ exception leaves
    the C ABI (the process would abort)."""
    import struct
    p = tmp_path / "huge.bin"
    p.write_bytes(b"SPADACSR" + struct.pack("<4Q", 1, 1 << 40, 10, 1 << 44))
    with pytest.raises(S.SpadaError) as e:
        S.read_bin(p)
    assert e.value.code == 6
    p.write_bytes(b"SPADACSR" + struct.pack("<4Q", 1, 3, 10, 1 << 43) + b"\0" * 64)     # inside the bounds, still not the file's size
    with pytest.raises(S.SpadaError) as e:
        S.read_bin(p)
    assert e.value.code == 6
    (tmp_path / "big.mtx").write_text("%%MatrixMarket matrix coordinate real general\n10 10 1152921504606846976\n1 1 1.0\n")
    with pytest.raises(S.SpadaError) as e:
        S.load_mm_mat(str(tmp_path), "big")
    assert e.value.code == 6 and "announces" in str(e.value)


BIN = os.environ.get("SPADA_BIN_PATH") or os.path.join(ROOT, "spada_sim_amd", "bin", "spada-sim")   # (env: the sanitizer build)
CFG = os.path.join(ROOT, "config", "config_1mb_row1.json")


def _run_cli(args, cwd):
    import subprocess
    return subprocess.run([BIN] + args, cwd=cwd, capture_output=True, text=True, timeout=300)


def test_cli_usage_and_exit_codes(matrices_dir):
    """Exit codes of the drop-in binary: 1 = usage error (clap), 101 = run-time failure (a Rust panic's code; main.rs:119 for the
    unimplemented simulators, py2rust.rs for NN workloads)."""
    cwd = os.path.dirname(matrices_dir)
    os.makedirs(os.path.join(cwd, "matrices"), exist_ok=True)
    import shutil
    shutil.copy(os.path.join(matrices_dir, "cari.mtx"), os.path.join(cwd, "matrices", "cari.mtx"))
    r = _run_cli([], cwd)
    assert r.returncode == 1 and "USAGE" in r.stderr
    r = _run_cli(["accuratesimu", "gpu", "ss", "cari", CFG], cwd)
    assert r.returncode == 1 and "isn't a valid value for '<accelerator>'" in r.stderr
    r = _run_cli(["accuratesimu", "spada", "ss", "cari", CFG, "--bogus"], cwd)
    assert r.returncode == 1 and "wasn't expected" in r.stderr
    r = _run_cli(["accuratesimu", "spada", "ss", "cari", CFG, "--preprocess-by", "width"], cwd)
    assert r.returncode == 1
    r = _run_cli(["accuratesimu", "spada", "nn", "alexnet", CFG], cwd)
    assert r.returncode == 101 and "NN" in r.stderr
    r = _run_cli(["trafficmodel", "spada", "ss", "cari", CFG], cwd)
    assert r.returncode == 101 and "Unimplemented simulator" in r.stderr
    assert r.stdout.startswith(CFG + "\n---- Python Interface ----\n% Load cari from ./matrices\nGet GEMM cari\n---- cari ----\n--A: (400, 1200)\n")   # frontend.rs:79 prints the configuration path first
    assert "Avg row len of A: 382, Avg row len of B: 127" in r.stdout
    # `gpus` > 1 in a configuration is refused (the sharded path is libspada_comm.so's, one process per GPU), not ignored
    cfg2 = os.path.join(cwd, "two_gpus.json")
    d = json.load(open(CFG))
    d["gpus"] = 2
    json.dump(d, open(cfg2, "w"))
    r = _run_cli(["accuratesimu", "spada", "ss", "cari", cfg2], cwd)
    assert r.returncode == 101 and "`gpus` = 2" in r.stderr and "-----Result-----" not in r.stdout
    # SPADA_TRACE=1: the trace lines that replace the reference's trace_exec feature go to stderr, stdout keeps the skeleton
    r = subprocess.run([BIN, "trafficmodel", "spada", "ss", "cari", CFG], cwd=cwd, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, SPADA_TRACE="1"))
    assert "[spada trace 1] load_mm_mat" in r.stderr and "400 x 1200, 152800 entries" in r.stderr and "spada trace" not in r.stdout
    r = _run_cli(["accuratesimu", "spada", "ss", "missing", CFG], cwd)
    assert r.returncode == 101 and "missing.mtx" in r.stderr
    if S.device_count() == 0:      # no GPU: the run itself must fail loudly, not fall back to anything
        r = _run_cli(["accuratesimu", "spada", "ss", "cari", CFG], cwd)
        assert r.returncode == 101 and "no HIP device" in r.stderr and "-----Result-----" not in r.stdout


def _mock_lib():
    """lib/libspada_comm_mock.so: the host half of spada_comm.hip compiled against test doubles of HIP, RCCL and the engine (csrc/mock/)."""
    path = os.environ.get("SPADA_COMM_MOCK_LIB_PATH") or os.path.join(os.path.dirname(_ffi.LIB_PATH), "libspada_comm_mock.so")
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not built (make -C spada_sim_amd/csrc all)")
    _ffi.lib()                                       # (libspada_spgemm.so first: spada_last_error / fail live there)
    L = ctypes.CDLL(path)
    u64p, vp = ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p
    L.spada_comm_create.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
    L.spada_comm_destroy.argtypes = [vp]
    L.spada_comm_destroy.restype = None
    L.spada_comm_allgatherv_c.argtypes = [vp, vp, vp, vp, u64p, u64p, vp, vp, vp]
    L.spada_dist_spgemm_symbolic.argtypes = [vp, vp, vp, vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, u64p, u64p]
    L.spada_dist_spgemm_numeric.argtypes = [vp, vp, vp, vp, vp]
    L.spada_mock_begin.argtypes = [ctypes.c_int]
    L.spada_mock_set_gathered.argtypes = [u64p, ctypes.c_uint64]
    L.spada_mock_set_block.argtypes = [ctypes.c_uint64, u64p, ctypes.c_uint32, u64p, ctypes.c_uint64, ctypes.c_uint32]
    L.spada_mock_set_bases.argtypes = [vp, vp, vp]
    L.spada_mock_log_size.restype = ctypes.c_uint64
    L.spada_mock_log_get.argtypes = [ctypes.c_uint64, u64p]
    for f in (L.spada_mock_begin, L.spada_mock_reset, L.spada_mock_set_gathered, L.spada_mock_set_block, L.spada_mock_set_bases, L.spada_mock_log_get):
        f.restype = None
    return L


@pytest.mark.parametrize("form", ["after", "overlap"])
@pytest.mark.parametrize("nranks", [2, 8])
def test_exchange_call_sequence_on_mock_ranks(form, nranks):
    """Both native exchange forms (spada_comm_allgatherv_c after a one-pass product; spada_dist_spgemm_symbolic / _numeric with the
    pieces broadcast while the next is computed) for N = 2 and 8 ranks -- with empty row blocks, blocks without entries and empty
    pieces -- on lib/libspada_comm_mock.so: spada_comm.hip's own host code against test doubles of the HIP / RCCL / engine calls it
    makes (csrc/mock/).  Checked: EVERY RANK POSTS THE SAME COLLECTIVES IN THE SAME ORDER (operation, root, count, type, offset in
    the whole C, group boundaries) -- the condition under which RCCL's grouped broadcasts match, including ranks whose own piece is
    empty -- and, with the broadcasts replayed between the ranks' buffers, every rank ends with the whole C and its indptr.
    No N > 1 RCCL run exists (one GPU per box); this is the host logic of that run, on the CPU."""
    L = _mock_lib()
    u64p = ctypes.POINTER(ctypes.c_uint64)
    rng = np.random.default_rng(7 * nranks + len(form))
    chunks = 4
    for trial in range(6):
        rows = rng.integers(1, 30, nranks).astype(np.uint64)
        empty = rng.choice(nranks, 2 if nranks > 2 else 1, replace=False)
        rows[empty[0]] = 0                                        # an empty row block
        lens = [rng.integers(0, 7, int(r)) for r in rows]
        if len(empty) > 1:
            lens[int(empty[1])][:] = 0                            # a block whose rows are all empty
        nnz = np.array([int(l.sum()) for l in lens], np.uint64)
        ptrs = [np.concatenate([[0], np.cumsum(l)]).astype(np.uint64) for l in lens]
        pos = np.zeros((nranks, chunks + 1), np.uint64)
        for r in range(nranks):
            cuts = np.sort(rng.integers(0, int(nnz[r]) + 1, chunks - 1))
            if trial % 2 == 0:
                cuts[0] = 0                                       # an empty first piece
            pos[r] = np.concatenate([[0], cuts, [int(nnz[r])]])
        fill = [100000 * (r + 1) for r in range(nranks)]
        # what the mock engine writes for rank r's entry i: index fill + i, value (fill + i) / 2 (comm_mock.cpp)
        blocks = [((fill[r] + np.arange(int(nnz[r]))).astype(np.uint32), (fill[r] + np.arange(int(nnz[r]))) * 0.5) for r in range(nranks)]
        tot_rows, tot_nnz = int(rows.sum()), int(nnz.sum())
        exp_idx = np.concatenate([b[0] for b in blocks]) if tot_nnz else np.zeros(0, np.uint32)
        exp_val = np.concatenate([b[1] for b in blocks]) if tot_nnz else np.zeros(0)
        exp_ptr = np.concatenate([[0], np.cumsum(np.concatenate(lens))]).astype(np.uint64)
        gathered = np.concatenate([np.concatenate([[rows[r], nnz[r]], pos[r]]) for r in range(nranks)]).astype(np.uint64)
        logs, finals = None, []
        for pas in (1, 2):
            L.spada_mock_begin(pas)
            logs, finals = [], []
            for me in range(nranks):
                L.spada_mock_reset()
                ci = np.full(tot_nnz + 1, 0xFFFFFFFF, np.uint32)
                cv = np.full(tot_nnz + 1, np.nan)
                cp = np.full(tot_rows + 1, 0xFFFFFFFFFFFFFFFF, np.uint64)
                L.spada_mock_set_bases(ci.ctypes.data, cv.ctypes.data, cp.ctypes.data)
                comm = ctypes.c_void_p()
                ident = (ctypes.c_char * 128)()
                assert L.spada_comm_create(ident, me, nranks, 0, ctypes.byref(comm)) == 0
                if form == "after":
                    my_i, my_v = blocks[me][0].copy(), blocks[me][1].copy()
                    assert L.spada_comm_allgatherv_c(comm, ptrs[me].ctypes.data, my_i.ctypes.data, my_v.ctypes.data, rows.ctypes.data_as(u64p),
                                                     nnz.ctypes.data_as(u64p), cp.ctypes.data, ci.ctypes.data, cv.ctypes.data) == 0, S.last_error()
                else:
                    L.spada_mock_set_block(int(nnz[me]), pos[me].ctypes.data_as(u64p), chunks, ptrs[me].ctypes.data_as(u64p), int(rows[me]), fill[me])
                    L.spada_mock_set_gathered(gathered.ctypes.data_as(u64p), len(gathered))
                    r_out, n_out = np.zeros(nranks, np.uint64), np.zeros(nranks, np.uint64)
                    assert L.spada_dist_spgemm_symbolic(ctypes.c_void_p(1), comm, None, None, 0, int(rows[me]), chunks, r_out.ctypes.data_as(u64p),
                                                        n_out.ctypes.data_as(u64p)) == 0, S.last_error()
                    assert np.array_equal(r_out, rows) and np.array_equal(n_out, nnz)
                    assert L.spada_dist_spgemm_numeric(ctypes.c_void_p(1), comm, cp.ctypes.data, ci.ctypes.data, cv.ctypes.data) == 0, S.last_error()
                L.spada_comm_destroy(comm)
                log = []
                rec = (ctypes.c_uint64 * 5)()
                for i in range(L.spada_mock_log_size()):
                    L.spada_mock_log_get(i, rec)
                    log.append(tuple(rec))
                logs.append(log)
                finals.append((ci[:tot_nnz].copy(), cv[:tot_nnz].copy(), cp.copy()))
            # every rank posts the same sequence: operations, roots, counts, types, places, group boundaries
            for me in range(1, nranks):
                assert logs[me] == logs[0], (form, nranks, trial, me)
        ops = [c[0] for c in logs[0]]
        assert ops.count(1) == ops.count(2) and ops.count(3) > 0
        assert all(c[4] != 0xFFFFFFFFFFFFFFFF for c in logs[0] if c[0] == 3)           # every broadcast lies inside the whole C
        assert all((c[3] >> 16) & 1 for c in logs[0] if c[0] == 3)                     # ... and is in place
        assert all(c[2] > 0 for c in logs[0] if c[0] == 3)                             # no empty collective is posted
        for me in range(nranks):
            fi, fv, fp = finals[me]
            assert np.array_equal(fi, exp_idx) and np.array_equal(fv, exp_val), (form, nranks, trial, me)
            assert np.array_equal(fp, exp_ptr), (form, nranks, trial, me)


def test_counter_summaries_drop_the_aborted_launch(tmp_path):
    """scripts/pmc_per_launch.py (what collect_sq.sh / collect_traffic.sh summarise the rocprofv3 --pmc passes with): the value per
    kernel is the MEDIAN over its steady launches; a launch far shorter than the kernel's median duration -- the aborted pipeline run
    of a context's first call -- is dropped, and launches_used / launches_seen say so.  (Round 4's files averaged it in.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_per_launch", os.path.join(ROOT, "scripts", "pmc_per_launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = tmp_path / "p1"
    d.mkdir()
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,"
            "LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"]
    def add(disp, name, counter, value, dur):
        rows.append(f'1,{disp},"Agent 2",4,1,1,256,7,"{name}",512,0,0,64,0,80,"{counter}",{value},1000,{1000 + dur}')
    kern = "void spada::k_task<2, 2048>(spada::TaskArgs const*)"
    add(1, kern, "SQ_INSTS_VALU", 0.0, 1600)                       # the aborted run: 1.6 us
    for i, v in enumerate((262.0e6, 262.4e6, 261.8e6, 262.2e6)):   # four steady launches; one counter split over two rows (XCDs)
        add(2 + i, kern, "SQ_INSTS_VALU", v / 2, 770000 + 1000 * i)
        add(2 + i, kern, "SQ_INSTS_VALU", v / 2, 770000 + 1000 * i)
    add(9, "void at::native::fill(int)", "SQ_INSTS_VALU", 5.0, 3000)   # not one of ours
    (d / "p_counter_collection.csv").write_text("\n".join(rows) + "\n")
    res = mod.summarize([str(d)])
    k = res["k_task<2, 2048>"]
    assert k["launches_seen"] == 5 and k["launches_used"] == 4
    assert abs(k["SQ_INSTS_VALU"] - 262.1e6) < 1e3 and abs(k["SQ_INSTS_VALU__mean"] - 262.1e6) < 1e3
    assert list(res) == ["k_task<2, 2048>"]
