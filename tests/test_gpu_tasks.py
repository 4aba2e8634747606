"""GPU (-m gpu): the task pipeline's one-pass entry point (spada_dev_spgemm_fused) and its special paths, through the
C ABI, against the CPU oracle.  Bar as in test_gpu_parity.py: structure bit-exact, values within 1e-9 relative."""
import numpy as np
import pytest

from conftest import assert_parity, to_oracle
from fuzz_cases import random_case
from oracle import oracle

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def fused(engine, a, b, capacity=None, r0=0, r1=None):
    """C = A[r0:r1] * B by the one-pass entry point into context-owned device buffers of `capacity` entries."""
    import spada_sim_amd as S
    r1 = a.shape[0] if r1 is None else r1
    da = engine.upload(a)
    db = da if b is a else engine.upload(b)
    try:
        if capacity is None:
            capacity = S.count_products(a, b, r0, r1)
        p, i, v, nnz = engine.fused_owned(da, db, r0, r1, capacity)
        st = engine.stats()
        return engine.download(p, i, v, r1 - r0, nnz, b.shape[1]), st
    finally:
        engine.free(da)
        if db is not da:
            engine.free(db)


@pytest.fixture(scope="module")
def engine_sm():
    import spada_sim_amd as S
    e = S.Engine(accumulator=S.ACC_SORT_MERGE)
    yield e
    e.close()


GEN = [
    ("uniform_small", 5, 2000, 6, 1),
    ("rmat_s12", 0, 12, 8, 2),
    ("rmat_s14", 0, 14, 16, 3),          # long rows: BIG rows, many range tasks
    ("webbase_like_50k", 1, 50000, 155000, 4),
    ("cop20k_like_20k", 2, 20000, 0, 5),
    ("cage12_like_20k", 3, 20000, 0, 6),
    ("mc2depi_like", 4, 779 * 40, 0, 7),
]


@pytest.mark.parametrize("name,kind,p0,p1,seed", GEN)
def test_fused_generated_workloads(engine, name, kind, p0, p1, seed):
    import spada_sim_amd as S
    m = S.generate(kind, p0, p1, seed)
    c, st = fused(engine, m, m)
    a = to_oracle(m)
    ref = oracle.spgemm_spa(a, a)
    assert assert_parity(c, ref, a, a, RTOL) == 0
    assert st["nprod"] == oracle.count_products(a, a) and st["c_nnz"] == ref.nnz
    assert sum(st["cls_rows"]) == m.shape[0] and sum(st["cls_prod"]) == st["nprod"]
    assert st["scratch_products"] <= st["cls_prod"][4] and st["spill_rows"] <= st["cls_rows"][4]


def test_task_capacity_is_the_register_budget(engine):
    """A task hashes at most 2040 products on every input (round 2 chose between 1920 and 2040 from a sampled products / outputs
    ratio; with the table keyed by blocks of columns it never fills, and the fullest tasks are fastest everywhere); mesh-like and
    web-like inputs alike give the oracle's product."""
    import spada_sim_amd as S
    for name, kind, p0, p1, seed in (("cop20k", S.GEN_COP20K_LIKE, 20000, 0, 5), ("cage12", S.GEN_CAGE12_LIKE, 20000, 0, 6),
                                     ("web", S.GEN_WEBBASE_LIKE, 50000, 155000, 4), ("mc2depi", S.GEN_MC2DEPI_LIKE, 779 * 40, 0, 7)):
        m = S.generate(kind, p0, p1, seed)
        c, st = fused(engine, m, m)
        a = to_oracle(m)
        assert_parity(c, oracle.spgemm_spa(a, a), a, a, RTOL)
        assert st["task_product_limit"] == 2040, (name, st["task_product_limit"])


@pytest.mark.parametrize("seed", range(24))
def test_fused_random_sweep(engine, seed):
    a, b, desc = random_case(seed)
    c, _ = fused(engine, a, b)
    ao, bo = to_oracle(a), to_oracle(b)
    ref = oracle.spgemm_sortmerge(ao, bo)
    assert_parity(c, ref, ao, bo, RTOL)


def test_fused_capacity_error_then_numeric(engine):
    """capacity < nnz(C): SPADA_ERR_CAPACITY, C.indptr and nnz(C) complete; a numeric call into large enough buffers follows."""
    import spada_sim_amd as S
    m = S.generate(S.GEN_RMAT, 11, 8, 5)
    a = to_oracle(m)
    ref = oracle.spgemm_spa(a, a)
    with pytest.raises(S.SpadaError) as e:
        fused(engine, m, m, capacity=ref.nnz // 2)
    assert e.value.code == 9 and str(ref.nnz) in str(e.value)
    # the same through the device API, keeping the handles alive for the numeric call
    da = engine.upload(m)
    with pytest.raises(S.SpadaError) as e:
        engine.fused_owned(da, da, 0, m.shape[0], 16)
    assert e.value.code == 9
    p, i, v = engine.numeric_owned()
    c = engine.download(p, i, v, m.shape[0], ref.nnz, m.shape[1])
    assert_parity(c, ref, a, a, RTOL)
    engine.free(da)
    # host-pointer form: indptr and nnz(C) come back with the error, spada_spgemm_numeric completes the product
    with pytest.raises(S.SpadaError) as e:
        engine.spgemm_fused(m, m, capacity=10)
    assert e.value.code == 9
    assert_parity(engine.spgemm_fused(m, m), ref, a, a, RTOL)


def test_fused_row_blocks(engine):
    """Row ranges (the A-row block of one GPU): every block's C.indptr starts at 0 and the blocks concatenate."""
    import spada_sim_amd as S
    m = S.generate(S.GEN_WEBBASE_LIKE, 30000, 95000, 8)
    a = to_oracle(m)
    ref = oracle.spgemm_spa(a, a)
    bounds = [0, 1, 7000, 7000, 19999, 30000]
    pos = 0
    for r0, r1 in zip(bounds[:-1], bounds[1:]):
        c, _ = fused(engine, m, m, r0=r0, r1=r1)
        lo, hi = int(ref.indptr[r0]), int(ref.indptr[r1])
        assert np.array_equal(c.indptr, ref.indptr[r0:r1 + 1] - ref.indptr[r0])
        assert np.array_equal(c.indices, ref.indices[lo:hi])
        assert np.all(np.abs(c.data - ref.data[lo:hi]) <= RTOL * np.abs(ref.data[lo:hi]))
        pos += c.nnz()
    assert pos == ref.nnz


@pytest.mark.parametrize("cols,multi_pass", [(6_000_000, False), (40_000_000, True)])
def test_heavy_buckets_wider_than_the_table(engine, cols, multi_pass):
    """A BIG row whose products crowd into histogram buckets wider than the table (more than 2 M columns).  6 M columns: buckets of
    8192 columns, every heavy one becomes four column sub-range tasks over the same scratch slice (no multi-pass task); 40 M
    columns: buckets of 65536 columns, more than eight sub-ranges, so the range task halves its column range depth first
    (stats.multi_pass_tasks > 0).  Also rows with many entries and short B rows."""
    import spada_sim_amd as S
    rng = np.random.default_rng(3)
    k = 4000
    # B: row j has 6 entries inside a 40 000-column window + 2 far outliers
    bi, bv, bptr = [], [], [0]
    for j in range(k):
        c = np.unique(np.concatenate([3_000_000 + rng.integers(0, 40_000, 6), rng.integers(0, cols, 2)]))
        bi.append(c)
        bv.append(rng.uniform(0.1, 1.0, len(c)))
        bptr.append(bptr[-1] + len(c))
    b = S.CsMat((k, cols), np.array(bptr, np.uint64), np.concatenate(bi).astype(np.uint64), np.concatenate(bv))
    # A: row 0 selects every B row (32 000 products, ~24 000 of them in one 5 859-column-wide bucket range), the others few
    ai, av, aptr = [np.arange(k)], [rng.uniform(0.1, 1.0, k)], [0, k]
    for r in range(1, 40):
        c = np.unique(rng.integers(0, k, int(rng.integers(0, 700))))
        ai.append(c)
        av.append(rng.uniform(0.1, 1.0, len(c)))
        aptr.append(aptr[-1] + len(c))
    a = S.CsMat((40, k), np.array(aptr, np.uint64), np.concatenate(ai).astype(np.uint64), np.concatenate(av))
    ao, bo = to_oracle(a), to_oracle(b)
    ref = oracle.spgemm_sortmerge(ao, bo)
    c, st = fused(engine, a, b)
    assert_parity(c, ref, ao, bo, RTOL)
    assert (st["multi_pass_tasks"] > 0) == multi_pass and st["cls_rows"][4] > 0 and st["spill_rows"] > 0
    # and through the two-phase contract
    c2 = engine.spgemm(a, b)
    assert_parity(c2, ref, ao, bo, RTOL)


def test_counting_mode_positions_across_tiles(engine, engine_sm):
    """The counting mode has no chain: tasks leave their counts and k_pos1/2/3 scan them into C.indptr and the range positions,
    2048 tasks per tile.  R-MAT 16 has 232 k tasks (114 tiles) and BIG rows whose range tasks straddle tile boundaries: the
    two-phase product must equal the one-pass product (chain) entry for entry, and its row pointers the oracle's counts."""
    import spada_sim_amd as S
    m = S.generate(S.GEN_RMAT, 16, 16, 22)
    c1, st1 = fused(engine, m, m)
    assert st1["n_tasks"] > 50 * 2048 and st1["cls_rows"][4] > 0
    d = engine.upload(m)
    nnz = engine.symbolic(d, d, 0, m.shape[0])
    p, i, v = engine.numeric_owned()
    c2 = engine.download(p, i, v, m.shape[0], nnz, m.shape[1])
    engine.free(d)
    assert nnz == c1.nnz()
    assert np.array_equal(c1.indptr, c2.indptr) and np.array_equal(c1.indices, c2.indices)
    assert np.all(np.abs(c1.data - c2.data) <= RTOL * np.abs(c1.data))
    ao = to_oracle(m)
    ref = oracle.spgemm_spa(ao, ao)          # both entry points against the ORACLE, values included (164 M entries)
    for c in (c1, c2):
        assert np.array_equal(c.indptr, ref.indptr) and np.array_equal(c.indices, ref.indices)
        assert np.all(np.abs(c.data - ref.data) <= RTOL * np.abs(ref.data))
    del ref
    # the sort-merge kernel shares the position kernels; a row range that starts inside the matrix
    r0, r1 = 1000, 30000
    d = engine_sm.upload(m)
    nnz = engine_sm.symbolic(d, d, r0, r1)
    p, i, v = engine_sm.numeric_owned()
    c3 = engine_sm.download(p, i, v, r1 - r0, nnz, m.shape[1])
    engine_sm.free(d)
    lo, hi = int(c1.indptr[r0]), int(c1.indptr[r1])
    assert nnz == hi - lo
    assert np.array_equal(c3.indptr.astype(np.int64), c1.indptr[r0:r1 + 1].astype(np.int64) - lo)
    assert np.array_equal(c3.indices, c1.indices[lo:hi])
    assert np.all(np.abs(c3.data - c1.data[lo:hi]) <= RTOL * np.abs(c1.data[lo:hi]))


def test_phase_timing_switch():
    """spada_set_phase_timing(0): the event records between the small kernels are left out (their three times read 0), the call
    and the task kernel are still timed and the product is the same.  On a context of its own: which of the small kernels run on
    the side streams depends on the context's previous call, and a phase may legitimately read 0.0 (intervals of the engine
    stream between back-to-back event records) -- what must hold is that the phases are not negative, that together they are a
    part of the call, and that the switch turns them off."""
    import spada_sim_amd as S
    m = S.generate(S.GEN_RMAT, 12, 8, 31)
    eng = S.Engine()
    try:
        for _ in range(2):   # (the second call forks the side streams the first call found work for)
            c1, st1 = fused(eng, m, m)
            phases = [st1["ms_row_stats"], st1["ms_big_expand"], st1["ms_cut"]]
            assert all(p >= 0 for p in phases) and sum(phases) > 0 and st1["ms_task"] > 0
            assert st1["ms_fused_call"] >= sum(phases) + st1["ms_task"] - 1e-3
        eng.set_phase_timing(False)
        c2, st2 = fused(eng, m, m)
        assert st2["ms_row_stats"] == 0 and st2["ms_big_expand"] == 0 and st2["ms_cut"] == 0
        assert st2["ms_task"] > 0 and st2["ms_fused_call"] >= st2["ms_task"]
        assert np.array_equal(c1.indptr, c2.indptr) and np.array_equal(c1.indices, c2.indices)
    finally:
        eng.close()


def test_call_intervals_are_never_lost():
    """The host learns of a run's end from a sequence number the last kernel writes into pinned memory -- possibly before the runtime
    has taken in the completion of the event records in front of that kernel.  The intervals of spada_stats must be there all the
    same (an interval that came back as 0.0 once in a few hundred calls made a caller's rate a division by zero): 3000 calls of the
    one-pass entry point on device-resident operands, every one with a positive call and task-kernel time."""
    import spada_sim_amd as S
    m = S.generate(S.GEN_UNIFORM, 20000, 6, 9)
    eng = S.Engine()
    try:
        d = eng.upload(m)
        cap = S.count_products(m, m, 0, m.shape[0])
        for call in range(3000):
            eng.fused_owned(d, d, 0, m.shape[0], cap)
            st = eng.stats()
            assert st["ms_fused_call"] > 0 and st["ms_task"] > 0, (call, st["ms_fused_call"], st["ms_task"])
        eng.free(d)
    finally:
        eng.close()


def test_workspace_growth_reruns(engine):
    """A fresh context sizes its workspaces (part records, range descriptors, scratch, task list) from the counters of the runs
    that overflowed them and runs the pipeline again: the part records first, then whatever the plan behind them needs."""
    import spada_sim_amd as S
    eng = S.Engine()
    try:
        m = S.generate(S.GEN_RMAT, 13, 16, 4)
        c, st = fused(eng, m, m)
        a = to_oracle(m)
        assert_parity(c, oracle.spgemm_spa(a, a), a, a, RTOL)
        assert st["pipeline_runs"] in (1, 2, 3)
        # (the second call on an input of BIG rows may run the engine's other pipeline for the first time -- count + numeric inside the
        # one-pass entry point -- whose cut tables and task list can want room once more; from then on nothing grows)
        runs = []
        for _ in range(3):
            c, st = fused(eng, m, m)
            runs.append(st["pipeline_runs"])
        assert runs[-1] == 1 and max(runs) <= 3
        assert_parity(c, oracle.spgemm_spa(a, a), a, a, RTOL)
    finally:
        eng.close()


def test_one_output_with_thousands_of_products():
    """A row whose 5000 entries all select B rows that contain column 7: one output entry is the sum of 5000 products, more
    than a task's table / sorting network holds at once.  Hash accumulator: within 1e-9; sort-merge: bit-identical (the run
    is added in ascending k across the pieces)."""
    import spada_sim_amd as S
    rng = np.random.default_rng(17)
    k, cols = 5000, 3000
    bi, bv, bptr = [], [], [0]
    for j in range(k):
        c = np.unique(np.concatenate([[7], rng.integers(0, cols, 3)]))
        bi.append(c)
        bv.append(rng.uniform(-1.0, 1.0, len(c)))
        bptr.append(bptr[-1] + len(c))
    b = S.CsMat((k, cols), np.array(bptr, np.uint64), np.concatenate(bi).astype(np.uint64), np.concatenate(bv))
    a = S.CsMat((2, k), np.array([0, k, k + 3], np.uint64), np.concatenate([np.arange(k), [1, 5, 9]]).astype(np.uint64),
                rng.uniform(-1.0, 1.0, k + 3))
    ao, bo = to_oracle(a), to_oracle(b)
    ref = oracle.spgemm_sortmerge(ao, bo)
    for acc in (S.ACC_LDS_HASH, S.ACC_SORT_MERGE):
        eng = S.Engine(accumulator=acc)
        try:
            for c in (eng.spgemm(a, b), eng.spgemm_fused(a, b)):
                assert_parity(c, ref, ao, bo, RTOL)
                if acc == S.ACC_SORT_MERGE:
                    assert np.array_equal(c.data, ref.data)
        finally:
            eng.close()


def test_numeric_in_chunks(engine):
    """spada_dev_spgemm_numeric_plan / _chunk: the numeric phase in pieces (what the overlapped exchange uses) fills exactly the
    entries its plan announces, piece by piece, and the pieces add up to the product."""
    import ctypes
    import spada_sim_amd as S
    from spada_sim_amd import _ffi
    m = S.generate(S.GEN_RMAT, 12, 12, 21)
    a = to_oracle(m)
    ref = oracle.spgemm_spa(a, a)
    d = engine.upload(m)
    nnz = engine.symbolic(d, d, 0, m.shape[0])
    p, i, v = engine.numeric_owned()          # reference run; also provides device buffers of the right size
    L = _ffi.lib()
    K = 5
    pos = np.zeros(K + 1, np.uint64)
    _ffi.check(L.spada_dev_spgemm_numeric_plan(engine._ctx, K, pos.ctypes.data_as(_ffi.u64p)))
    assert pos[0] == 0 and pos[-1] == nnz == ref.nnz and np.all(np.diff(pos.astype(np.int64)) >= 0)
    for k in [3, 0, 4, 1, 2]:                 # pieces are independent: any order
        ev = _ffi.vp()
        _ffi.check(L.spada_dev_spgemm_numeric_chunk(engine._ctx, k, _ffi.vp(i), _ffi.vp(v), ctypes.byref(ev)))
        assert ev.value
    _ffi.check(L.spada_dev_synchronize(engine._ctx))
    c = engine.download(p, i, v, m.shape[0], nnz, m.shape[1])
    engine.free(d)
    assert_parity(c, ref, a, a, RTOL)


def test_rccl_exchange_single_rank(engine):
    """libspada_comm.so on a one-rank RCCL communicator (all this box has): both forms -- allgatherv after a one-pass SpGEMM and
    the distributed two-phase call with the overlapped, chunked exchange -- must reproduce the product in the caller's buffers.
    The N > 1 logic is covered by tests/test_multi_rank_gloo.py (same offsets arithmetic) and by the driver's 8-GPU run."""
    import spada_sim_amd as S
    m = S.generate(S.GEN_WEBBASE_LIKE, 40000, 125000, 3)
    a = to_oracle(m)
    ref = oracle.spgemm_spa(a, a)
    comm = S.Comm(S.Comm.unique_id(), 0, 1, 0)
    d = engine.upload(m)
    try:
        cap = S.count_products(m, m, 0, m.shape[0])
        p, i, v, nnz = engine.fused_owned(d, d, 0, m.shape[0], cap)
        rows_r, nnz_r = comm.allgather_counts(m.shape[0], nnz)
        assert list(rows_r) == [m.shape[0]] and list(nnz_r) == [ref.nnz]
        # receive buffers of the caller: separate allocations, pre-filled with a poison pattern, so that an exchange that wrote
        # nothing (or wrote to the wrong offsets) cannot pass
        import torch
        dev = torch.device("cuda", 0)

        def poisoned():
            fp = torch.full((m.shape[0] + 1,), -1, dtype=torch.int64, device=dev)
            fi = torch.full((max(ref.nnz, 1),), -1, dtype=torch.int32, device=dev)
            fv = torch.full((max(ref.nnz, 1),), float("nan"), dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            return fp, fi, fv

        def check(fp, fi, fv):
            torch.cuda.synchronize()
            got = S.CsMat(m.shape, fp.cpu().numpy().astype(np.uint64), fi.cpu().numpy().astype(np.uint32).astype(np.uint64),
                          fv.cpu().numpy())
            assert_parity(got, ref, a, a, RTOL)

        fp, fi, fv = poisoned()
        comm.allgatherv_c(p, i, v, rows_r, nnz_r, fp.data_ptr(), fi.data_ptr(), fv.data_ptr())
        check(fp, fi, fv)
        # distributed two-phase call, 4 pieces, into freshly poisoned buffers
        rows_r, nnz_r = comm.dist_symbolic(engine, d, d, 0, m.shape[0], 4)
        assert list(nnz_r) == [ref.nnz]
        fp, fi, fv = poisoned()
        comm.dist_numeric(engine, fp.data_ptr(), fi.data_ptr(), fv.data_ptr())
        check(fp, fi, fv)
        assert engine.stats()["ms_numeric_call"] > 0       # the distributed numeric call is timed as a whole (bench roofline)
    finally:
        engine.free(d)
        comm.close()


def test_engine_first_then_torch_in_one_process():
    """The engine's library and PyTorch share ONE HIP runtime whichever is loaded first (a torch wheel carries its own copy; two
    runtimes in a process leave the second without a GPU): a fresh process runs the engine, then imports torch and uses the GPU."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, faulthandler; sys.path.insert(0, %r)\n"
        "faulthandler.dump_traceback_later(150, exit=True)\n"   # (a hang says where)
        "import spada_sim_amd as S\n"
        "assert 'torch' not in sys.modules\n"
        "eng = S.Engine(); m = S.generate(S.GEN_RMAT, 8, 4, 3); c = eng.spgemm(m, m); n = c.nnz()\n"
        "import torch\n"
        "x = torch.arange(8, device='cuda:0').sum().item()\n"
        "c2 = eng.spgemm(m, m)\n"
        "assert x == 28 and c2.nnz() == n and n > 0\n"
        "print('ok', n)\n" % repo)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=400)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok"), r.stdout + r.stderr


@pytest.mark.parametrize("span_blocks", [40, 81, 82, 83, 120, 163, 164, 165, 244, 245, 246, 247, 700, 2047, 2048, 2049, 4094, 4096, 4098,
                                         6142, 6144, 6146])
def test_dense_tasks_at_the_slot_boundary(engine, span_blocks):
    """Batches whose rows' column spans add up to about the table's slots -- 3072 since round 4, 2048 before (a slot of the dense
    layout covers 32 columns, i.e. two of the 16-column blocks counted here): below the bound a batch (or a single row) takes the
    DENSE path of the batch task (slot = place of the columns in their row, no keys), above it the monotone table with keys; every
    row of C spans exactly `span_blocks` blocks, ~25 rows make a batch (the bound of a batch: ~246 blocks per row; of a single row:
    6144).  Both entry points against the oracle."""
    import spada_sim_amd as S
    rng = np.random.default_rng(span_blocks)
    rows, k = 3000, 1500
    width = span_blocks * 16
    n = 200000 + width
    # B: rows 2 i and 2 i + 1 share a window of `width` columns that starts at a multiple of 16: first entry at its first column,
    # last entry at its last, 38 random ones in between
    b_ptr = np.zeros(k + 1, np.uint64)
    b_idx, b_val = [], []
    lo = 0
    for j in range(k):
        if j % 2 == 0:
            lo = int(rng.integers(0, (n - width) // 16)) * 16
        inner = np.unique(rng.integers(lo + 1, lo + width - 1, 38)) if width > 2 else np.zeros(0, np.int64)
        c = np.unique(np.concatenate([[lo, lo + width - 1], inner]))
        b_idx.append(c.astype(np.uint64))
        b_val.append(rng.uniform(0.5, 1.5, len(c)))
        b_ptr[j + 1] = b_ptr[j] + len(c)
    b = S.CsMat((k, n), b_ptr, np.concatenate(b_idx), np.concatenate(b_val))
    # A: every row selects one such pair of B rows: its row of C spans the pair's window, i.e. exactly span_blocks blocks
    a_ptr = np.arange(0, 2 * rows + 1, 2, dtype=np.uint64)
    sel = 2 * rng.integers(0, k // 2, rows)
    a_idx = np.stack([sel, sel + 1], 1).reshape(-1).astype(np.uint64)
    a = S.CsMat((rows, k), a_ptr, a_idx, rng.uniform(0.5, 1.5, 2 * rows))
    ao, bo = to_oracle(a), to_oracle(b)
    ref = oracle.spgemm_sortmerge(ao, bo)
    c, st = fused(engine, a, b)
    assert_parity(c, ref, ao, bo, RTOL)
    c2 = engine.spgemm(a, b)
    assert_parity(c2, ref, ao, bo, RTOL)


@pytest.mark.parametrize("cluster_blocks", [40, 300, 620])
def test_clustered_columns_take_the_second_attempt(engine, cluster_blocks):
    """Rows whose blocks CLUSTER: every row of C has one column near 0, one near 2 M and `cluster_blocks` x 8 columns, one per
    32-column block, in a window around column 1 M.  The monotone table's home slots are a linear function of the column inside the
    row's span (62 500 blocks here), so the cluster's blocks share a handful of home slots; past SPADA_PROBE_MAX = 24 slots of
    displacement the task starts over with home slots equalised over its own histogram (spgemm_batch.hip.hpp).  40 blocks per B
    row stay within what two or three rows of a batch may displace; 300 and 620 force the second attempt.  Both entry points
    (the counting mode of the two-phase one hashes its home slots instead) against the oracle."""
    import spada_sim_amd as S
    rng = np.random.default_rng(cluster_blocks)
    per = max(cluster_blocks // 8, 1)          # cluster columns per B row: eight B rows make a row of C
    rows, k, n = 600, 1600, 2_100_000
    b_ptr = np.zeros(k + 1, np.uint64)
    b_idx, b_val = [], []
    for j in range(k):
        base = 1_000_000 + (j % 8) * per * 32 + int(rng.integers(0, 4)) * 8 * per * 32
        cl = base + 32 * np.arange(per) + rng.integers(0, 32, per)
        c = np.unique(np.concatenate([[j % 97], cl, [2_000_000 + j % 89]]))
        b_idx.append(c.astype(np.uint64))
        b_val.append(rng.uniform(0.5, 1.5, len(c)))
        b_ptr[j + 1] = b_ptr[j] + len(c)
    b = S.CsMat((k, n), b_ptr, np.concatenate(b_idx), np.concatenate(b_val))
    # A: row i selects eight B rows with the eight residues mod 8 (disjoint cluster windows inside one row of C)
    a_ptr = np.arange(0, 8 * rows + 1, 8, dtype=np.uint64)
    a_idx = np.sort(8 * rng.integers(0, k // 8, (rows, 8)) + np.arange(8)[None, :], axis=1).reshape(-1).astype(np.uint64)
    a = S.CsMat((rows, k), a_ptr, a_idx, rng.uniform(0.5, 1.5, 8 * rows))
    ao, bo = to_oracle(a), to_oracle(b)
    ref = oracle.spgemm_sortmerge(ao, bo)
    c, st = fused(engine, a, b)
    assert_parity(c, ref, ao, bo, RTOL)
    c2 = engine.spgemm(a, b)
    assert_parity(c2, ref, ao, bo, RTOL)


def test_cut_table_switch_gives_the_same_product():
    """SPADA_CUT_TABLE=0 (the direct range tasks search for their bounds themselves) and the default (k_big_cuts leaves them in a
    table; in the one-pass mode only for rows with few searches per product) give the same C: R-MAT 13, both entry points."""
    import os
    import spada_sim_amd as S
    m = S.generate(S.GEN_RMAT, 13, 16, 5)
    a = to_oracle(m)
    ref = oracle.spgemm_spa(a, a)
    old = os.environ.get("SPADA_CUT_TABLE")
    try:
        for flag in ("0", "1"):
            os.environ["SPADA_CUT_TABLE"] = flag
            eng = S.Engine()
            try:
                c, st = fused(eng, m, m)
                assert_parity(c, ref, a, a, RTOL)
                assert_parity(eng.spgemm(m, m), ref, a, a, RTOL)
            finally:
                eng.close()
    finally:
        if old is None:
            os.environ.pop("SPADA_CUT_TABLE", None)
        else:
            os.environ["SPADA_CUT_TABLE"] = old


def test_scatter_cursor_layouts_give_the_same_product():
    """Spilled rows: one scatter cursor per (part, range) (default: k_big_plan leaves marks in the other buckets of a range) and one per
    (part, bucket) (SPADA_RANGE_CURSORS=0) give the same C, through both entry points and both accumulators (the sort-merge tasks
    read the same scratch slices): R-MAT 14, whose hub rows are spilled."""
    import os
    import spada_sim_amd as S
    m = S.generate(S.GEN_RMAT, 14, 16, 4)
    a = to_oracle(m)
    ref = oracle.spgemm_sortmerge(a, a)
    old = os.environ.get("SPADA_RANGE_CURSORS")
    try:
        for flag in ("0", "1"):
            os.environ["SPADA_RANGE_CURSORS"] = flag
            for acc in (S.ACC_LDS_HASH, S.ACC_SORT_MERGE):
                eng = S.Engine(accumulator=acc)
                try:
                    c = eng.spgemm(m, m)
                    assert eng.stats()["spill_rows"] > 0
                    assert_parity(c, ref, a, a, RTOL)
                    if acc == S.ACC_LDS_HASH:
                        c1, st = fused(eng, m, m)
                        assert st["spill_rows"] > 0
                        assert_parity(c1, ref, a, a, RTOL)
                finally:
                    eng.close()
    finally:
        if old is None:
            os.environ.pop("SPADA_RANGE_CURSORS", None)
        else:
            os.environ["SPADA_RANGE_CURSORS"] = old


def test_part_sizes_give_the_same_product():
    """The parts of a BIG row are 8 K products by default and 64 K on calls whose predecessor had a billion products in BIG rows;
    any size is correct.  SPADA_PART_SHIFT forces 1 K and 64 K parts on R-MAT 14 (hub rows: spilled, walked entry by entry where a
    part is a few long B rows, their scatter runs handed out by ticket): both entry points and both accumulators give the oracle's
    product, and the two-phase contract gives it again on the same context (the second call's guesses come from the first)."""
    import os
    import spada_sim_amd as S
    m = S.generate(S.GEN_RMAT, 14, 16, 4)
    a = to_oracle(m)
    ref = oracle.spgemm_sortmerge(a, a)
    old = os.environ.get("SPADA_PART_SHIFT")
    try:
        for shift in ("10", "16"):
            os.environ["SPADA_PART_SHIFT"] = shift
            for acc in (S.ACC_LDS_HASH, S.ACC_SORT_MERGE):
                eng = S.Engine(accumulator=acc)
                try:
                    for _ in range(2):
                        c = eng.spgemm(m, m)
                        assert eng.stats()["spill_rows"] > 0
                        assert_parity(c, ref, a, a, RTOL)
                    if acc == S.ACC_LDS_HASH:
                        c1, st = fused(eng, m, m)
                        assert_parity(c1, ref, a, a, RTOL)
                finally:
                    eng.close()
    finally:
        if old is None:
            os.environ.pop("SPADA_PART_SHIFT", None)
        else:
            os.environ["SPADA_PART_SHIFT"] = old


@pytest.mark.parametrize("run", [6, 10, 11])
def test_many_displaced_blocks_take_the_cluster_fix(engine, run):
    """Batches of ~90 short rows, each with two runs of `run` ADJACENT 32-column blocks and a far column that stretches the row's span
    to 2 000 blocks: the ~22 blocks of a row share one or two home slots of its ~30-slot region, so most blocks of the task are
    displaced -- by less than SPADA_PROBE_MAX slots each, i.e. without a second attempt, but far more of them than the list of
    displaced blocks holds (352): the order stage examines whole clusters instead (spgemm_batch.hip.hpp).  Both entry points
    against the oracle."""
    import spada_sim_amd as S
    rng = np.random.default_rng(run)
    rows, k, n = 4000, 2000, 200_000
    b_ptr = np.zeros(k + 1, np.uint64)
    b_idx, b_val = [], []
    for j in range(k):
        lo = int(rng.integers(0, 3000)) * 32
        c = np.unique(np.concatenate([lo + 32 * np.arange(run) + rng.integers(0, 32, run), [lo + 64_000 + j % 31]]))
        b_idx.append(c.astype(np.uint64))
        b_val.append(rng.uniform(0.5, 1.5, len(c)))
        b_ptr[j + 1] = b_ptr[j] + len(c)
    b = S.CsMat((k, n), b_ptr, np.concatenate(b_idx), np.concatenate(b_val))
    a_ptr = np.arange(0, 2 * rows + 1, 2, dtype=np.uint64)
    a_idx = np.sort(rng.choice(k, (rows, 2)), axis=1)
    a_idx[:, 1] = np.where(a_idx[:, 1] == a_idx[:, 0], (a_idx[:, 0] + 1) % k, a_idx[:, 1])
    a_idx = np.sort(a_idx, axis=1).reshape(-1).astype(np.uint64)
    a = S.CsMat((rows, k), a_ptr, a_idx, rng.uniform(0.5, 1.5, 2 * rows))
    ao, bo = to_oracle(a), to_oracle(b)
    ref = oracle.spgemm_sortmerge(ao, bo)
    c, st = fused(engine, a, b)
    assert_parity(c, ref, ao, bo, RTOL)
    c2 = engine.spgemm(a, b)
    assert_parity(c2, ref, ao, bo, RTOL)


def test_a_run_that_expected_no_big_rows_is_repeated_when_there_are_some():
    """A context whose last run found no BIG row does not launch the BIG-row kernels in the next one; when that next product does
    have BIG rows the run is thrown away and repeated with them (stats.pipeline_runs = 2), and the product is right -- also for
    the call after it (no second repetition) and in the two-phase contract."""
    import spada_sim_amd as S
    flat = S.generate(S.GEN_UNIFORM, 3000, 4, 5)            # every row 4 entries: no BIG rows
    hubs = S.generate(S.GEN_RMAT, 12, 16, 9)                # power-law rows: BIG rows
    eng = S.Engine()
    try:
        c0, st0 = fused(eng, flat, flat)
        assert st0["cls_rows"][4] == 0
        assert_parity(c0, oracle.spgemm_sortmerge(to_oracle(flat), to_oracle(flat)), to_oracle(flat), to_oracle(flat), RTOL)
        ref = oracle.spgemm_sortmerge(to_oracle(hubs), to_oracle(hubs))
        c1, st1 = fused(eng, hubs, hubs)
        assert st1["cls_rows"][4] > 0 and st1["pipeline_runs"] >= 2
        assert_parity(c1, ref, to_oracle(hubs), to_oracle(hubs), RTOL)
        c2, st2 = fused(eng, hubs, hubs)
        assert st2["pipeline_runs"] == 1
        assert_parity(c2, ref, to_oracle(hubs), to_oracle(hubs), RTOL)
        c3, _ = fused(eng, flat, flat)                       # ... and back: the kernels run once more, find nothing, and are left out again
        assert np.array_equal(c3.indptr, c0.indptr) and np.array_equal(c3.indices, c0.indices)
        c4 = eng.spgemm(hubs, hubs)                          # two-phase contract after a run without BIG rows
        assert_parity(c4, ref, to_oracle(hubs), to_oracle(hubs), RTOL)
    finally:
        eng.close()


def test_one_context_across_row_blocks_matrices_and_failed_calls():
    """No kernel clears anything at the head of a run: the counters of a run are cleared behind the run before it and the per-row
    accumulators are put back behind every run.  One context through row blocks of different sizes and offsets, matrices of
    different shapes, a call that fails for capacity and a retry of the workspaces in between: every product is the oracle's."""
    import spada_sim_amd as S
    mats = [S.generate(S.GEN_RMAT, 11, 8, 3), S.generate(S.GEN_UNIFORM, 5000, 3, 8), S.generate(S.GEN_RMAT, 13, 12, 4),
            S.generate(S.GEN_UNIFORM, 700, 9, 2)]
    refs = [oracle.spgemm_sortmerge(to_oracle(m), to_oracle(m)) for m in mats]
    eng = S.Engine()
    try:
        for rnd in range(2):
            for m, ref in zip(mats, refs):
                n = m.shape[0]
                for (r0, r1) in ((0, n), (n // 3, n - n // 5), (n - 7, n), (5, 6)):
                    c, st = fused(eng, m, m, r0=r0, r1=r1)
                    lo, hi = int(ref.indptr[r0]), int(ref.indptr[r1])
                    assert st["c_nnz"] == hi - lo
                    assert np.array_equal(c.indptr.astype(np.int64), ref.indptr[r0:r1 + 1].astype(np.int64) - lo)
                    assert np.array_equal(c.indices, ref.indices[lo:hi])
                    assert np.all(np.abs(c.data - ref.data[lo:hi]) <= RTOL * np.abs(ref.data[lo:hi]))
                if rnd == 0:
                    with pytest.raises(S.SpadaError):          # capacity too small: the call fails, the context stays usable
                        fused(eng, m, m, capacity=max(ref.nnz // 2, 1))
                c2 = eng.spgemm(m, m)                          # two-phase contract on the same context
                assert_parity(c2, ref, to_oracle(m), to_oracle(m), RTOL)
    finally:
        eng.close()


def test_a_run_that_expected_no_spilled_rows_is_stopped_and_repeated():
    """The scatter kernel is left out when the context's previous run spilled no row; k_cut3 stops a run whose plan spills rows after
    all (before any task can read an unfilled scratch slice) and the engine repeats it with the scatter."""
    import spada_sim_amd as S
    mild = S.generate(S.GEN_COP20K_LIKE, 20000, 0, 5)       # BIG rows, none spilled
    hubs = S.generate(S.GEN_RMAT, 14, 16, 4)                # hub rows: spilled
    eng = S.Engine()
    try:
        c0, st0 = fused(eng, mild, mild)
        assert st0["spill_rows"] == 0
        assert_parity(c0, oracle.spgemm_sortmerge(to_oracle(mild), to_oracle(mild)), to_oracle(mild), to_oracle(mild), RTOL)
        c0b, st0b = fused(eng, mild, mild)                   # (this run leaves the scatter out)
        assert np.array_equal(c0b.indices, c0.indices) and st0b["pipeline_runs"] == 1
        ref = oracle.spgemm_sortmerge(to_oracle(hubs), to_oracle(hubs))
        c1, st1 = fused(eng, hubs, hubs)
        assert st1["spill_rows"] > 0 and st1["pipeline_runs"] >= 2
        assert_parity(c1, ref, to_oracle(hubs), to_oracle(hubs), RTOL)
        c2 = eng.spgemm(mild, mild)                          # two-phase, back on the input without spilled rows ...
        assert np.array_equal(c2.indices, c0.indices)
        c3 = eng.spgemm(hubs, hubs)                          # ... and the hubs again through the two-phase contract
        assert_parity(c3, ref, to_oracle(hubs), to_oracle(hubs), RTOL)
    finally:
        eng.close()


@pytest.mark.gpu
def test_one_pass_entry_point_chooses_its_pipeline_by_rule_on_the_first_call():
    """spada_dev_spgemm_fused on an input whose products lie mostly in BIG rows runs count + numeric into the caller's buffers -- on
    the FIRST call of a fresh context already (the run reads its row statistics back once and goes on in the right mode;
    stats: pipeline_kind = 1, ms_symbolic_call > 0 next to ms_fused_call) -- and one pass on every other input; every call returns
    the oracle's product.  SPADA_AUTO=0 keeps one pass everywhere, SPADA_AUTO=measure is round 5's three-call comparison."""
    import spada_sim_amd as S
    hubs = S.generate(S.GEN_RMAT, 13, 16, 11)
    flat = S.generate(S.GEN_UNIFORM, 4000, 5, 3)
    ref = oracle.spgemm_sortmerge(to_oracle(hubs), to_oracle(hubs))
    cap = S.count_products(hubs, hubs, 0, hubs.shape[0])
    eng = S.Engine()
    try:
        d = eng.upload(hubs)
        for call in range(3):
            p, i, v, nnz = eng.fused_owned(d, d, 0, hubs.shape[0], cap)
            st = eng.stats()
            assert st["cls_prod"][4] * 2 > st["nprod"]
            assert st["pipeline_kind"] == 1 and st["ms_symbolic_call"] > 0, (call, st["pipeline_kind"])
            assert st["ms_fused_call"] > 0 and st["ms_fused_call"] >= st["ms_task"] > 0 and st["ms_wall_call"] > 0
            c = eng.download(p, i, v, hubs.shape[0], nnz, hubs.shape[1])
            assert_parity(c, ref, to_oracle(hubs), to_oracle(hubs), RTOL)
        with pytest.raises(S.SpadaError):                    # capacity too small
            eng.fused_owned(d, d, 0, hubs.shape[0], max(ref.nnz // 3, 1))
        eng.free(d)
        d2 = eng.upload(flat)                                # another input on the same context: one pass from its first call
        for call in range(3):
            eng.fused_owned(d2, d2, 0, flat.shape[0], S.count_products(flat, flat, 0, flat.shape[0]))
            st = eng.stats()
            assert st["pipeline_kind"] == 0 and st["ms_symbolic_call"] == 0
        eng.free(d2)
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("auto", ["0", "measure"])
def test_one_pass_entry_point_switches(auto, monkeypatch):
    """SPADA_AUTO=0: one pass whatever the input; SPADA_AUTO=measure: first call one pass, second count + numeric, then the faster."""
    import spada_sim_amd as S
    monkeypatch.setenv("SPADA_AUTO", auto)
    hubs = S.generate(S.GEN_RMAT, 13, 16, 11)
    ref = oracle.spgemm_sortmerge(to_oracle(hubs), to_oracle(hubs))
    cap = S.count_products(hubs, hubs, 0, hubs.shape[0])
    eng = S.Engine()
    try:
        d = eng.upload(hubs)
        kinds = []
        for call in range(4):
            p, i, v, nnz = eng.fused_owned(d, d, 0, hubs.shape[0], cap)
            kinds.append(eng.stats()["pipeline_kind"])
            c = eng.download(p, i, v, hubs.shape[0], nnz, hubs.shape[1])
            assert_parity(c, ref, to_oracle(hubs), to_oracle(hubs), RTOL)
        if auto == "0":
            assert kinds == [0, 0, 0, 0]
        else:
            assert kinds[0] == 0 and kinds[1] == 1
        eng.free(d)
    finally:
        eng.close()


@pytest.mark.gpu
def test_the_first_call_of_a_context_is_one_pipeline_run():
    """A fresh context sizes its data-dependent workspaces from the row statistics it reads back behind the row classes: the first
    call on the web and mesh surrogates no longer finds a workspace too small, stops and runs again (pipeline_runs == 1)."""
    import spada_sim_amd as S
    for kind, seed in ((S.GEN_WEBBASE_LIKE, 12347), (S.GEN_COP20K_LIKE, 12346), (S.GEN_MC2DEPI_LIKE, 12349)):
        m = S.generate(kind, 0, 0, seed)
        cap = S.count_products(m, m, 0, m.shape[0])
        eng = S.Engine()
        try:
            d = eng.upload(m)
            _, _, _, nnz1 = eng.fused_owned(d, d, 0, m.shape[0], cap)
            st1 = eng.stats()
            assert st1["pipeline_runs"] == 1, (kind, st1["pipeline_runs"])
            _, _, _, nnz2 = eng.fused_owned(d, d, 0, m.shape[0], cap)
            assert nnz1 == nnz2 and eng.stats()["pipeline_runs"] == 1
            eng.free(d)
        finally:
            eng.close()


@pytest.mark.gpu
def test_a_stalled_chain_gives_up_instead_of_hanging(monkeypatch):
    """Every wait on the one-pass chain is bounded (chain_gave_up): with a task that never publishes its count (test hook) the call
    returns SPADA_ERR_HIP after SPADA_CHAIN_TIMEOUT_MS instead of holding the GPU; a context without the hook computes the product."""
    import spada_sim_amd as S
    a = S.generate(S.GEN_UNIFORM, 60000, 6, 5)
    ref = oracle.spgemm_sortmerge(to_oracle(a), to_oracle(a))
    monkeypatch.setenv("SPADA_TEST_STALL_TASK", "7")
    monkeypatch.setenv("SPADA_CHAIN_TIMEOUT_MS", "200")
    monkeypatch.setenv("SPADA_AUTO", "0")
    eng = S.Engine()
    try:
        with pytest.raises(S.SpadaError) as ei:
            eng.spgemm_fused(a, a)
        assert "no progress" in str(ei.value)
    finally:
        eng.close()
    monkeypatch.delenv("SPADA_TEST_STALL_TASK")
    monkeypatch.delenv("SPADA_CHAIN_TIMEOUT_MS")
    eng = S.Engine()
    try:
        c = eng.spgemm_fused(a, a)
        assert_parity(c, ref, to_oracle(a), to_oracle(a), RTOL)
    finally:
        eng.close()


@pytest.mark.gpu
def test_moving_the_scratch_arrays_to_a_faster_place_keeps_the_product(monkeypatch):
    """A context that keeps scattering into the same scratch arrays tries other places for the column array (place_scratch: where the two
    arrays lie in physical memory decides between two regimes of the scatter, a tenth apart, and a probe with the scatter's store pattern
    tells them apart).  With the thresholds lowered (test hooks) the choice is made on R-MAT 14's few megabytes after two runs: every call
    before and after it gives the same C, through both entry points; SPADA_PLACE=0 leaves the arrays alone."""
    import ctypes
    import spada_sim_amd as S
    from spada_sim_amd import _ffi
    fn = _ffi.lib().spada_dev_scratch_placement
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    m = S.generate(S.GEN_RMAT, 14, 16, 4)
    a = to_oracle(m)
    ref = oracle.spgemm_sortmerge(a, a)
    monkeypatch.setenv("SPADA_PLACE_MIN", "1024")
    monkeypatch.setenv("SPADA_PLACE_AFTER", "2")
    for place in ("1", "0"):
        monkeypatch.setenv("SPADA_PLACE", place)
        eng = S.Engine()
        try:
            for call in range(5):
                if call % 2:
                    c = eng.spgemm(m, m)
                else:
                    c, st = fused(eng, m, m)
                    assert st["spill_rows"] > 0
                assert_parity(c, ref, a, a, RTOL)
            tried, first, kept = ctypes.c_uint32(0), ctypes.c_float(0), ctypes.c_float(0)
            assert fn(eng._ctx, ctypes.byref(tried), ctypes.byref(first), ctypes.byref(kept)) == 0
            if place == "1":
                assert tried.value <= 6 and 0 < kept.value <= first.value and (tried.value >= 1 or first.value <= 2.5)
            else:
                assert tried.value == 0
        finally:
            eng.close()
