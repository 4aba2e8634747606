"""Python mirror of the reference's host-side interface for the SpGEMM path, on top of the C ABI.

Same names and argument meaning as the Rust originals (citations into /root/reference/src) so that
tests read like tests of the reference:

    load_mm_mat(dir, name)                       py2rust.rs:62-97
    parse_config(path)                           frontend.rs:77-85
    GEMM.from_mat(name, mat)                     gemm.rs:41-53
    CsrMatStorage.init_with_gemm(gemm) -> (A, B) storage.rs:214-239
    CsrRow                                       storage.rs:34-126
    Simulator(...).execute() / get_exec_result() simulator.rs:431-507, :509-890, :1034-1062

Every array lives in numpy only as the caller-owned storage the ABI borrows (Vec<f64>/Vec<usize>);
all computation happens in libspada_spgemm.so.  The C++ twin used by the `spada-sim` binary is
spada_sim_amd/csrc/spada_host.hpp.
"""
import ctypes
import os

import numpy as np

from . import _ffi
from ._ffi import SpadaError, check

ACC_LDS_HASH = 0
ACC_SORT_MERGE = 1
REORDER_BY_LENGTH = 0      # sort_by_length, preprocessing.rs:76-89 (what -p does in the reference)
REORDER_BY_PRODUCTS = 1    # rows sorted by their number of products (GPU load balance)

GEN_RMAT, GEN_WEBBASE_LIKE, GEN_COP20K_LIKE, GEN_CAGE12_LIKE, GEN_MC2DEPI_LIKE, GEN_UNIFORM = range(6)


def _view(indptr, indices, data, rows, cols):
    v = _ffi.CsrView()
    v.rows, v.cols, v.nnz = int(rows), int(cols), int(indptr[-1]) if len(indptr) else 0
    v.indptr = indptr.ctypes.data_as(_ffi.u64p)
    v.indices = indices.ctypes.data_as(_ffi.u64p)
    v.data = data.ctypes.data_as(_ffi.f64p)
    return v


class CsMat:
    """Owning CSR container (what `sprs::CsMat<f64>` is to the reference): shape + three arrays."""

    def __init__(self, shape, indptr, indices, data):
        self.shape = (int(shape[0]), int(shape[1]))
        self.indptr = np.ascontiguousarray(indptr, dtype=np.uint64)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint64)
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        if self.indptr.shape[0] != self.shape[0] + 1:
            raise ValueError("indptr length must be rows + 1")

    def rows(self):
        return self.shape[0]

    def cols(self):
        return self.shape[1]

    def nnz(self):
        return int(self.indptr[-1])

    def view(self):
        return _view(self.indptr, self.indices, self.data, *self.shape)

    def validate(self):
        check(_ffi.lib().spada_csr_validate(ctypes.byref(self.view())))

    @classmethod
    def _from_handle(cls, h):
        """Copy a library-owned spada_host_csr into numpy arrays and free the handle."""
        L = _ffi.lib()
        v = _ffi.CsrView()
        check(L.spada_host_csr_view(h, ctypes.byref(v)))
        rows, cols, nnz = v.rows, v.cols, v.nnz
        indptr = np.ctypeslib.as_array(v.indptr, shape=(rows + 1,)).copy()
        indices = np.ctypeslib.as_array(v.indices, shape=(nnz,)).copy() if nnz else np.zeros(0, np.uint64)
        data = np.ctypeslib.as_array(v.data, shape=(nnz,)).copy() if nnz else np.zeros(0, np.float64)
        L.spada_host_csr_free(h)
        return cls((rows, cols), indptr, indices, data)

    def transpose(self):
        h = _ffi.vp()
        check(_ffi.lib().spada_transpose(ctypes.byref(self.view()), ctypes.byref(h)))
        return CsMat._from_handle(h)


def load_mm_mat(dir_path, gemm_nm):
    """<dir_path>/<gemm_nm>.mtx -> CsMat, as scipy.io.mmread(f).tocsr() (py2rust.rs:62-97)."""
    h = _ffi.vp()
    path = os.path.join(dir_path, gemm_nm + ".mtx")
    check(_ffi.lib().spada_mtx_read(path.encode(), ctypes.byref(h)))
    return CsMat._from_handle(h)


def write_mm_mat(path, mat):
    check(_ffi.lib().spada_mtx_write(str(path).encode(), ctypes.byref(mat.view())))


def write_bin(path, mat):
    """Binary CSR dump (SPADACSR version 1: header, the three arrays, checksums)."""
    check(_ffi.lib().spada_csr_write_bin(str(path).encode(), ctypes.byref(mat.view())))


def read_bin(path):
    h = _ffi.vp()
    check(_ffi.lib().spada_csr_read_bin(str(path).encode(), ctypes.byref(h)))
    return CsMat._from_handle(h)


def checksum(mat):
    """(dict, one-line text) -- the line `spada-sim --checksum` prints and the .mtx writer embeds."""
    cs = _ffi.Checksum()
    check(_ffi.lib().spada_csr_checksum(ctypes.byref(mat.view()), ctypes.byref(cs)))
    buf = ctypes.create_string_buffer(512)
    check(_ffi.lib().spada_checksum_format(ctypes.byref(cs), buf, 512))
    return {n: getattr(cs, n) for n, _ in cs._fields_}, buf.value.decode()


def generate(kind, p0=0, p1=0, seed=0):
    h = _ffi.vp()
    check(_ffi.lib().spada_generate(int(kind), int(p0), int(p1), int(seed), ctypes.byref(h)))
    return CsMat._from_handle(h)


def parse_config(config_fp):
    """13 required keys of OmegaConfig (frontend.rs:8-23); returns a dict."""
    cfg = _ffi.Config()
    check(_ffi.lib().spada_config_parse(str(config_fp).encode(), ctypes.byref(cfg)))
    d = {}
    for name, _ in cfg._fields_:
        v = getattr(cfg, name)
        if isinstance(v, bytes):
            v = v.decode()
        elif hasattr(v, "__len__"):
            v = list(v)
        d[name] = v
    return d


class GEMM:
    """Workload container {name, a, b} (gemm.rs:26-53)."""

    def __init__(self, name, a, b):
        self.name, self.a, self.b = name, a, b

    @classmethod
    def from_mat(cls, mn, mat):
        # square -> A * A (B is the same object); otherwise A * A^T
        h = _ffi.vp()
        same = ctypes.c_int(0)
        check(_ffi.lib().spada_from_mat(ctypes.byref(mat.view()), ctypes.byref(h), ctypes.byref(same)))
        return cls(mn, mat, mat if same.value else CsMat._from_handle(h))


class CsrRow:
    """One fiber: rowptr, column indices (`indptr` in the reference's naming) and values (storage.rs:34-40)."""

    def __init__(self, rowptr, data=None, indptr=None):
        self.rowptr = int(rowptr)
        self.data = np.zeros(0) if data is None else data
        self.indptr = np.zeros(0, np.uint64) if indptr is None else indptr

    def len(self):
        return len(self.indptr)

    def size(self):
        return 2 * len(self.indptr)

    def __str__(self):   # storage.rs:115-126
        n = min(len(self.data), 5)
        cols = ", ".join(str(int(c)) for c in self.indptr[:n])
        vals = ", ".join(_rust_f64(v) for v in self.data[:n])
        return f"rowptr: {self.rowptr} indptr: [{cols}] data: [{vals}]"


def _rust_f64(v):
    """Rust's {:?} for f64: shortest round-trip, always with a decimal point or exponent."""
    s = repr(float(v))
    if "e" in s or "inf" in s or "nan" in s:
        return s
    return s if "." in s else s + ".0"


class CsrMatStorage:
    """data / indptr / indices + mat_shape = [cols, rows] (storage.rs:150-160, :214-239)."""

    def __init__(self, mat):
        self.data, self.indptr, self.indices = mat.data, mat.indptr, mat.indices
        self.mat_shape = [mat.shape[1], mat.shape[0]]
        self.read_count = 0
        self.write_count = 0
        self.remapped = False
        self.row_remap = {}
        self._mat = mat

    @classmethod
    def init_with_gemm(cls, gemm):
        a = cls(gemm.a)
        return a, (a if gemm.b is gemm.a else cls(gemm.b))

    def row_num(self):
        return len(self.indptr) - 1

    def reorder_row(self, rowmap):
        """storage.rs:252-255: mark the storage as remapped.  `rowmap` is what sort_by_length / sort_by_products returned: the
        order itself is computed on the device when the simulator executes, row_remap holds it afterwards."""
        self.remapped = True
        self._reorder_key = rowmap.key
        self.row_remap = rowmap

    def view(self):
        return self._mat.view()

    def read_row(self, row_ptr):
        if row_ptr >= len(self.indptr) - 1:
            raise SpadaError(1, f"Invalid row_ptr: {row_ptr}")
        s, t = int(self.indptr[row_ptr]), int(self.indptr[row_ptr + 1])
        return CsrRow(row_ptr, self.data[s:t], self.indices[s:t])


class RowMap(dict):
    """Result of sort_by_length / sort_by_products: new position -> original row (HashMap<usize, usize> in the reference).  It
    is filled by the device pre-pass of the run that uses it."""

    def __init__(self, key):
        super().__init__()
        self.key = key


def sort_by_length(amat):
    """preprocessing.rs:76-89: rows of A in ascending order of their length (stable)."""
    return RowMap(REORDER_BY_LENGTH)


def sort_by_products(amat):
    """Like sort_by_length, by the number of products of a row (sum of the lengths of the B rows it selects)."""
    return RowMap(REORDER_BY_PRODUCTS)


class Engine:
    """Owns one spada_ctx (one GPU).  Raises SpadaError(NO_DEVICE) without a gfx950 device."""

    def __init__(self, device=-1, accumulator=ACC_LDS_HASH):
        self._L = _ffi.lib()
        self._ctx = _ffi.vp()
        opts = _ffi.Options(ctypes.sizeof(_ffi.Options), device, accumulator, 0)
        check(self._L.spada_create(ctypes.byref(opts), ctypes.byref(self._ctx)))

    def close(self):
        if self._ctx:
            self._L.spada_destroy(self._ctx)
            self._ctx = _ffi.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host-pointer two-phase path (the drop-in seam) --
    def spgemm(self, a, b, reorder=None):
        """C = A * B through spada_spgemm_symbolic / spada_spgemm_numeric; returns a CsMat.  reorder = REORDER_BY_LENGTH /
        REORDER_BY_PRODUCTS: the -p pre-pass (rows of A reordered on the device, product mapped back; C is the same);
        the row map is left in self.last_rowmap."""
        nnz = ctypes.c_uint64(0)
        va = a.view()
        vb = va if b is a else b.view()
        if reorder is not None:
            self.last_rowmap = np.zeros(a.shape[0], np.uint64)
            check(self._L.spada_spgemm_symbolic_reordered(self._ctx, ctypes.byref(va), ctypes.byref(vb), int(reorder),
                                                          ctypes.byref(nnz), self.last_rowmap.ctypes.data_as(_ffi.u64p)))
        else:
            check(self._L.spada_spgemm_symbolic(self._ctx, ctypes.byref(va), ctypes.byref(vb), ctypes.byref(nnz)))
        n = nnz.value
        c_indptr = np.zeros(a.shape[0] + 1, np.uint64)
        c_indices = np.zeros(n, np.uint64)
        c_data = np.zeros(n, np.float64)
        check(self._L.spada_spgemm_numeric(self._ctx, c_indptr.ctypes.data_as(_ffi.u64p),
                                           c_indices.ctypes.data_as(_ffi.u64p), c_data.ctypes.data_as(_ffi.f64p)))
        return CsMat((a.shape[0], b.shape[1]), c_indptr, c_indices, c_data)

    # -- device-resident path --
    def upload(self, mat):
        h = _ffi.vp()
        check(self._L.spada_dev_csr_upload(self._ctx, ctypes.byref(mat.view()), ctypes.byref(h)))
        return h

    def free(self, h):
        self._L.spada_dev_csr_free(self._ctx, h)

    def symbolic(self, da, db, row_begin, row_end):
        nnz = ctypes.c_uint64(0)
        check(self._L.spada_dev_spgemm_symbolic(self._ctx, da, db, row_begin, row_end, ctypes.byref(nnz)))
        return nnz.value

    def numeric(self, d_indptr, d_indices, d_data):
        """Device pointers (ints) of caller-allocated u64[rows+1], u32[nnz], f64[nnz]."""
        check(self._L.spada_dev_spgemm_numeric(self._ctx, _ffi.vp(d_indptr), _ffi.vp(d_indices), _ffi.vp(d_data)))

    def fused(self, da, db, row_begin, row_end, d_indptr, d_indices, d_data, capacity):
        """One-pass SpGEMM into caller buffers of `capacity` entries (any upper bound of nnz(C)); returns nnz(C)."""
        nnz = ctypes.c_uint64(0)
        check(self._L.spada_dev_spgemm_fused(self._ctx, da, db, row_begin, row_end, _ffi.vp(d_indptr), _ffi.vp(d_indices),
                                             _ffi.vp(d_data), capacity, ctypes.byref(nnz)))
        return nnz.value

    def fused_owned(self, da, db, row_begin, row_end, capacity):
        """One-pass SpGEMM into context-owned device buffers; returns (d_indptr, d_indices, d_data, nnz)."""
        p, i, v, nnz = _ffi.vp(), _ffi.vp(), _ffi.vp(), ctypes.c_uint64(0)
        check(self._L.spada_dev_spgemm_fused_owned(self._ctx, da, db, row_begin, row_end, capacity, ctypes.byref(p),
                                                   ctypes.byref(i), ctypes.byref(v), ctypes.byref(nnz)))
        return p.value, i.value, v.value, nnz.value

    def spgemm_fused(self, a, b, capacity=None):
        """C = A * B in one pass through spada_spgemm_fused (host pointers); capacity defaults to the product count."""
        if capacity is None:
            capacity = count_products(a, b, 0, a.shape[0])
        nnz = ctypes.c_uint64(0)
        va = a.view()
        vb = va if b is a else b.view()
        c_indptr = np.zeros(a.shape[0] + 1, np.uint64)
        c_indices = np.zeros(max(capacity, 1), np.uint64)
        c_data = np.zeros(max(capacity, 1), np.float64)
        check(self._L.spada_spgemm_fused(self._ctx, ctypes.byref(va), ctypes.byref(vb), capacity,
                                         c_indptr.ctypes.data_as(_ffi.u64p), c_indices.ctypes.data_as(_ffi.u64p),
                                         c_data.ctypes.data_as(_ffi.f64p), ctypes.byref(nnz)))
        n = nnz.value
        return CsMat((a.shape[0], b.shape[1]), c_indptr, c_indices[:n].copy(), c_data[:n].copy())

    def numeric_owned(self):
        p, i, v = _ffi.vp(), _ffi.vp(), _ffi.vp()
        check(self._L.spada_dev_spgemm_numeric_owned(self._ctx, ctypes.byref(p), ctypes.byref(i), ctypes.byref(v)))
        return p.value, i.value, v.value

    def download(self, d_indptr, d_indices, d_data, rows, nnz, cols):
        c_indptr = np.zeros(rows + 1, np.uint64)
        c_indices = np.zeros(nnz, np.uint64)
        c_data = np.zeros(nnz, np.float64)
        check(self._L.spada_dev_download_c(self._ctx, _ffi.vp(d_indptr), _ffi.vp(d_indices), _ffi.vp(d_data), rows, nnz,
                                           c_indptr.ctypes.data_as(_ffi.u64p), c_indices.ctypes.data_as(_ffi.u64p),
                                           c_data.ctypes.data_as(_ffi.f64p)))
        return CsMat((rows, cols), c_indptr, c_indices, c_data)

    def spgemm_row_chunks(self, da, db, bounds, alloc, consume):
        """C = A * B one A-row chunk at a time (bounds[i] .. bounds[i + 1]), for products too large to hold at once
        (R-MAT scale 22: nnz(C) x 12 B exceeds one GPU's HBM).  `alloc(rows, nnz)` returns three device pointers
        (u64[rows + 1], u32[nnz], f64[nnz]); `consume(row_begin, row_end, nnz, stats)` sees the finished chunk before
        the next one overwrites nothing of it -- the caller owns the buffers.  Returns the total nnz(C)."""
        total = 0
        for i in range(len(bounds) - 1):
            b0, b1 = int(bounds[i]), int(bounds[i + 1])
            if b1 <= b0:
                continue
            nnz = self.symbolic(da, db, b0, b1)
            p, ix, v = alloc(b1 - b0, nnz)
            self.numeric(p, ix, v)
            consume(b0, b1, nnz, self.stats())
            total += nnz
        return total

    def fused_row_chunks(self, da, db, bounds, caps, alloc, consume):
        """The one-pass variant of spgemm_row_chunks: every chunk through spada_dev_spgemm_fused.  `caps[i]` is the capacity of
        the C buffers of chunk i -- an upper bound of its nnz(C) the caller knows beforehand (its product count: count_products);
        `alloc(rows, cap)` and `consume(row_begin, row_end, nnz, stats)` as above.  Returns the total nnz(C)."""
        total = 0
        for i in range(len(bounds) - 1):
            b0, b1 = int(bounds[i]), int(bounds[i + 1])
            if b1 <= b0:
                continue
            p, ix, v = alloc(b1 - b0, int(caps[i]))
            nnz = self.fused(da, db, b0, b1, p, ix, v, int(caps[i]))
            consume(b0, b1, nnz, self.stats())
            total += nnz
        return total

    def set_phase_timing(self, enabled):
        """Event records between the small kernels of a call (ms_row_stats / ms_big_expand / ms_cut) on or off: each idles the
        stream for about 5 us (spada_set_phase_timing)."""
        check(self._L.spada_set_phase_timing(self._ctx, 1 if enabled else 0))

    def stats(self):
        st = _ffi.Stats()
        check(self._L.spada_get_stats(self._ctx, ctypes.byref(st)))
        return st.as_dict()

    def stats_raw(self, st=None):
        """The statistics of the last call as the C structure itself (spada_stats; fields as in stats()): no conversion -- for loops
        that time calls (bench.py), where building the dictionary of stats() costs as much host time as a small call takes."""
        st = st if st is not None else _ffi.Stats()
        check(self._L.spada_get_stats(self._ctx, ctypes.byref(st)))
        return st


class Comm:
    """One RCCL communicator per process / GPU (libspada_comm.so, include/spada_comm.h): the allgatherv of the C row blocks.
    `unique_id()` is called on rank 0 and its 128 bytes are handed to the other ranks by the launcher."""

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(_ffi.COMM_ID_BYTES)
        check(_ffi.comm_lib().spada_comm_get_unique_id(buf))
        return buf.raw

    def __init__(self, uid, rank, nranks, device):
        self._L = _ffi.comm_lib()
        self._h = _ffi.vp()
        self.rank, self.nranks = int(rank), int(nranks)
        check(self._L.spada_comm_create(ctypes.c_char_p(bytes(uid)), self.rank, self.nranks, int(device), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            self._L.spada_comm_destroy(self._h)
            self._h = _ffi.vp()

    def allgather_counts(self, rows, nnz):
        r = np.zeros(self.nranks, np.uint64)
        n = np.zeros(self.nranks, np.uint64)
        check(self._L.spada_comm_allgather_counts(self._h, int(rows), int(nnz), r.ctypes.data_as(_ffi.u64p), n.ctypes.data_as(_ffi.u64p)))
        return r, n

    def allgatherv_c(self, d_indptr, d_indices, d_data, rows, nnz, d_full_indptr, d_full_indices, d_full_data):
        """Device pointers (ints); rows / nnz = the arrays allgather_counts returned."""
        check(self._L.spada_comm_allgatherv_c(self._h, _ffi.vp(d_indptr), _ffi.vp(d_indices), _ffi.vp(d_data),
                                              rows.ctypes.data_as(_ffi.u64p), nnz.ctypes.data_as(_ffi.u64p),
                                              _ffi.vp(d_full_indptr), _ffi.vp(d_full_indices), _ffi.vp(d_full_data)))

    def dist_symbolic(self, engine, da, db, row_begin, row_end, chunks):
        r = np.zeros(self.nranks, np.uint64)
        n = np.zeros(self.nranks, np.uint64)
        check(self._L.spada_dist_spgemm_symbolic(engine._ctx, self._h, da, db, int(row_begin), int(row_end), int(chunks),
                                                 r.ctypes.data_as(_ffi.u64p), n.ctypes.data_as(_ffi.u64p)))
        return r, n

    def dist_numeric(self, engine, d_full_indptr, d_full_indices, d_full_data):
        check(self._L.spada_dist_spgemm_numeric(engine._ctx, self._h, _ffi.vp(d_full_indptr), _ffi.vp(d_full_indices),
                                                _ffi.vp(d_full_data)))


def comm_plan(rows, nnz, chunk_pos=None):
    """spada_comm_plan: (row_off, nnz_off, piece_begin, piece_count) of the exchange from the gathered counts; chunk_pos =
    array [nranks, chunks + 1] of piece positions inside every block, or None.  Host arithmetic only."""
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    nnz = np.ascontiguousarray(nnz, dtype=np.uint64)
    n = len(rows)
    row_off, nnz_off = np.zeros(n + 1, np.uint64), np.zeros(n + 1, np.uint64)
    if chunk_pos is None:
        check(_ffi.comm_lib().spada_comm_plan(n, rows.ctypes.data_as(_ffi.u64p), nnz.ctypes.data_as(_ffi.u64p), 0, None,
                                              row_off.ctypes.data_as(_ffi.u64p), nnz_off.ctypes.data_as(_ffi.u64p), None, None))
        return row_off, nnz_off, None, None
    pos = np.ascontiguousarray(chunk_pos, dtype=np.uint64)
    k = pos.shape[1] - 1
    pb, pc = np.zeros((n, k), np.uint64), np.zeros((n, k), np.uint64)
    check(_ffi.comm_lib().spada_comm_plan(n, rows.ctypes.data_as(_ffi.u64p), nnz.ctypes.data_as(_ffi.u64p), k,
                                          pos.ctypes.data_as(_ffi.u64p), row_off.ctypes.data_as(_ffi.u64p),
                                          nnz_off.ctypes.data_as(_ffi.u64p), pb.ctypes.data_as(_ffi.u64p), pc.ctypes.data_as(_ffi.u64p)))
    return row_off, nnz_off, pb, pc


class Simulator:
    """Drop-in for the reference's Simulator: same constructor arguments (simulator.rs:431-448); the
    accelerator-model parameters are accepted and kept but do not steer the GPU kernels."""

    def __init__(self, pe_num, at_num, lane_num, cache_size, word_byte, output_base_addr, default_block_shape,
                 a_matrix, b_matrix, psum_matrix=None, accelerator="Spada", mem_latency=0, cache_latency=0, freq=1.0,
                 channel=1, bandwidth_per_channel=1.0, engine=None):
        self.params = dict(pe_num=pe_num, at_num=at_num, lane_num=lane_num, cache_size=cache_size, word_byte=word_byte,
                           output_base_addr=output_base_addr, default_block_shape=default_block_shape,
                           accelerator=accelerator, mem_latency=mem_latency, cache_latency=cache_latency, freq=freq,
                           channel=channel, bandwidth_per_channel=bandwidth_per_channel)
        self.a_matrix, self.b_matrix = a_matrix, b_matrix
        self._engine = engine or Engine()
        self._c = None
        self._stats = None

    def execute(self):
        a, b = self.a_matrix._mat, self.b_matrix._mat
        key = getattr(self.a_matrix, "_reorder_key", None) if getattr(self.a_matrix, "remapped", False) else None
        self._c = self._engine.spgemm(a, a if self.b_matrix is self.a_matrix else b, reorder=key)
        if key is not None:     # row_remap as the reference keeps it (storage.rs:157); the result rows are already mapped back
            self.a_matrix.row_remap.update({i: int(r) for i, r in enumerate(self._engine.last_rowmap)})
        self._stats = self._engine.stats()

    def get_exec_result(self):
        """One CsrRow per A row, ascending (simulator.rs:1034-1062)."""
        c = self._c
        out = []
        for r in range(c.shape[0]):
            s, t = int(c.indptr[r]), int(c.indptr[r + 1])
            out.append(CsrRow(r, c.data[s:t], c.indices[s:t]))
        return out

    def result_matrix(self):
        return self._c

    # The reference reports simulated word counts; here they are the measured algorithmic traffic in
    # words of `word_byte` bytes (documented re-definition).
    def get_a_mat_stat(self):
        st, w = self._stats, self.params["word_byte"]
        return [(st["a_nnz"] * 12 + (st["rows"] + 1) * 8) // w, 0]

    def get_b_mat_stat(self):
        st, w = self._stats, self.params["word_byte"]
        return [(st["a_nnz"] * 16 + st["nprod"] * 12) // w, 0]

    def get_c_mat_stat(self):
        st, w = self._stats, self.params["word_byte"]
        return [0, st["bytes_write"] // w]

    def get_exec_cycle(self):
        st = self._stats
        return int((st["ms_symbolic_call"] + st["ms_numeric_call"]) * 1e6 * self.params["freq"])

    def get_cache_stat(self):
        return [0, 0]


ACCELERATORS = {"ip": 0, "op": 1, "multirow": 2, "spada": 3}   # frontend.rs:26-33, case-insensitive


class CycleModel:
    """The cycle-level Spada model (include/spada_cycle.h): what `accuratesimu` simulates -- the product in
    the accelerator's order of operations plus the counters main.rs:97-108 prints.  Host code; parity with the Rust tool is
    unpinned (see the header)."""

    def __init__(self, a, b, config, accelerator="spada", row_remap=None):
        """a, b: CsMat; config: dict with the keys of the JSON configuration (parse_config)."""
        cfg = _ffi.CycleConfig()
        cfg.struct_size = ctypes.sizeof(cfg)
        for k in ("pe_num", "at_num", "lane_num", "cache_size", "word_byte", "mem_latency", "cache_latency", "channel"):
            setattr(cfg, k, int(config[k]))
        cfg.block_shape[0], cfg.block_shape[1] = int(config["block_shape"][0]), int(config["block_shape"][1])
        cfg.freq = float(config["freq"])
        cfg.bandwidth_per_channel = float(config["bandwidth_per_channel"])
        cfg.accelerator = ACCELERATORS[str(accelerator).lower()]
        self._a, self._b = a, b     # borrowed by the model
        self._remap = None if row_remap is None else np.ascontiguousarray(row_remap, dtype=np.uint64)
        self._h = _ffi.vp()
        self._L = _ffi.lib()
        va, vb = a.view(), b.view()
        rp = None if self._remap is None else self._remap.ctypes.data_as(_ffi.u64p)
        check(self._L.spada_cycle_create(ctypes.byref(cfg), ctypes.byref(va), ctypes.byref(vb), rp, ctypes.byref(self._h)))

    def execute(self, max_cycles=0):
        check(self._L.spada_cycle_execute(self._h, int(max_cycles)))
        return self

    def counts(self):
        c = _ffi.CycleCounts()
        c.struct_size = ctypes.sizeof(c)
        check(self._L.spada_cycle_get_counts(self._h, ctypes.byref(c)))
        return {name: int(getattr(c, name)) for name, _ in c._fields_ if name != "struct_size"}

    def result(self):
        n = self._a.shape[0]
        nnz = self.counts()["c_nnz"]
        indptr = np.zeros(n + 1, np.uint64)
        indices = np.zeros(max(nnz, 1), np.uint64)
        data = np.zeros(max(nnz, 1), np.float64)
        check(self._L.spada_cycle_get_result(self._h, indptr.ctypes.data_as(_ffi.u64p), indices.ctypes.data_as(_ffi.u64p),
                                             data.ctypes.data_as(_ffi.f64p)))
        return CsMat((n, self._b.shape[1]), indptr, indices[:nnz], data[:nnz])

    def close(self):
        if self._h:
            self._L.spada_cycle_destroy(self._h)
            self._h = _ffi.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def partition_rows(a, b, nparts):
    """A-row block boundaries balanced on per-row products (scheduler.rs:296-379 issues disjoint row blocks)."""
    bounds = np.zeros(nparts + 1, np.uint64)
    va = a.view()
    vb = va if b is a else b.view()
    check(_ffi.lib().spada_partition_rows(ctypes.byref(va), ctypes.byref(vb), nparts, bounds.ctypes.data_as(_ffi.u64p)))
    return [int(x) for x in bounds]


def count_products(a, b, row_begin=0, row_end=None):
    n = ctypes.c_uint64(0)
    va = a.view()
    vb = va if b is a else b.view()
    check(_ffi.lib().spada_count_products(ctypes.byref(va), ctypes.byref(vb), row_begin,
                                          a.shape[0] if row_end is None else row_end, ctypes.byref(n)))
    return n.value


def device_count():
    return _ffi.lib().spada_device_count()
