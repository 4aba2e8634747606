"""ctypes access to the CPU oracle (oracle/spgemm_ref.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py.  Nothing under spada_sim_amd/ imports this module.
Parity status and reference citations are in the header of spgemm_ref.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_spgemm.so")
_SO_OVERRIDE = os.environ.get("SPADA_ORACLE_LIB_PATH")     # scripts/run_asan_tests.sh: the sanitizer build of this same source
_lib = None

_u64p = ctypes.POINTER(ctypes.c_uint64)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    """Compile liboracle_spgemm.so with gcc (idempotent)."""
    src = os.path.join(_HERE, "spgemm_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle_spgemm.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not _SO_OVERRIDE:
            build()
        L = ctypes.CDLL(_SO_OVERRIDE or _SO)
        L.oracle_count_products.restype = ctypes.c_uint64
        L.oracle_count_products.argtypes = [ctypes.c_uint64, _u64p, _u64p, _u64p]
        L.oracle_spgemm_sortmerge.restype = ctypes.c_uint64
        L.oracle_spgemm_sortmerge.argtypes = [ctypes.c_uint64, _u64p, _u64p, _f64p,
                                              _u64p, _u64p, _f64p, _u64p, _u64p, _f64p]
        L.oracle_spgemm_spa.restype = ctypes.c_uint64
        L.oracle_spgemm_spa.argtypes = [ctypes.c_uint64, ctypes.c_uint64, _u64p, _u64p, _f64p,
                                        _u64p, _u64p, _f64p, _u64p, _u64p, _f64p, ctypes.c_int]
        L.oracle_spgemm_windowed.restype = ctypes.c_uint64
        L.oracle_spgemm_windowed.argtypes = [ctypes.c_uint64, ctypes.c_uint64, _u64p, _u64p, _f64p,
                                             _u64p, _u64p, _f64p, _u64p, _u64p, _f64p]
        L.oracle_transpose_csr.restype = None
        L.oracle_transpose_csr.argtypes = [ctypes.c_uint64, ctypes.c_uint64, _u64p, _u64p, _f64p,
                                           _u64p, _u64p, _f64p]
        L.oracle_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    if a is None:
        return None
    if a.dtype == np.uint64:
        return a.ctypes.data_as(_u64p)
    return a.ctypes.data_as(_f64p)


class Csr:
    """Plain CSR triple in the reference's CsrMatStorage layout (storage.rs:150-160):
    indptr/indices uint64 (usize), data float64."""

    def __init__(self, rows, cols, indptr, indices, data):
        self.rows, self.cols = int(rows), int(cols)
        self.indptr, self.indices, self.data = _u64(indptr), _u64(indices), _f64(data)
        assert self.indptr.shape[0] == self.rows + 1
        assert self.indices.shape[0] == self.data.shape[0] == int(self.indptr[-1])

    @property
    def nnz(self):
        return int(self.indptr[-1])

    @classmethod
    def from_scipy(cls, m):
        m = m.tocsr()
        return cls(m.shape[0], m.shape[1], m.indptr, m.indices, m.data)

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.data, self.indices.astype(np.int64), self.indptr.astype(np.int64)),
                             shape=(self.rows, self.cols))


def count_products(a, b):
    return int(lib().oracle_count_products(a.rows, _p(a.indptr), _p(a.indices), _p(b.indptr)))


def spgemm_sortmerge(a, b):
    """C = A*B by the sort-merge restatement (single thread)."""
    assert a.cols == b.rows
    L = lib()
    c_indptr = np.zeros(a.rows + 1, dtype=np.uint64)
    args = (a.rows, _p(a.indptr), _p(a.indices), _p(a.data), _p(b.indptr), _p(b.indices), _p(b.data))
    nnz = L.oracle_spgemm_sortmerge(*args, _p(c_indptr), None, None)
    if nnz == 2**64 - 1:
        raise MemoryError("oracle allocation failed")
    c_indices = np.zeros(nnz, dtype=np.uint64)
    c_data = np.zeros(nnz, dtype=np.float64)
    L.oracle_spgemm_sortmerge(*args, _p(c_indptr), _p(c_indices), _p(c_data))
    return Csr(a.rows, b.cols, c_indptr, c_indices, c_data)


def spgemm_windowed(a, b, lane_num=8):
    """C = A*B with the products added in the reference's order of K-windows of `lane_num` A scalars, then first-in-first-out
    pairwise merges of the partial fibers (scheduler.rs:482-606, :381-480; see spgemm_ref.c).  Same structure as
    spgemm_sortmerge; values differ by floating-point re-association only."""
    assert a.cols == b.rows
    L = lib()
    c_indptr = np.zeros(a.rows + 1, dtype=np.uint64)
    args = (a.rows, lane_num, _p(a.indptr), _p(a.indices), _p(a.data), _p(b.indptr), _p(b.indices), _p(b.data))
    nnz = L.oracle_spgemm_windowed(*args, _p(c_indptr), None, None)
    if nnz == 2**64 - 1:
        raise MemoryError("oracle allocation failed")
    c_indices = np.zeros(nnz, dtype=np.uint64)
    c_data = np.zeros(nnz, dtype=np.float64)
    L.oracle_spgemm_windowed(*args, _p(c_indptr), _p(c_indices), _p(c_data))
    return Csr(a.rows, b.cols, c_indptr, c_indices, c_data)


def spgemm_spa(a, b, n_threads=0, symbolic_only=False):
    """C = A*B by the SPA restatement (OpenMP); bit-identical to spgemm_sortmerge."""
    assert a.cols == b.rows
    L = lib()
    c_indptr = np.zeros(a.rows + 1, dtype=np.uint64)
    args = (a.rows, b.cols, _p(a.indptr), _p(a.indices), _p(a.data), _p(b.indptr), _p(b.indices), _p(b.data))
    nnz = L.oracle_spgemm_spa(*args, _p(c_indptr), None, None, n_threads)
    if nnz == 2**64 - 1:
        raise MemoryError("oracle allocation failed")
    if symbolic_only:
        return c_indptr
    c_indices = np.zeros(nnz, dtype=np.uint64)
    c_data = np.zeros(nnz, dtype=np.float64)
    L.oracle_spgemm_spa(*args, _p(c_indptr), _p(c_indices), _p(c_data), n_threads)
    return Csr(a.rows, b.cols, c_indptr, c_indices, c_data)


def transpose(a):
    """Sorted CSR of A^T (gemm.rs:46)."""
    t_indptr = np.zeros(a.cols + 1, dtype=np.uint64)
    t_indices = np.zeros(a.nnz, dtype=np.uint64)
    t_data = np.zeros(a.nnz, dtype=np.float64)
    lib().oracle_transpose_csr(a.rows, a.cols, _p(a.indptr), _p(a.indices), _p(a.data),
                               _p(t_indptr), _p(t_indices), _p(t_data))
    return Csr(a.cols, a.rows, t_indptr, t_indices, t_data)


def from_mat(a):
    """(A, B) per gemm.rs:41-53: square -> B = A, else B = A^T."""
    return (a, a) if a.rows == a.cols else (a, transpose(a))


def num_threads():
    return int(lib().oracle_num_threads())
