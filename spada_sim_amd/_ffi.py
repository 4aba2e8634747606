"""ctypes binding of libspada_spgemm.so (include/spada_ffi.h).

The library is the product; this module only declares its entry points.  It fails loudly when
the shared object is missing -- there is no Python or CPU stand-in for the HIP path.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPADA_LIB_PATH") or os.path.join(_HERE, "lib", "libspada_spgemm.so")   # env: development A/B builds

u64 = ctypes.c_uint64
u64p = ctypes.POINTER(ctypes.c_uint64)
f64p = ctypes.POINTER(ctypes.c_double)
vp = ctypes.c_void_p

STATUS_NAMES = {0: "OK", 1: "INVALID", 2: "NO_DEVICE", 3: "HIP", 4: "OOM", 5: "IO", 6: "PARSE", 7: "STATE",
                8: "UNSUPPORTED", 9: "CAPACITY"}


class SpadaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"SPADA_ERR_{STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code


class CsrView(ctypes.Structure):
    _fields_ = [("rows", u64), ("cols", u64), ("nnz", u64), ("indptr", u64p), ("indices", u64p), ("data", f64p)]


class Options(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("accumulator", ctypes.c_int32),
                ("flags", ctypes.c_int32)]


class Stats(ctypes.Structure):
    _fields_ = [("rows", u64), ("a_nnz", u64), ("b_nnz", u64), ("nprod", u64), ("c_nnz", u64),
                ("bytes_read", u64), ("bytes_write", u64),
                ("ms_symbolic_call", ctypes.c_double), ("ms_numeric_call", ctypes.c_double), ("ms_fused_call", ctypes.c_double),
                ("ms_row_stats", ctypes.c_double), ("ms_big_expand", ctypes.c_double), ("ms_cut", ctypes.c_double),
                ("ms_task", ctypes.c_double), ("cls_rows", u64 * 8), ("cls_prod", u64 * 8), ("n_tasks", u64),
                ("multi_pass_tasks", u64), ("scratch_products", u64), ("spill_rows", u64), ("pipeline_runs", u64),
                ("workspace_bytes", u64), ("task_product_limit", u64), ("chain_fallbacks", u64), ("pipeline_kind", u64),
                ("ms_wall_call", ctypes.c_double)]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if hasattr(v, "__len__") else v
        return d


class Checksum(ctypes.Structure):
    _fields_ = [("rows", u64), ("cols", u64), ("nnz", u64), ("structure_hash", u64), ("value_hash", u64),
                ("value_sum", ctypes.c_double), ("value_abs_sum", ctypes.c_double)]


class Config(ctypes.Structure):
    _fields_ = [("ss_filepath", ctypes.c_char * 1024), ("nn_filepath", ctypes.c_char * 1024),
                ("pe_num", u64), ("at_num", u64), ("lane_num", u64), ("cache_size", u64), ("word_byte", u64),
                ("block_shape", u64 * 2), ("mem_latency", u64), ("cache_latency", u64),
                ("freq", ctypes.c_float), ("channel", u64), ("bandwidth_per_channel", ctypes.c_float),
                ("gpus", ctypes.c_uint32), ("accumulator", ctypes.c_int32), ("repeat", ctypes.c_uint32)]


class CycleConfig(ctypes.Structure):   # include/spada_cycle.h
    _fields_ = [("struct_size", u64), ("pe_num", u64), ("at_num", u64), ("lane_num", u64), ("cache_size", u64),
                ("word_byte", u64), ("block_shape", u64 * 2), ("mem_latency", u64), ("cache_latency", u64),
                ("freq", ctypes.c_float), ("channel", u64), ("bandwidth_per_channel", ctypes.c_float),
                ("accelerator", ctypes.c_int32), ("pad", ctypes.c_int32)]


class CycleCounts(ctypes.Structure):
    _fields_ = [("struct_size", u64), ("exec_cycles", u64), ("raw_cycles", u64), ("a_read", u64), ("a_write", u64),
                ("b_read", u64), ("b_write", u64), ("c_read", u64), ("c_write", u64), ("cache_read", u64),
                ("cache_write", u64), ("cache_miss", u64), ("b_evict", u64), ("psum_evict", u64), ("blocks", u64),
                ("windows", u64), ("pe_merge_tasks", u64), ("tree_merge_tasks", u64), ("c_nnz", u64)]


# name -> (restype, argtypes); every symbol include/spada_ffi.h and include/spada_cycle.h declare
SIGNATURES = {
    "spada_cycle_create": (ctypes.c_int, [ctypes.POINTER(CycleConfig), ctypes.POINTER(CsrView), ctypes.POINTER(CsrView), u64p,
                                          ctypes.POINTER(vp)]),
    "spada_cycle_execute": (ctypes.c_int, [vp, u64]),
    "spada_cycle_get_counts": (ctypes.c_int, [vp, ctypes.POINTER(CycleCounts)]),
    "spada_cycle_get_result": (ctypes.c_int, [vp, u64p, u64p, f64p]),
    "spada_cycle_destroy": (None, [vp]),
    "spada_last_error": (ctypes.c_char_p, []),
    "spada_abi_version": (ctypes.c_int, []),
    "spada_device_count": (ctypes.c_int, []),
    "spada_create": (ctypes.c_int, [ctypes.POINTER(Options), ctypes.POINTER(vp)]),
    "spada_destroy": (None, [vp]),
    "spada_spgemm_symbolic": (ctypes.c_int, [vp, ctypes.POINTER(CsrView), ctypes.POINTER(CsrView), u64p]),
    "spada_spgemm_symbolic_reordered": (ctypes.c_int, [vp, ctypes.POINTER(CsrView), ctypes.POINTER(CsrView), ctypes.c_int, u64p, u64p]),
    "spada_spgemm_numeric": (ctypes.c_int, [vp, u64p, u64p, f64p]),
    "spada_dev_csr_upload": (ctypes.c_int, [vp, ctypes.POINTER(CsrView), ctypes.POINTER(vp)]),
    "spada_dev_csr_free": (None, [vp, vp]),
    "spada_dev_csr_reorder": (ctypes.c_int, [vp, vp, vp, ctypes.c_int, ctypes.POINTER(vp)]),
    "spada_dev_csr_rowmap": (ctypes.c_int, [vp, vp, u64p]),
    "spada_dev_unpermute_c": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "spada_dev_spgemm_symbolic": (ctypes.c_int, [vp, vp, vp, u64, u64, u64p]),
    "spada_dev_spgemm_numeric": (ctypes.c_int, [vp, vp, vp, vp]),
    "spada_dev_spgemm_numeric_plan": (ctypes.c_int, [vp, ctypes.c_uint32, u64p]),
    "spada_dev_spgemm_numeric_chunk": (ctypes.c_int, [vp, ctypes.c_uint32, vp, vp, ctypes.POINTER(vp)]),
    "spada_dev_spgemm_indptr": (ctypes.c_int, [vp, vp]),
    "spada_dev_synchronize": (ctypes.c_int, [vp]),
    "spada_dev_spgemm_fused": (ctypes.c_int, [vp, vp, vp, u64, u64, vp, vp, vp, u64, u64p]),
    "spada_dev_spgemm_fused_owned": (ctypes.c_int, [vp, vp, vp, u64, u64, u64, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                                    ctypes.POINTER(vp), u64p]),
    "spada_spgemm_fused": (ctypes.c_int, [vp, ctypes.POINTER(CsrView), ctypes.POINTER(CsrView), u64, u64p, u64p, f64p, u64p]),
    "spada_dev_spgemm_numeric_owned": (ctypes.c_int, [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]),
    "spada_dev_download_c": (ctypes.c_int, [vp, vp, vp, vp, u64, u64, u64p, u64p, f64p]),
    "spada_get_stats": (ctypes.c_int, [vp, ctypes.POINTER(Stats)]),
    "spada_set_phase_timing": (ctypes.c_int, [vp, ctypes.c_int]),
    "spada_mtx_read": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(vp)]),
    "spada_mtx_write": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(CsrView)]),
    "spada_csr_checksum": (ctypes.c_int, [ctypes.POINTER(CsrView), ctypes.POINTER(Checksum)]),
    "spada_checksum_format": (ctypes.c_int, [ctypes.POINTER(Checksum), ctypes.c_char_p, ctypes.c_size_t]),
    "spada_csr_write_bin": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(CsrView)]),
    "spada_csr_read_bin": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(vp)]),
    "spada_host_csr_from_view": (ctypes.c_int, [ctypes.POINTER(CsrView), ctypes.POINTER(vp)]),
    "spada_host_csr_view": (ctypes.c_int, [vp, ctypes.POINTER(CsrView)]),
    "spada_host_csr_free": (None, [vp]),
    "spada_transpose": (ctypes.c_int, [ctypes.POINTER(CsrView), ctypes.POINTER(vp)]),
    "spada_from_mat": (ctypes.c_int, [ctypes.POINTER(CsrView), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int)]),
    "spada_csr_validate": (ctypes.c_int, [ctypes.POINTER(CsrView)]),
    "spada_partition_rows": (ctypes.c_int, [ctypes.POINTER(CsrView), ctypes.POINTER(CsrView), ctypes.c_uint32, u64p]),
    "spada_count_products": (ctypes.c_int, [ctypes.POINTER(CsrView), ctypes.POINTER(CsrView), u64, u64, u64p]),
    "spada_generate": (ctypes.c_int, [ctypes.c_int, u64, u64, u64, ctypes.POINTER(vp)]),
    "spada_config_parse": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(Config)]),
}

# libspada_comm.so (include/spada_comm.h): the RCCL exchange; loaded on demand, it pulls in librccl
COMM_LIB_PATH = os.environ.get("SPADA_COMM_LIB_PATH") or os.path.join(os.path.dirname(LIB_PATH), "libspada_comm.so")
COMM_ID_BYTES = 128
COMM_SIGNATURES = {
    "spada_comm_get_unique_id": (ctypes.c_int, [vp]),
    "spada_comm_create": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]),
    "spada_comm_destroy": (None, [vp]),
    "spada_comm_rank": (ctypes.c_int, [vp]),
    "spada_comm_size": (ctypes.c_int, [vp]),
    "spada_comm_allgather_counts": (ctypes.c_int, [vp, u64, u64, u64p, u64p]),
    "spada_comm_plan": (ctypes.c_int, [ctypes.c_int, u64p, u64p, ctypes.c_uint32, u64p, u64p, u64p, u64p, u64p]),
    "spada_comm_allgatherv_c": (ctypes.c_int, [vp, vp, vp, vp, u64p, u64p, vp, vp, vp]),
    "spada_dist_spgemm_symbolic": (ctypes.c_int, [vp, vp, vp, vp, u64, u64, ctypes.c_uint32, u64p, u64p]),
    "spada_dist_spgemm_numeric": (ctypes.c_int, [vp, vp, vp, vp, vp]),
}

_lib = None
_comm_lib = None


def comm_lib():
    """Load libspada_comm.so (once).  Raises ImportError if it has not been built."""
    global _comm_lib
    if _comm_lib is None:
        lib()
        if not os.path.exists(COMM_LIB_PATH):
            raise ImportError(f"{COMM_LIB_PATH} is missing: build it with `make -C spada_sim_amd/csrc`")
        L = ctypes.CDLL(COMM_LIB_PATH)
        for name, (res, args) in COMM_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _comm_lib = L
    return _comm_lib


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels carry their own libamdhip64.so (soname libamdhip64.so.7, asked for by
    torch as plain `libamdhip64.so`): if this package's library pulls in /opt/rocm's copy first, a later `import torch` loads
    the wheel's copy as a SECOND runtime, which then finds no GPU ("No HIP GPUs are available").  The other order is harmless
    (our NEEDED entry matches the soname of the copy torch loaded).  So when a torch wheel with its own runtime is installed, its
    copy is mapped first -- without importing torch.  SPADA_HIP_RUNTIME=system switches this off."""
    if os.environ.get("SPADA_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        for root in (spec.submodule_search_locations if spec is not None else []):
            cand = os.path.join(root, "lib", "libamdhip64.so")
            if os.path.exists(cand):
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
                return
    except (ImportError, OSError, ValueError):
        pass   # no torch, or an unusual layout: the system runtime is used


def lib():
    """Load the shared library (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C spada_sim_amd/csrc`).  spada_sim_amd has no fallback path.")
        _share_torch_hip_runtime()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise SpadaError(rc, lib().spada_last_error().decode("utf-8", "replace"))
