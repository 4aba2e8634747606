"""spada_sim_amd -- MI355X (gfx950) SpGEMM engine behind spada-sim's host interface.

The product is spada_sim_amd/lib/libspada_spgemm.so (hand-written HIP kernels + C ABI, see
include/spada_ffi.h).  This package is the thin host-side mirror of the reference's interface.
"""
from ._ffi import SpadaError, LIB_PATH  # noqa: F401
from .host import (ACC_LDS_HASH, ACC_SORT_MERGE, REORDER_BY_LENGTH, REORDER_BY_PRODUCTS, sort_by_length, sort_by_products, GEN_CAGE12_LIKE, GEN_COP20K_LIKE, GEN_MC2DEPI_LIKE,  # noqa: F401
                   GEN_RMAT, GEN_UNIFORM, GEN_WEBBASE_LIKE, Comm, CsMat, CsrMatStorage, CsrRow, CycleModel, Engine, GEMM,
                   Simulator, count_products, device_count, generate, load_mm_mat, parse_config,
                   partition_rows, write_mm_mat, write_bin, read_bin, checksum, comm_plan)
