/*
 * spada_ffi.h -- C ABI of libspada_spgemm.so, the MI355X (gfx950) SpGEMM engine that replaces
 * spada-sim's simulated multiply/merge dataflow.
 *
 * The reference (tsinghua-ideal/spada-sim) is a binary crate with no FFI of its own.  The seam this
 * ABI replaces is the one src/main.rs drives (all citations into /root/reference/src):
 *
 *   CsrMatStorage::init_with_gemm(GEMM) -> (A, B)        storage.rs:214-239   inputs: three Vecs per matrix
 *   Simulator::new(cfg.., &mut A, &mut B, &mut psum, ..) simulator.rs:431-507 borrows A/B for its lifetime
 *   Simulator::execute(&mut self)                        simulator.rs:509-890 blocking, single thread
 *   Simulator::get_exec_result() -> Vec<CsrRow>          simulator.rs:1034-1062
 *   get_exec_cycle / get_{a,b,c}_mat_stat / get_cache_stat   simulator.rs:1008-1032
 *   GEMM::from_mat (square ? A*A : A*A^T)                gemm.rs:41-53
 *   load_mm_mat (scipy mmread(...).tocsr())              py2rust.rs:62-97
 *   parse_config (13 required JSON keys)                 frontend.rs:8-23, :77-85
 *
 * Conventions
 *   - plain C, no C++ or torch types; `usize` == uint64_t; values are f64.
 *   - every function returns SPADA_OK (0) or a positive error code; the message is available from
 *     spada_last_error() (thread local).  Nothing throws or aborts across the boundary.
 *   - inputs are BORROWED (the caller keeps its Vec<f64>/Vec<usize> alive between the symbolic and the
 *     numeric call, as Simulator<'a> borrows A and B); outputs are CALLER-ALLOCATED between the two
 *     phases, so no memory crosses allocators.
 *   - a context is bound to one GPU and is not thread safe; calls block until the result is complete.
 *   - there is NO CPU fallback: without a usable gfx950 device spada_create() fails with
 *     SPADA_ERR_NO_DEVICE and every compute entry point fails with SPADA_ERR_STATE.
 *   - column indices are narrowed to 32 bit on the device (cols < 2^32 is checked); the host ABI
 *     stays 64 bit to match `usize`.
 *   - STREAM ORDER (spada_dev_* entry points): the engine queues its kernels on a stream of its own (non-blocking, i.e. not
 *     ordered against the caller's default or torch stream).  Every spada_dev_* compute call returns only after its RESULT is
 *     complete on that stream (what may still run behind it touches the engine's own workspaces only: the counters and per-row
 *     accumulators of the symbolic / one-pass pipeline are put back for the next run there) -- EXCEPT spada_dev_spgemm_numeric_chunk
 *     and spada_dev_spgemm_indptr, which return as soon as their work is queued (wait for the event the former returns, or call
 *     spada_dev_synchronize).  So results may be read on any stream once a
 *     call has returned; but work the CALLER has queued on its own streams is not waited for: a caller that reads, fills or
 *     frees a buffer on its own stream must have that stream finished with the buffer before it hands it (or memory reused
 *     from it) to the next spada_dev_* call (hipStreamSynchronize / an event wait of its own).
 *   - SPADA_TRACE=0/1/2 in the environment (replaces the trace_exec cargo feature, util.rs:1-24, Cargo.toml:19-21): 1 = one line per
 *     call on stderr (what was loaded, products / nnz(C) / tasks / device time of a pipeline run), 2 = also row classes, phase
 *     times and workspace growth.  0 / unset: silent.
 */
#ifndef SPADA_FFI_H
#define SPADA_FFI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPADA_ABI_VERSION 5

enum spada_status {
    SPADA_OK = 0,
    SPADA_ERR_INVALID = 1,      /* bad argument / malformed CSR */
    SPADA_ERR_NO_DEVICE = 2,    /* no HIP device, or not gfx950 */
    SPADA_ERR_HIP = 3,          /* a HIP runtime call failed */
    SPADA_ERR_OOM = 4,
    SPADA_ERR_IO = 5,           /* file could not be opened / read */
    SPADA_ERR_PARSE = 6,        /* MatrixMarket / JSON syntax, missing config key */
    SPADA_ERR_STATE = 7,        /* call sequence violated (numeric before symbolic, ...) */
    SPADA_ERR_UNSUPPORTED = 8,  /* e.g. cols >= 2^32, array-format .mtx, NN workload */
    SPADA_ERR_CAPACITY = 9      /* one-pass SpGEMM: the caller's C buffers hold fewer than nnz(C) entries */
};

/* CsrMatStorage.{indptr, indices, data} (storage.rs:150-160) viewed without copying. */
typedef struct spada_csr_view {
    uint64_t rows, cols, nnz;
    const uint64_t *indptr;   /* rows + 1 */
    const uint64_t *indices;  /* nnz, ascending and unique inside a row */
    const double *data;       /* nnz */
} spada_csr_view;

/* Per-row accumulator used by the numeric phase (BASELINE.json configs[2] compares the two). */
enum spada_accumulator {
    SPADA_ACC_LDS_HASH = 0,   /* LDS hash accumulator + ordered emission (default) */
    SPADA_ACC_SORT_MERGE = 1  /* products to LDS, bitonic sort by (column, k), runs added left to right: bit-identical to a
                                 sequential sort-merge; same tasks as the hash variant */
};

typedef struct spada_options {
    uint32_t struct_size;     /* = sizeof(spada_options) */
    int32_t device;           /* HIP device ordinal; -1 = current device */
    int32_t accumulator;      /* enum spada_accumulator */
    int32_t flags;            /* reserved, 0 */
} spada_options;

/* Replaces get_exec_cycle / get_*_mat_stat (simulator.rs:1008-1032) with measured quantities of the last call(s). */
typedef struct spada_stats {
    uint64_t rows;            /* C rows computed by the last call (row range length) */
    uint64_t a_nnz;           /* A nonzeros inside the row range */
    uint64_t b_nnz;
    uint64_t nprod;           /* sum over those A nonzeros of nnz(B row) */
    uint64_t c_nnz;
    uint64_t bytes_read;      /* algorithmic: (m+1)*8 + a_nnz*12 + a_nnz*16 + nprod*12    */
    uint64_t bytes_write;     /* algorithmic: (m+1)*8 + c_nnz*12 */
    /* HIP-event times on the engine stream, milliseconds */
    double ms_symbolic_call;  /* whole spada_dev_spgemm_symbolic call (row statistics ... counting task kernel) */
    double ms_numeric_call;   /* whole spada_dev_spgemm_numeric call */
    double ms_fused_call;     /* whole spada_dev_spgemm_fused call (with ms_symbolic_call > 0: its count + numeric pipeline ran, ms_task = both task kernels) */
    double ms_row_stats;      /* B-row descriptors, products per row, row classes */
    double ms_big_expand;     /* BIG rows, the kernels on the engine stream: parts, column histograms, ranges (and the scatter of the spilled rows /
                                 the cut table of the direct rows when they are not forked to the side streams) */
    double ms_cut;            /* from the end of the plan to the start of the task kernel: the task list (scan kernels) and, next to them on the
                                 side streams, scatter and cut table.  The three phases are consecutive intervals of the engine stream */
    double ms_task;           /* the task kernel of the last call (count, numeric or one-pass) */
    uint64_t cls_rows[8];     /* rows per class: 0 EMPTY, 1 COPY (one A entry), 2 SMALL, 3 SOLO, 4 BIG (column-range tasks) */
    uint64_t cls_prod[8];     /* products per class */
    uint64_t n_tasks;         /* tasks of the last pipeline run */
    uint64_t multi_pass_tasks;/* range tasks that had to halve their column range (more distinct columns than the table takes) */
    uint64_t scratch_products;/* products spilled to HBM scratch (BIG rows whose range tasks do not read B directly) */
    uint64_t spill_rows;      /* BIG rows that took the HBM spill path */
    uint64_t pipeline_runs;   /* > 1 when a workspace had to grow and the pipeline was run again (first call of a context) */
    uint64_t workspace_bytes; /* device scratch owned by the context */
    uint64_t task_product_limit; /* products one task hashes at most: 2040 (lds_hash) / 1536 (sort_merge) */
    /* ABI 5 */
    uint64_t chain_fallbacks; /* pipeline runs of the last call that gave themselves up on the one-pass chain (a wait that exceeded
                                 SPADA_CHAIN_TIMEOUT_MS: the call then fails with SPADA_ERR_HIP instead of holding the GPU) */
    uint64_t pipeline_kind;   /* what spada_dev_spgemm_fused ran: 0 one pass, 1 count + positions + numeric into the caller's buffers (most
                                 products in BIG rows, or tasks that need the older range path) */
    double ms_wall_call;      /* host wall-clock time of the last spada_dev_spgemm_symbolic / _numeric / _fused call, everything included
                                 (workspace growth, repeated pipeline runs, the wait for the result) */
} spada_stats;

typedef struct spada_ctx spada_ctx;          /* engine context: one GPU, one stream, scratch */
typedef struct spada_dev_csr spada_dev_csr;  /* CSR resident in HBM (u64 indptr, u32 indices, f64 data) */
typedef struct spada_host_csr spada_host_csr;/* CSR owned by the library on the host */

/* ---- diagnostics ------------------------------------------------------------------------- */
const char *spada_last_error(void);
int spada_abi_version(void);
/* number of usable gfx950 devices; 0 when none (never fails) */
int spada_device_count(void);

/* ---- engine life cycle (replaces Simulator::new / drop, simulator.rs:431-507) ---------------- */
int spada_create(const spada_options *opts, spada_ctx **out);
void spada_destroy(spada_ctx *ctx);

/* ---- host-pointer two-phase SpGEMM: the drop-in seam ---------------------------------------
 * symbolic: uploads A and B (B may alias A), computes nnz(C) -> *nnz_c.      [execute(), 1st half]
 * numeric : fills c_indptr (A.rows + 1), c_indices (nnz_c), c_data (nnz_c)   [execute() + get_exec_result()]
 * C rows are ascending, columns ascending and unique, explicit zeros kept (simulator.rs:1034-1062). */
int spada_spgemm_symbolic(spada_ctx *ctx, const spada_csr_view *a, const spada_csr_view *b, uint64_t *nnz_c);
int spada_spgemm_numeric(spada_ctx *ctx, uint64_t *c_indptr, uint64_t *c_indices, double *c_data);
/* symbolic phase with -p / --preprocess: A's rows are reordered on the device first (enum spada_reorder below), rowmap (A.rows
 * entries, may be NULL) receives row_remap (storage.rs:157); the numeric call that follows maps the rows of the product back,
 * so C comes out in the original row order, identical to the plain call (main.rs:60-63, simulator.rs:1039-1055). */
int spada_spgemm_symbolic_reordered(spada_ctx *ctx, const spada_csr_view *a, const spada_csr_view *b, int key, uint64_t *nnz_c,
                                    uint64_t *rowmap);
/* one pass, host pointers: c_indices / c_data hold `capacity` entries (e.g. Vec::with_capacity(spada_count_products)),
 * *nnz_c receives nnz(C).  SPADA_ERR_CAPACITY: c_indptr and *nnz_c are valid; resize to *nnz_c and call
 * spada_spgemm_numeric. */
int spada_spgemm_fused(spada_ctx *ctx, const spada_csr_view *a, const spada_csr_view *b, uint64_t capacity,
                       uint64_t *c_indptr, uint64_t *c_indices, double *c_data, uint64_t *nnz_c);

/* ---- device-resident SpGEMM (inputs already in HBM; what bench.py times) ---------------------
 * Row range [row_begin, row_end) of A selects the A-row block of this GPU (scheduler.rs:296-379
 * issues disjoint row blocks; B is replicated).  Outputs are device pointers supplied by the caller:
 * d_c_indptr  : uint64_t[row_end - row_begin + 1], offsets local to the block (starts at 0)
 * d_c_indices : uint32_t[nnz_c]     d_c_data : double[nnz_c]                                   */
int spada_dev_csr_upload(spada_ctx *ctx, const spada_csr_view *m, spada_dev_csr **out);
void spada_dev_csr_free(spada_ctx *ctx, spada_dev_csr *m);
/* -p / --preprocess as a device pre-pass.  The reference sorts the rows of A by length before the simulator visits them
 * (sort_by_length, preprocessing.rs:76-89; main.rs:60-63; stable, ascending) and maps the result rows back afterwards
 * (simulator.rs:1039-1055), so C is unchanged.  spada_dev_csr_reorder returns the reordered A in HBM (to be freed with
 * spada_dev_csr_free); key BY_PRODUCTS sorts by the number of products of a row instead (needs B), which is what balances
 * the GPU work.  spada_dev_csr_rowmap: row i of the reordered matrix is original row rowmap[i] (row_remap, storage.rs:157).
 * spada_dev_unpermute_c: the product of the reordered A (device CSR) -> C in the original row order (device, caller-allocated,
 * same sizes). */
enum spada_reorder { SPADA_REORDER_BY_LENGTH = 0, SPADA_REORDER_BY_PRODUCTS = 1 };
int spada_dev_csr_reorder(spada_ctx *ctx, const spada_dev_csr *a, const spada_dev_csr *b, int key, spada_dev_csr **a_reordered);
int spada_dev_csr_rowmap(spada_ctx *ctx, const spada_dev_csr *a_reordered, uint64_t *rowmap /* rows */);
int spada_dev_unpermute_c(spada_ctx *ctx, const spada_dev_csr *a_reordered, const void *d_p_indptr, const void *d_p_indices,
                          const void *d_p_data, void *d_c_indptr, void *d_c_indices, void *d_c_data);
int spada_dev_spgemm_symbolic(spada_ctx *ctx, const spada_dev_csr *a, const spada_dev_csr *b,
                              uint64_t row_begin, uint64_t row_end, uint64_t *nnz_c);
int spada_dev_spgemm_numeric(spada_ctx *ctx, void *d_c_indptr, void *d_c_indices, void *d_c_data);
/* Numeric phase in pieces, for callers that overlap the exchange of finished parts of C with the computation of the rest
 * (libspada_comm.so does: spada_comm.h).  After a symbolic call: _plan cuts the task list into `chunks` pieces and returns the
 * C entries each completes, chunk_pos[k] .. chunk_pos[k + 1] (chunks + 1 values, ascending, last = nnz(C)); _chunk queues piece k
 * on the engine stream WITHOUT waiting and hands back an event (hipEvent_t) that fires when its entries are in place;
 * _indptr queues the copy of C.indptr; _synchronize waits for everything queued. */
int spada_dev_spgemm_numeric_plan(spada_ctx *ctx, uint32_t chunks, uint64_t *chunk_pos);
int spada_dev_spgemm_numeric_chunk(spada_ctx *ctx, uint32_t k, void *d_c_indices, void *d_c_data, void **done_event);
int spada_dev_spgemm_indptr(spada_ctx *ctx, void *d_c_indptr);
int spada_dev_synchronize(spada_ctx *ctx);
/* One-pass SpGEMM (additive to the two-phase contract): no symbolic phase.  The caller supplies C buffers of `capacity`
 * entries -- any upper bound of nnz(C), e.g. spada_count_products (one product per entry at most) -- and receives
 * C.indptr, the first *nnz_c entries of indices / data, and *nnz_c.  SPADA_ERR_CAPACITY when capacity < nnz(C): indptr and
 * *nnz_c are complete, indices / data are not; allocate *nnz_c entries and call spada_dev_spgemm_numeric (the context then
 * holds the state of a finished symbolic phase).  Replaces Simulator::execute + get_exec_result in one call
 * (simulator.rs:509-890, :1034-1062).
 * Inside, the engine runs either its one-pass pipeline (row counts exchanged between the tasks while the rows are computed) or
 * count + positions + numeric into the same buffers: on an input whose products lie mostly in rows too large for one task it
 * measures both -- first call one pass, second call two phases -- and stays with the faster (spada_stats: ms_symbolic_call > 0
 * next to ms_fused_call says the two-phase pipeline ran; SPADA_AUTO=0 in the environment: always one pass).  C is the same. */
int spada_dev_spgemm_fused(spada_ctx *ctx, const spada_dev_csr *a, const spada_dev_csr *b, uint64_t row_begin,
                           uint64_t row_end, void *d_c_indptr, void *d_c_indices, void *d_c_data, uint64_t capacity,
                           uint64_t *nnz_c);
/* the same into device buffers owned by the context (capacity entries), valid until the next call that produces C */
int spada_dev_spgemm_fused_owned(spada_ctx *ctx, const spada_dev_csr *a, const spada_dev_csr *b, uint64_t row_begin,
                                 uint64_t row_end, uint64_t capacity, void **d_c_indptr, void **d_c_indices,
                                 void **d_c_data, uint64_t *nnz_c);
/* Convenience for callers without their own device allocator: device buffers owned by the context,
 * valid until the next symbolic call.  d_* receive device pointers. */
int spada_dev_spgemm_numeric_owned(spada_ctx *ctx, void **d_c_indptr, void **d_c_indices, void **d_c_data);
/* copy a finished block back: widens indices to u64 (usize) on the device, then D2H */
int spada_dev_download_c(spada_ctx *ctx, const void *d_c_indptr, const void *d_c_indices, const void *d_c_data,
                         uint64_t rows, uint64_t nnz_c, uint64_t *c_indptr, uint64_t *c_indices, double *c_data);

int spada_get_stats(const spada_ctx *ctx, spada_stats *out);
/* Phase times inside a call (ms_row_stats, ms_big_expand, ms_cut) need an event record between the kernels, and each record idles
 * the stream for about 5 us.  enabled = 0: those three fields stay 0 and the records are left out (ms_task and the call times
 * are still measured); default 1. */
int spada_set_phase_timing(spada_ctx *ctx, int enabled);

/* ---- host-side ingest (CPU only, usable without a GPU) -------------------------------------- */
/* load_mm_mat (py2rust.rs:62-97): <dir>/<name>.mtx -> canonical CSR as scipy mmread(...).tocsr() */
int spada_mtx_read(const char *path, spada_host_csr **out);
/* C writer (the reference never persists C): coordinate real general, 1-based */
int spada_mtx_write(const char *path, const spada_csr_view *m);   /* with a `% spada-sim checksum: ...` comment line */
/* checksum of a CSR: FNV-1a 64 over indptr then indices (structure) and over the value bytes, plus the sequential sum and
 * absolute sum of the values -- for cross-run parity and for comparing the C of 1 and 8 GPUs without shipping it */
typedef struct spada_checksum {
    uint64_t rows, cols, nnz;
    uint64_t structure_hash, value_hash;
    double value_sum, value_abs_sum;
} spada_checksum;
int spada_csr_checksum(const spada_csr_view *m, spada_checksum *out);
int spada_checksum_format(const spada_checksum *cs, char *buf, size_t n);   /* the one-line text form */
/* binary CSR dump (the three Vecs of CsrMatStorage as they are in memory, + header and checksums) and its reader */
int spada_csr_write_bin(const char *path, const spada_csr_view *m);
int spada_csr_read_bin(const char *path, spada_host_csr **out);
int spada_host_csr_from_view(const spada_csr_view *m, spada_host_csr **out);   /* deep copy */
int spada_host_csr_view(const spada_host_csr *m, spada_csr_view *out);
void spada_host_csr_free(spada_host_csr *m);
/* sorted CSR of A^T (gemm.rs:46) */
int spada_transpose(const spada_csr_view *a, spada_host_csr **out);
/* GEMM::from_mat (gemm.rs:41-53): square -> *b_out = NULL and *b_is_a = 1; else *b_out = A^T */
int spada_from_mat(const spada_csr_view *a, spada_host_csr **b_out, int *b_is_a);
/* structural validation used by every entry point: monotone indptr, indices < cols, ascending unique */
int spada_csr_validate(const spada_csr_view *m);

/* A-row block boundaries balanced on the prefix sum of per-row products.
 * bounds has nparts + 1 entries, bounds[0] = 0, bounds[nparts] = A.rows. */
int spada_partition_rows(const spada_csr_view *a, const spada_csr_view *b, uint32_t nparts, uint64_t *bounds);
/* sum over A nonzeros of nnz(B row) for rows [row_begin, row_end) */
int spada_count_products(const spada_csr_view *a, const spada_csr_view *b, uint64_t row_begin, uint64_t row_end,
                         uint64_t *nprod);

/* ---- synthetic workloads (deterministic, SplitMix64 counter based) ---------------- */
enum spada_gen_kind {
    SPADA_GEN_RMAT = 0,          /* p0 = scale, p1 = edge factor; (a,b,c,d) = (.57,.19,.19,.05) */
    SPADA_GEN_WEBBASE_LIKE = 1,  /* p0 = rows (0 -> 1000005), p1 = nnz target (0 -> 3105536) */
    SPADA_GEN_COP20K_LIKE = 2,   /* p0 = rows (0 -> 121192) */
    SPADA_GEN_CAGE12_LIKE = 3,   /* p0 = rows (0 -> 130228) */
    SPADA_GEN_MC2DEPI_LIKE = 4,  /* p0 = rows (0 -> 525825) */
    SPADA_GEN_UNIFORM = 5        /* p0 = rows, p1 = nnz per row (Erdos-Renyi rows) */
};
int spada_generate(int kind, uint64_t p0, uint64_t p1, uint64_t seed, spada_host_csr **out);

/* ---- configuration (frontend.rs:8-23, :77-85) ------------------------------------------------ */
typedef struct spada_config {
    char ss_filepath[1024];
    char nn_filepath[1024];
    uint64_t pe_num, at_num, lane_num, cache_size, word_byte;
    uint64_t block_shape[2];
    uint64_t mem_latency, cache_latency;
    float freq;
    uint64_t channel;
    float bandwidth_per_channel;
    /* optional engine keys (absent -> defaults): "gpus", "accumulator", "repeat" */
    uint32_t gpus;
    int32_t accumulator;
    uint32_t repeat;
} spada_config;
int spada_config_parse(const char *path, spada_config *out);

#ifdef __cplusplus
}
#endif
#endif /* SPADA_FFI_H */
