/* spada_cycle.h -- C ABI of the cycle-level Spada model.
 *
 * A CPU restatement of what `spada-sim accuratesimu <accelerator> ...` simulates (reference: src/simulator.rs:509-890 the
 * cycle loop, src/scheduler.rs the block / window / merge-task planner, src/storage.rs:460-1007 the fiber cache,
 * src/adder_tree.rs the 8-way mergers, src/rowwise_perf_adjust.rs the block-height policy): it produces the product C in the
 * order the accelerator would and the five counters main.rs:97-108 prints (execution cycles, A / B / C memory words, cache
 * words).  It is host code and does not touch the GPU; the HIP path of spada_ffi.h computes the same C thousands of times
 * faster and is what the bench measures.
 *
 * PARITY: UNPINNED.  The reference cannot be built in this image (no Rust toolchain) and ships no expected counters, so the
 * numbers of this model are checked only against their own invariants and C against the CPU oracle.  The reference itself is
 * not reproducible run to run: it iterates std HashMap / HashSet (scheduler.rs:385-404, :827-856, storage.rs:604-609,
 * simulator.rs:986-994), whose order is randomised per process, when it picks fibers to merge and victims to evict.  This
 * model visits those containers in ascending key order -- one of the orders the reference can take.
 */
#ifndef SPADA_CYCLE_H
#define SPADA_CYCLE_H
#include <stdint.h>
#include "spada_ffi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* frontend.rs:26-33 (the spellings are accepted case-insensitively by the CLI) */
enum spada_accelerator { SPADA_ACCEL_IP = 0, SPADA_ACCEL_OP = 1, SPADA_ACCEL_MULTIROW = 2, SPADA_ACCEL_SPADA = 3 };

/* the fields of the JSON configuration (frontend.rs:9-23) that steer the model */
typedef struct spada_cycle_config {
    uint64_t struct_size;            /* sizeof(spada_cycle_config) */
    uint64_t pe_num, at_num, lane_num;
    uint64_t cache_size, word_byte;  /* bytes, bytes per word */
    uint64_t block_shape[2];
    uint64_t mem_latency, cache_latency;
    float freq;                      /* GHz */
    uint64_t channel;
    float bandwidth_per_channel;     /* GB/s */
    int32_t accelerator;             /* enum spada_accelerator */
    int32_t pad;
} spada_cycle_config;

/* what main.rs:97-108 prints, plus diagnostics */
typedef struct spada_cycle_counts {
    uint64_t struct_size;
    uint64_t exec_cycles;            /* get_exec_cycle(): cycles of the loop minus the smallest per-PE drain discount */
    uint64_t raw_cycles;             /* cycles of the loop */
    uint64_t a_read, a_write;        /* words */
    uint64_t b_read, b_write;
    uint64_t c_read, c_write;        /* the psum DRAM */
    uint64_t cache_read, cache_write;
    uint64_t cache_miss, b_evict, psum_evict;
    uint64_t blocks, windows, pe_merge_tasks, tree_merge_tasks;
    uint64_t c_nnz;
} spada_cycle_counts;

typedef struct spada_cycle_model spada_cycle_model;

/* A, B: borrowed until spada_cycle_destroy.  row_remap (may be NULL): the -p permutation, new row -> row of A
 * (CsrMatStorage::reorder_row, storage.rs:256-259); the result is reported under the original row numbers. */
int spada_cycle_create(const spada_cycle_config *cfg, const spada_csr_view *A, const spada_csr_view *B,
                       const uint64_t *row_remap, spada_cycle_model **out);
/* Simulator::execute (simulator.rs:509).  max_cycles = 0: no limit; otherwise SPADA_ERR_UNSUPPORTED when exceeded. */
int spada_cycle_execute(spada_cycle_model *m, uint64_t max_cycles);
int spada_cycle_get_counts(const spada_cycle_model *m, spada_cycle_counts *out);
/* get_exec_result (simulator.rs:1034-1062) as CSR: c_indptr[rows + 1], c_indices / c_data [c_nnz] */
int spada_cycle_get_result(const spada_cycle_model *m, uint64_t *c_indptr, uint64_t *c_indices, double *c_data);
void spada_cycle_destroy(spada_cycle_model *m);

#ifdef __cplusplus
}
#endif
#endif
