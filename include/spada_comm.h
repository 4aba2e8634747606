/*
 * spada_comm.h -- C ABI of libspada_comm.so: the one exchange step of the multi-GPU SpGEMM path, on RCCL over xGMI.
 *
 * The reference has no collective; its scheduler hands out disjoint A-row blocks (scheduler.rs:296-379).  Here every GPU
 * (one process per GPU) computes the C rows of its block with libspada_spgemm.so (B replicated) and the row blocks are
 * concatenated on every rank: an allgatherv, done as one group of per-rank ncclBroadcast calls per array, so every segment
 * crosses each xGMI link once and lands at its final offset -- no staging copy.  Counts travel first (ncclAllGather).
 *
 * Two forms:
 *   spada_comm_allgatherv_c     after a finished SpGEMM of the block (e.g. spada_dev_spgemm_fused): exchange only.
 *   spada_dist_spgemm_*         two-phase distributed SpGEMM with OVERLAP: the symbolic call also exchanges the sizes, so every
 *                               rank knows where every block goes; the numeric call computes the own block directly at its
 *                               final offset of the full C, in `chunks` pieces, and broadcasts every finished piece on a
 *                               communication stream while the next piece is computed.
 * Conventions as in spada_ffi.h (int status, spada_last_error(), no exceptions, caller-allocated outputs).
 * The communicator id (128 bytes) is created by rank 0 and handed to the other ranks by whatever the host uses to start them
 * (torch.distributed in bench.py; MPI, a file, a socket in a Rust host).
 */
#ifndef SPADA_COMM_H
#define SPADA_COMM_H

#include "spada_ffi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SPADA_COMM_ID_BYTES 128

typedef struct spada_comm spada_comm;

int spada_comm_get_unique_id(void *id_out /* SPADA_COMM_ID_BYTES */);
/* one communicator per process / GPU; device = HIP ordinal (the one of the rank's spada_ctx) */
int spada_comm_create(const void *id, int rank, int nranks, int device, spada_comm **out);
void spada_comm_destroy(spada_comm *comm);
int spada_comm_rank(const spada_comm *comm);
int spada_comm_size(const spada_comm *comm);

/* rows and nnz(C) of every rank's block (host arrays of nranks entries) */
int spada_comm_allgather_counts(spada_comm *comm, uint64_t my_rows, uint64_t my_nnz, uint64_t *rows_of_rank,
                                uint64_t *nnz_of_rank);
/* The offset arithmetic of the exchange, as a pure function of the gathered counts (host only: no GPU, no RCCL call; both
 * exchange forms below place every segment with it, and tests/test_host_cpu.py checks it for N = 2, 3, 8 with empty blocks and
 * empty pieces against the offsets of the torch.distributed path).  Inputs: rows / nnz of every rank's block (nranks entries
 * each) and, when chunks > 0, the piece positions of every rank inside its own block, chunk_pos[r * (chunks + 1) + k], k = 0 ..
 * chunks (first 0, last nnz[r], ascending -- what spada_dev_spgemm_numeric_plan returned on rank r).  Outputs: row_off / nnz_off
 * [nranks + 1] = first row / first entry of every block in the whole C (last = totals); piece_begin / piece_count [nranks * chunks]
 * = where piece k of rank r starts in the whole C and how many entries it holds (null when chunks == 0).
 * Replaces nothing in the reference (it has no exchange); the row blocks are those of scheduler.rs:296-379. */
int spada_comm_plan(int nranks, const uint64_t *rows_of_rank, const uint64_t *nnz_of_rank, uint32_t chunks,
                    const uint64_t *chunk_pos, uint64_t *row_off, uint64_t *nnz_off, uint64_t *piece_begin,
                    uint64_t *piece_count);

/* Concatenation of the finished blocks on every rank.  Inputs: this rank's block (device; indptr u64[my_rows + 1] starting
 * at 0, indices u32, data f64).  Outputs (device, caller-allocated from the counts): indptr u64[sum rows + 1],
 * indices u32[sum nnz], data f64[sum nnz] of the whole C.  Returns when the result is complete. */
int spada_comm_allgatherv_c(spada_comm *comm, const void *d_my_indptr, const void *d_my_indices, const void *d_my_data,
                            const uint64_t *rows_of_rank, const uint64_t *nnz_of_rank, void *d_c_indptr,
                            void *d_c_indices, void *d_c_data);

/* Distributed two-phase SpGEMM with overlapped exchange.  symbolic: spada_dev_spgemm_symbolic of rows [row_begin, row_end)
 * on this rank, then the exchange of all sizes; rows_of_rank / nnz_of_rank (nranks entries each) tell the caller how large
 * the whole C is.  numeric: fills the caller's buffers with the whole C on every rank. */
int spada_dist_spgemm_symbolic(spada_ctx *ctx, spada_comm *comm, const spada_dev_csr *a, const spada_dev_csr *b,
                               uint64_t row_begin, uint64_t row_end, uint32_t chunks, uint64_t *rows_of_rank,
                               uint64_t *nnz_of_rank);
int spada_dist_spgemm_numeric(spada_ctx *ctx, spada_comm *comm, void *d_c_indptr, void *d_c_indices, void *d_c_data);

#ifdef __cplusplus
}
#endif
#endif /* SPADA_COMM_H */
