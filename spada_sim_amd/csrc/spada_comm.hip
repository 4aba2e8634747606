// libspada_comm.so: allgatherv of the C row blocks on RCCL over xGMI (include/spada_comm.h).  One process per GPU; the
// compute side is libspada_spgemm.so, used here only through its C ABI.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <vector>

#include "spada_comm.h"
#include "spada_internal.hpp"

using namespace spada;

#define HIP_TRY(expr)                                                                                                      \
    do {                                                                                                                   \
        hipError_t e_ = (expr);                                                                                            \
        if (e_ != hipSuccess)                                                                                              \
            return fail(e_ == hipErrorOutOfMemory ? SPADA_ERR_OOM : SPADA_ERR_HIP, "%s failed: %s (%s:%d)", #expr,          \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                                        \
    } while (0)
#define NCCL_TRY(expr)                                                                                                     \
    do {                                                                                                                   \
        ncclResult_t r_ = (expr);                                                                                          \
        if (r_ != ncclSuccess)                                                                                             \
            return fail(SPADA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__);         \
    } while (0)

static_assert(sizeof(ncclUniqueId) <= SPADA_COMM_ID_BYTES, "unique id fits the ABI's 128 bytes");

struct spada_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    hipStream_t stream = nullptr;     // communication stream
    uint64_t *d_small = nullptr;      // device staging of the count exchanges
    size_t small_words = 0;
    void *d_tmp_ptr = nullptr;        // this rank's C.indptr (distributed numeric call)
    size_t tmp_ptr_bytes = 0;
    // state of the distributed two-phase call
    uint32_t chunks = 0;
    std::vector<uint64_t> rows, nnz, pos;   // per rank: rows, nnz, (chunks + 1) chunk positions
};

namespace {

// segment r of the concatenated indptr holds block-local offsets: add the nnz of the blocks before it
#ifndef SPADA_COMM_MOCK
__global__ void k_shift_indptr(uint64_t *__restrict__ p, uint64_t n, uint64_t add)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] += add;
}
#endif

int ensure_small(spada_comm *m, size_t words)
{
    if (words <= m->small_words) return SPADA_OK;
    if (m->d_small) (void)hipFree(m->d_small);
    m->d_small = nullptr;
    m->small_words = 0;
    HIP_TRY(hipMalloc((void **)&m->d_small, words * 8));
    m->small_words = words;
    return SPADA_OK;
}

// every rank contributes `per` words; out (host) receives nranks * per words
int allgather_words(spada_comm *m, const uint64_t *mine, size_t per, uint64_t *out)
{
    int rc = ensure_small(m, per * (size_t)(m->nranks + 1));
    if (rc) return rc;
    uint64_t *d_send = m->d_small, *d_recv = m->d_small + per;
    HIP_TRY(hipMemcpyAsync(d_send, mine, per * 8, hipMemcpyHostToDevice, m->stream));
    NCCL_TRY(ncclAllGather(d_send, d_recv, per, ncclUint64, m->comm, m->stream));
    HIP_TRY(hipMemcpyAsync(out, d_recv, per * 8 * (size_t)m->nranks, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return SPADA_OK;
}

// indptr of the whole C from the blocks' local ones: broadcast entries 1 .. rows_r of every block behind one another, then
// shift every segment by the nnz before it.  d_local = this rank's local indptr (rows + 1 entries, first = 0); row_off / nnz_off =
// what spada_comm_plan computed from the gathered counts.
int gather_indptr(spada_comm *m, const uint64_t *d_local, const uint64_t *rows, const uint64_t *row_off, const uint64_t *nnz_off,
                  uint64_t *d_full)
{
    HIP_TRY(hipMemsetAsync(d_full, 0, 8, m->stream));
    if (rows[m->rank])
        HIP_TRY(hipMemcpyAsync(d_full + row_off[m->rank] + 1, d_local + 1, rows[m->rank] * 8, hipMemcpyDeviceToDevice, m->stream));
    NCCL_TRY(ncclGroupStart());
    for (int r = 0; r < m->nranks; ++r)
        if (rows[r])
            NCCL_TRY(ncclBroadcast(d_full + row_off[r] + 1, d_full + row_off[r] + 1, rows[r], ncclUint64, r, m->comm, m->stream));
    NCCL_TRY(ncclGroupEnd());
    for (int r = 0; r < m->nranks; ++r)
        if (rows[r] && nnz_off[r]) {
#ifdef SPADA_COMM_MOCK   /* the sanitizer / call-sequence build (mock/): "device" memory is host memory */
            for (uint64_t i = 0; i < rows[r]; ++i) d_full[row_off[r] + 1 + i] += nnz_off[r];
#else
            const uint32_t grid = (uint32_t)std::min<uint64_t>((rows[r] + 255) / 256, 2048);
            hipLaunchKernelGGL(k_shift_indptr, dim3(grid), dim3(256), 0, m->stream, d_full + row_off[r] + 1, rows[r], nnz_off[r]);
#endif
        }
    HIP_TRY(hipGetLastError());
    return SPADA_OK;
}

}  // namespace

extern "C" {

int spada_comm_get_unique_id(void *id_out)
{
    if (!id_out) return fail(SPADA_ERR_INVALID, "spada_comm_get_unique_id: null argument");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    std::memset(id_out, 0, SPADA_COMM_ID_BYTES);
    std::memcpy(id_out, &id, sizeof id);
    return SPADA_OK;
}

int spada_comm_create(const void *id, int rank, int nranks, int device, spada_comm **out)
{
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(SPADA_ERR_INVALID, "spada_comm_create: bad argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    spada_comm *m = new spada_comm;
    m->rank = rank;
    m->nranks = nranks;
    m->device = device;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclResult_t r = ncclCommInitRank(&m->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        delete m;
        return fail(SPADA_ERR_HIP, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)ncclCommDestroy(m->comm);
        delete m;
        return fail(SPADA_ERR_HIP, "hipStreamCreate failed for the communication stream");
    }
    *out = m;
    return SPADA_OK;
}

void spada_comm_destroy(spada_comm *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->comm) (void)ncclCommDestroy(m->comm);
    if (m->d_small) (void)hipFree(m->d_small);
    if (m->d_tmp_ptr) (void)hipFree(m->d_tmp_ptr);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

int spada_comm_rank(const spada_comm *m) { return m ? m->rank : -1; }
int spada_comm_size(const spada_comm *m) { return m ? m->nranks : 0; }

int spada_comm_allgather_counts(spada_comm *m, uint64_t my_rows, uint64_t my_nnz, uint64_t *rows_of_rank, uint64_t *nnz_of_rank)
{
    if (!m || !rows_of_rank || !nnz_of_rank) return fail(SPADA_ERR_INVALID, "spada_comm_allgather_counts: null argument");
    HIP_TRY(hipSetDevice(m->device));
    const uint64_t mine[2] = {my_rows, my_nnz};
    std::vector<uint64_t> all(2 * (size_t)m->nranks);
    int rc = allgather_words(m, mine, 2, all.data());
    if (rc) return rc;
    for (int r = 0; r < m->nranks; ++r) {
        rows_of_rank[r] = all[2 * r];
        nnz_of_rank[r] = all[2 * r + 1];
    }
    return SPADA_OK;
}

int spada_comm_plan(int nranks, const uint64_t *rows, const uint64_t *nnz, uint32_t chunks, const uint64_t *chunk_pos,
                    uint64_t *row_off, uint64_t *nnz_off, uint64_t *piece_begin, uint64_t *piece_count)
{
    if (nranks < 1 || !rows || !nnz || !row_off || !nnz_off) return fail(SPADA_ERR_INVALID, "spada_comm_plan: bad argument");
    if (chunks && (!chunk_pos || !piece_begin || !piece_count)) return fail(SPADA_ERR_INVALID, "spada_comm_plan: chunks without positions");
    uint64_t ro = 0, no = 0;
    for (int r = 0; r < nranks; ++r) {
        row_off[r] = ro;
        nnz_off[r] = no;
        if (ro + rows[r] < ro || no + nnz[r] < no) return fail(SPADA_ERR_INVALID, "spada_comm_plan: counts overflow 64 bits");
        if (chunks) {
            const uint64_t *p = chunk_pos + (size_t)r * (chunks + 1);
            if (p[0] != 0 || p[chunks] != nnz[r])
                return fail(SPADA_ERR_INVALID, "spada_comm_plan: pieces of rank %d do not cover its %llu entries", r, (unsigned long long)nnz[r]);
            for (uint32_t k = 0; k < chunks; ++k) {
                if (p[k + 1] < p[k]) return fail(SPADA_ERR_INVALID, "spada_comm_plan: piece positions of rank %d descend", r);
                piece_begin[(size_t)r * chunks + k] = no + p[k];
                piece_count[(size_t)r * chunks + k] = p[k + 1] - p[k];
            }
        }
        ro += rows[r];
        no += nnz[r];
    }
    row_off[nranks] = ro;
    nnz_off[nranks] = no;
    return SPADA_OK;
}

int spada_comm_allgatherv_c(spada_comm *m, const void *d_my_indptr, const void *d_my_indices, const void *d_my_data,
                            const uint64_t *rows, const uint64_t *nnz, void *d_c_indptr, void *d_c_indices, void *d_c_data)
{
    if (!m || !d_my_indptr || !rows || !nnz || !d_c_indptr) return fail(SPADA_ERR_INVALID, "spada_comm_allgatherv_c: null argument");
    HIP_TRY(hipSetDevice(m->device));
    std::vector<uint64_t> row_off(m->nranks + 1), nnz_off(m->nranks + 1);
    int rc = spada_comm_plan(m->nranks, rows, nnz, 0, nullptr, row_off.data(), nnz_off.data(), nullptr, nullptr);
    if (rc) return rc;
    const uint64_t total = nnz_off[m->nranks], off = nnz_off[m->rank];
    if (total && (!d_c_indices || !d_c_data || (nnz[m->rank] && (!d_my_indices || !d_my_data))))
        return fail(SPADA_ERR_INVALID, "spada_comm_allgatherv_c: null data pointer");
    uint32_t *ci = (uint32_t *)d_c_indices;
    double *cv = (double *)d_c_data;
    if (nnz[m->rank]) {   // own block to its final place; the broadcasts below are in place
        HIP_TRY(hipMemcpyAsync(ci + off, d_my_indices, nnz[m->rank] * 4, hipMemcpyDeviceToDevice, m->stream));
        HIP_TRY(hipMemcpyAsync(cv + off, d_my_data, nnz[m->rank] * 8, hipMemcpyDeviceToDevice, m->stream));
    }
    NCCL_TRY(ncclGroupStart());
    for (int r = 0; r < m->nranks; ++r)
        if (nnz[r]) {
            NCCL_TRY(ncclBroadcast(ci + nnz_off[r], ci + nnz_off[r], nnz[r], ncclUint32, r, m->comm, m->stream));
            NCCL_TRY(ncclBroadcast(cv + nnz_off[r], cv + nnz_off[r], nnz[r], ncclFloat64, r, m->comm, m->stream));
        }
    NCCL_TRY(ncclGroupEnd());
    if ((rc = gather_indptr(m, (const uint64_t *)d_my_indptr, rows, row_off.data(), nnz_off.data(), (uint64_t *)d_c_indptr))) return rc;
    HIP_TRY(hipStreamSynchronize(m->stream));
    return SPADA_OK;
}

int spada_dist_spgemm_symbolic(spada_ctx *ctx, spada_comm *m, const spada_dev_csr *a, const spada_dev_csr *b, uint64_t row_begin,
                               uint64_t row_end, uint32_t chunks, uint64_t *rows_of_rank, uint64_t *nnz_of_rank)
{
    if (!ctx || !m || !rows_of_rank || !nnz_of_rank) return fail(SPADA_ERR_INVALID, "spada_dist_spgemm_symbolic: null argument");
    if (!chunks || chunks > 256) return fail(SPADA_ERR_INVALID, "spada_dist_spgemm_symbolic: 1 .. 256 chunks");
    uint64_t my_nnz = 0;
    int rc = spada_dev_spgemm_symbolic(ctx, a, b, row_begin, row_end, &my_nnz);
    if (rc) return rc;
    // sizes of every block and of every piece of it: rows | nnz | chunk positions [chunks + 1]
    const size_t per = (size_t)chunks + 3;
    std::vector<uint64_t> mine(per), all(per * (size_t)m->nranks);
    mine[0] = row_end - row_begin;
    mine[1] = my_nnz;
    if ((rc = spada_dev_spgemm_numeric_plan(ctx, chunks, mine.data() + 2))) return rc;
    HIP_TRY(hipSetDevice(m->device));
    if ((rc = allgather_words(m, mine.data(), per, all.data()))) return rc;
    m->chunks = chunks;
    m->rows.assign(m->nranks, 0);
    m->nnz.assign(m->nranks, 0);
    m->pos.assign((size_t)m->nranks * (chunks + 1), 0);
    for (int r = 0; r < m->nranks; ++r) {
        rows_of_rank[r] = m->rows[r] = all[per * r];
        nnz_of_rank[r] = m->nnz[r] = all[per * r + 1];
        for (uint32_t k = 0; k <= chunks; ++k) m->pos[(size_t)r * (chunks + 1) + k] = all[per * r + 2 + k];
    }
    return SPADA_OK;
}

int spada_dist_spgemm_numeric(spada_ctx *ctx, spada_comm *m, void *d_c_indptr, void *d_c_indices, void *d_c_data)
{
    if (!ctx || !m || !d_c_indptr) return fail(SPADA_ERR_INVALID, "spada_dist_spgemm_numeric: null argument");
    if (!m->chunks || (int)m->rows.size() != m->nranks)
        return fail(SPADA_ERR_STATE, "spada_dist_spgemm_numeric called without spada_dist_spgemm_symbolic");
    HIP_TRY(hipSetDevice(m->device));
    const uint32_t K = m->chunks;
    std::vector<uint64_t> row_off(m->nranks + 1), nnz_off(m->nranks + 1), pbeg((size_t)m->nranks * K), pcnt((size_t)m->nranks * K);
    int rc = spada_comm_plan(m->nranks, m->rows.data(), m->nnz.data(), K, m->pos.data(), row_off.data(), nnz_off.data(), pbeg.data(),
                             pcnt.data());
    if (rc) return rc;
    const uint64_t total = nnz_off[m->nranks], off_me = nnz_off[m->rank];
    if (total && (!d_c_indices || !d_c_data)) return fail(SPADA_ERR_INVALID, "spada_dist_spgemm_numeric: null data pointer");
    uint32_t *ci = (uint32_t *)d_c_indices;
    double *cv = (double *)d_c_data;
    // every piece of the own block is computed at its final place in the whole C and broadcast from there as soon as it is
    // complete, on the communication stream, while the engine stream computes the next piece
    for (uint32_t k = 0; k < K; ++k) {
        void *ev = nullptr;
        if ((rc = spada_dev_spgemm_numeric_chunk(ctx, k, ci + off_me, cv + off_me, &ev))) return rc;
        HIP_TRY(hipStreamWaitEvent(m->stream, (hipEvent_t)ev, 0));
        NCCL_TRY(ncclGroupStart());
        for (int r = 0; r < m->nranks; ++r) {
            const uint64_t p0 = pbeg[(size_t)r * K + k], n = pcnt[(size_t)r * K + k];
            if (n) {
                NCCL_TRY(ncclBroadcast(ci + p0, ci + p0, n, ncclUint32, r, m->comm, m->stream));
                NCCL_TRY(ncclBroadcast(cv + p0, cv + p0, n, ncclFloat64, r, m->comm, m->stream));
            }
        }
        NCCL_TRY(ncclGroupEnd());
    }
    // C.indptr: small, after the data
    const size_t need = ((size_t)m->rows[m->rank] + 1) * 8;
    if (need > m->tmp_ptr_bytes) {
        if (m->d_tmp_ptr) (void)hipFree(m->d_tmp_ptr);
        m->d_tmp_ptr = nullptr;
        m->tmp_ptr_bytes = 0;
        HIP_TRY(hipMalloc(&m->d_tmp_ptr, need + need / 4));
        m->tmp_ptr_bytes = need + need / 4;
    }
    if ((rc = spada_dev_spgemm_indptr(ctx, m->d_tmp_ptr))) return rc;
    if ((rc = spada_dev_synchronize(ctx))) return rc;   // the copy above ran on the engine stream
    if ((rc = gather_indptr(m, (const uint64_t *)m->d_tmp_ptr, m->rows.data(), row_off.data(), nnz_off.data(), (uint64_t *)d_c_indptr)))
        return rc;
    HIP_TRY(hipStreamSynchronize(m->stream));
    return SPADA_OK;
}

}  // extern "C"
