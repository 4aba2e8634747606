// TEST DOUBLE behind lib_asan/libspada_comm_mock.so (see mock/hip/hip_runtime.h): the logged RCCL calls, the scripted engine entry
// points (include/spada_ffi.h: what spada_dist_spgemm_symbolic / _numeric call on the compute library) and the C interface the test
// drives them with.  Linked with -Bsymbolic so that spada_comm.hip binds to THESE spada_dev_* functions, not to the refusing ones of
// the sanitizer build of libspada_spgemm.so.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rccl/rccl.h"
#include "spada_ffi.h"

namespace {
struct Call {
    uint64_t op, root, count, dtype, offset;   // op: 1 group start, 2 group end, 3 broadcast, 4 allgather; offset: bytes from the registered base
};
std::vector<Call> g_log;
// The ranks of a test run one after the other in one process, so a broadcast cannot move data when it is posted.  The test runs every
// rank TWICE: in the first pass a root's broadcast deposits what it sends (keyed by the number of the call in the rank's sequence --
// all ranks post the same sequence, which is what the test asserts first), in the second pass the other ranks' broadcasts pick it up.
std::vector<std::vector<char>> g_store;
int g_pass = 1, g_rank = 0;
uint64_t g_bcast_no = 0;
std::vector<uint64_t> g_gathered;            // what ncclAllGather delivers: the words of all ranks, rank-major
const char *g_base_idx = nullptr, *g_base_val = nullptr, *g_base_ptr = nullptr;   // the rank's C buffers (offsets in the log are relative to them)
uint64_t g_my_nnz = 0;
std::vector<uint64_t> g_pos;                 // piece positions of the rank's block (chunks + 1)
uint32_t g_fill = 0;                         // the scripted pieces are filled with g_fill + index
std::vector<uint64_t> g_indptr;              // the rank's local C.indptr

uint64_t offset_of(const void *p, uint64_t *which)
{
    const char *c = (const char *)p;
    // (the buffers of one rank do not overlap; the largest base below the pointer is the one it points into)
    const char *best = nullptr;
    uint64_t w = 0;
    if (g_base_idx && c >= g_base_idx && (!best || g_base_idx > best)) { best = g_base_idx; w = 1; }
    if (g_base_val && c >= g_base_val && (!best || g_base_val > best)) { best = g_base_val; w = 2; }
    if (g_base_ptr && c >= g_base_ptr && (!best || g_base_ptr > best)) { best = g_base_ptr; w = 3; }
    *which = w;
    return best ? (uint64_t)(c - best) : ~0ull;
}
}  // namespace

extern "C" {

const char *ncclGetErrorString(ncclResult_t) { return "mock rccl error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { std::memset(id, 7, sizeof *id); return ncclSuccess; }
ncclResult_t ncclCommInitRank(ncclComm_t *c, int, ncclUniqueId, int rank)
{
    *c = (ncclComm_t)std::malloc(1);
    g_rank = rank;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { std::free(c); return ncclSuccess; }
ncclResult_t ncclGroupStart() { g_log.push_back({1, 0, 0, 0, 0}); return ncclSuccess; }
ncclResult_t ncclGroupEnd() { g_log.push_back({2, 0, 0, 0, 0}); return ncclSuccess; }
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t, hipStream_t)
{
    uint64_t which = 0;
    const uint64_t off = offset_of(recv, &which);
    g_log.push_back({3, (uint64_t)root, (uint64_t)count, (uint64_t)t | (which << 8) | ((uint64_t)(send == recv ? 1 : 0) << 16), off});
    const size_t bytes = count * (t == ncclUint32 ? 4u : 8u);
    const uint64_t no = g_bcast_no++;
    if (g_pass == 1 && root == g_rank) {
        if (g_store.size() <= no) g_store.resize(no + 1);
        g_store[no].assign((const char *)send, (const char *)send + bytes);
    } else if (g_pass == 2 && root != g_rank) {
        if (no >= g_store.size() || g_store[no].size() != bytes) return 1;   // (no root has posted this call: the sequences differ)
        std::memcpy(recv, g_store[no].data(), bytes);
    }
    return ncclSuccess;
}
ncclResult_t ncclAllGather(const void *, void *recv, size_t count, ncclDataType_t t, ncclComm_t, hipStream_t)
{
    g_log.push_back({4, 0, (uint64_t)count, (uint64_t)t, 0});
    if (g_gathered.size() % count != 0) return 1;
    std::memcpy(recv, g_gathered.data(), g_gathered.size() * 8);
    return ncclSuccess;
}

// ---- scripted compute library ---------------------------------------------------------------------------------------------
int spada_dev_spgemm_symbolic(spada_ctx *, const spada_dev_csr *, const spada_dev_csr *, uint64_t, uint64_t, uint64_t *nnz_c)
{
    *nnz_c = g_my_nnz;
    return SPADA_OK;
}
int spada_dev_spgemm_numeric_plan(spada_ctx *, uint32_t chunks, uint64_t *chunk_pos)
{
    if (g_pos.size() != (size_t)chunks + 1) return SPADA_ERR_INVALID;
    std::memcpy(chunk_pos, g_pos.data(), g_pos.size() * 8);
    return SPADA_OK;
}
int spada_dev_spgemm_numeric_chunk(spada_ctx *, uint32_t k, void *d_c_indices, void *d_c_data, void **done_event)
{
    // piece k of the own block, written at its final place (the caller passes the block's first entry)
    for (uint64_t i = g_pos[k]; i < g_pos[k + 1]; ++i) {
        ((uint32_t *)d_c_indices)[i] = g_fill + (uint32_t)i;
        ((double *)d_c_data)[i] = (double)(g_fill + i) * 0.5;
    }
    *done_event = nullptr;
    return SPADA_OK;
}
int spada_dev_spgemm_indptr(spada_ctx *, void *d_c_indptr)
{
    std::memcpy(d_c_indptr, g_indptr.data(), g_indptr.size() * 8);
    return SPADA_OK;
}
int spada_dev_synchronize(spada_ctx *) { return SPADA_OK; }

// ---- what the test calls -------------------------------------------------------------------------------------------------
void spada_mock_begin(int pass)   // 1: roots deposit what they send (clears the deposits), 2: the other ranks receive it
{
    g_pass = pass;
    if (pass == 1) g_store.clear();
}
void spada_mock_reset(void)   // before every rank
{
    g_log.clear();
    g_bcast_no = 0;
    g_gathered.clear();
    g_pos.clear();
    g_indptr.clear();
    g_base_idx = g_base_val = g_base_ptr = nullptr;
    g_my_nnz = 0;
    g_fill = 0;
}
void spada_mock_set_gathered(const uint64_t *words, uint64_t n) { g_gathered.assign(words, words + n); }
void spada_mock_set_block(uint64_t nnz, const uint64_t *pos, uint32_t chunks, const uint64_t *indptr, uint64_t rows, uint32_t fill)
{
    g_my_nnz = nnz;
    g_pos.assign(pos, pos + chunks + 1);
    g_indptr.assign(indptr, indptr + rows + 1);
    g_fill = fill;
}
void spada_mock_set_bases(const void *idx, const void *val, const void *ptr)
{
    g_base_idx = (const char *)idx;
    g_base_val = (const char *)val;
    g_base_ptr = (const char *)ptr;
}
uint64_t spada_mock_log_size(void) { return g_log.size(); }
void spada_mock_log_get(uint64_t i, uint64_t *out5)
{
    const Call &c = g_log[i];
    out5[0] = c.op;
    out5[1] = c.root;
    out5[2] = c.count;
    out5[3] = c.dtype;
    out5[4] = c.offset;
}

}  // extern "C"
