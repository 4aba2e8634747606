// Device half of libspada_spgemm.so: engine context, HBM-resident CSR, the two-phase SpGEMM pipeline
// and its C ABI (include/spada_ffi.h).  The seam it replaces is Simulator::new / execute /
// get_exec_result of the reference (simulator.rs:431-507, :509-890, :1034-1062).
//
// There is no CPU fallback in this file: every compute entry point needs a gfx950 device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "spada_internal.hpp"
#include "spgemm_flat.hip.hpp"
#include "spgemm_task.hip.hpp"

using namespace spada;

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(e_ == hipErrorOutOfMemory ? SPADA_ERR_OOM : SPADA_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                                 \
    } while (0)

struct spada_dev_csr {
    uint64_t rows = 0, cols = 0, nnz = 0;
    uint64_t *ptr = nullptr;
    uint32_t *idx = nullptr;
    double *val = nullptr;
    uint32_t *rowid = nullptr;   // row of every entry: lets the row statistics run entry-parallel whatever the row lengths are
    DevCsrView view() const { return DevCsrView{ptr, idx, val, nullptr, nullptr}; }
};

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // grow-only; newly allocated memory is zeroed when `zero` is set (returns whether it reallocated)
    int ensure(size_t bytes, bool zero, hipStream_t s, size_t *accounted)
    {
        if (bytes <= cap) return SPADA_OK;
        if (p) {
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipFree(p));
            *accounted -= cap;
            p = nullptr;
            cap = 0;
        }
        size_t want = bytes + bytes / 4 + 256;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        *accounted += cap;
        if (zero) HIP_TRY(hipMemsetAsync(p, 0, cap, s));
        return SPADA_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T *as() const { return (T *)p; }
};

// device counter block, zeroed at the start of every symbolic call
struct Counters {
    uint32_t sym_counts[SPADA_N_BINS];
    uint32_t sym_cursor[SPADA_N_BINS];
    uint32_t num_counts[SPADA_N_BINS];
    uint32_t num_cursor[SPADA_N_BINS];
    unsigned long long totals[2];   // nprod, a_nnz of the row range
    uint32_t nb_sym, nb_num;        // number of flat batches (symbolic / numeric)
    uint32_t queue[4];              // dynamic dequeue cursors of the huge-row kernels
    unsigned long long num_sums[3 * SPADA_N_BINS];   // per numeric bin: products | nnz(C) | A entries
    unsigned long long sym_prod[SPADA_N_BINS];
};

enum { EV_SYM_BEGIN, EV_STATS, EV_BINNED, EV_SYM, EV_SCAN, EV_NUM_BEGIN, EV_NUM_END, EV_SFLAT_0, EV_SFLAT_1, EV_NFLAT_0, EV_NFLAT_1, EV_NMID_0, EV_NMID_1, EV_COUNT };

}  // namespace

struct spada_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int accumulator = SPADA_ACC_LDS_HASH;
    hipEvent_t ev[EV_COUNT] = {};
    // independent bins run concurrently: one side stream per bin, forked from / joined to `stream`
    hipStream_t side[SPADA_N_BINS] = {};
    hipEvent_t ev_fork = nullptr, ev_join[SPADA_N_BINS] = {};
    hipStream_t cur = nullptr;        // stream the launch helpers use
    bool merge_on = false;            // SPADA_MERGE=1: multiway-merge class (k_num_merge) for rows with <= 8 long B rows; measured
                                      // neutral on the webbase surrogate (1.77 ms either way), so off by default
    size_t lds_pad = 0;               // SPADA_LDS_PAD=<bytes>: occupancy experiments (fewer flat workgroups per CU)
    bool flat_big = false;            // SPADA_FLAT_BIG=1: flat kernel (list mode) instead of k_num_hash for rows above the mid class
    bool sort_huge = false;           // SPADA_SORT_HUGE=1
    int flat_cfg = 1;                 // SPADA_FLAT_CFG: 0 = 256 threads x 4 entries, 1 = 512 x 2, 2 = 1024 x 1
    int dbg_g = 0;                    // SPADA_DBG_G=<G>: phase timestamps of k_num_hash<G,*> into `dbg`
    DevBuf dbg;
    bool serial_bins = false;         // SPADA_SERIAL_BINS=1: one stream, for per-kernel profiling
    // state carried from symbolic to numeric
    bool have_symbolic = false;
    const spada_dev_csr *A = nullptr, *B = nullptr;
    uint64_t r0 = 0;
    uint32_t nrows = 0;
    uint64_t nnz_c = 0;
    uint32_t h_sym_counts[SPADA_N_BINS] = {}, h_num_counts[SPADA_N_BINS] = {};
    unsigned long long h_sym_prod[SPADA_N_BINS] = {};
    uint32_t h_nb_sym = 1;
    // workspace
    size_t ws_bytes = 0;
    DevBuf row_nprod, row_nnzc, row_bin, sym_rows, num_rows, counters, cptr, tile_sums, bitmaps, slabs;
    DevBuf own_idx, own_val, own_ptr, wide_idx;
    DevBuf efl;   // per A entry: first / last column of the selected B row
    DevBuf eb0, elen, row_kmin, row_kmax, batch_sym, batch_num, tile_w;
    uint32_t colbits = 0, rmax_eff = 0, num_flat_max = 0;
    bool flat_on = false;
    DevCsrView a_view() const
    {
        return DevCsrView{A->ptr, A->idx, A->val, eb0.as<uint64_t>(), elen.as<uint32_t>()};
    }
    uint64_t spill_slabs = 0, spill_cols = 0;
    bool bm_fits = false;             // LDS column bitmap fits for the current B
    uint32_t bm_vcap = 0;             // value-row capacity of k_num_bitmap<true> (0 = variant unused)
    Counters *h_counters = nullptr;   // pinned
    uint64_t *h_u64 = nullptr;        // pinned
    // task pipeline (spgemm_task.hip.hpp): the default; SPADA_PIPELINE=legacy selects the round-1 per-bin kernels
    bool use_tasks = true;
    DevBuf t_rowP, t_rowm, t_rowt, t_rowtmp, t_big, t_tiles, t_tmp, t_tasks, t_status, t_rangeout, t_scrcol, t_scrval, t_ctr;
    uint64_t t_cap_tmp = 0, t_cap_tasks = 0, t_cap_scr = 0;
    TaskCounters *h_tctr = nullptr;   // pinned
    hipEvent_t tev[6] = {};
    uint32_t n_cu = 256;
    // matrices uploaded by the host-pointer API
    spada_dev_csr *hA = nullptr, *hB = nullptr;
    spada_stats stats = {};
};

namespace {

template <class K>
int allow_lds(K kernel, size_t bytes)
{
    HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return SPADA_OK;
}

template <int G, int LOG_T>
constexpr size_t sym_lds() { return 128 + (size_t)((G <= 64 ? 256 : G) / G) * sym_row_bytes<G, LOG_T>(); }
template <int G, int LOG_T>
constexpr size_t num_lds() { return 128 + (size_t)((G <= 64 ? 256 : G) / G) * num_row_bytes<G, LOG_T>(); }

template <int G, int LOG_T>
int launch_sym(spada_ctx *c, uint32_t off, uint32_t n)
{
    if (!n) return SPADA_OK;
    constexpr int BLOCK = G <= 64 ? 256 : G;
    constexpr int RPB = BLOCK / G;
    const uint32_t grid = (n + RPB - 1) / RPB;
    hipLaunchKernelGGL((k_sym_hash<G, LOG_T>), dim3(grid), dim3(BLOCK), (sym_lds<G, LOG_T>()), c->cur, c->a_view(),
                       c->B->view(), c->r0, c->sym_rows.as<uint32_t>() + off, n, c->row_nnzc.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    return SPADA_OK;
}

template <int G, int LOG_T>
int launch_num(spada_ctx *c, uint32_t off, uint32_t n, uint32_t *c_idx, double *c_val)
{
    if (!n) return SPADA_OK;
    constexpr int BLOCK = G <= 64 ? 256 : G;
    constexpr int RPB = BLOCK / G;
    const uint32_t grid = (n + RPB - 1) / RPB;
    hipLaunchKernelGGL((k_num_hash<G, LOG_T>), dim3(grid), dim3(BLOCK), (num_lds<G, LOG_T>()), c->cur, c->a_view(),
                       c->B->view(), c->r0, c->num_rows.as<uint32_t>() + off, n, c->cptr.as<uint64_t>(), c_idx, c_val,
                       (G == c->dbg_g) ? c->dbg.as<unsigned long long>() : nullptr, c->row_kmin.as<uint32_t>(),
                       c->row_kmax.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    return SPADA_OK;
}

int ensure_spill(spada_ctx *c, uint32_t rows_in_bin, bool need_slabs)
{
    const uint64_t cols = c->B->cols;
    const uint64_t nslab = std::min<uint64_t>(rows_in_bin, 512);
    const uint64_t words = ((cols + 31) / 32 + 3) & ~3ull;
    if (c->spill_cols != cols || c->spill_slabs < nslab) {
        // geometry changed: drop and re-zero
        c->ws_bytes -= c->bitmaps.cap + c->slabs.cap;
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->bitmaps.release();
        c->slabs.release();
        c->spill_cols = cols;
        c->spill_slabs = nslab;
    }
    int rc = c->bitmaps.ensure(c->spill_slabs * words * 4, true, c->stream, &c->ws_bytes);
    if (rc) return rc;
    if (need_slabs) rc = c->slabs.ensure(c->spill_slabs * words * 4, false, c->stream, &c->ws_bytes);
    return rc;
}

// fork: side stream `k` starts after everything queued on the main stream so far
int fork_to(spada_ctx *c, int k)
{
    if (c->serial_bins) return SPADA_OK;
    HIP_TRY(hipStreamWaitEvent(c->side[k], c->ev_fork, 0));
    c->cur = c->side[k];
    return SPADA_OK;
}
// join: the main stream continues after side stream `k`
int join_from(spada_ctx *c, int k)
{
    if (c->serial_bins) return SPADA_OK;
    HIP_TRY(hipEventRecord(c->ev_join[k], c->side[k]));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join[k], 0));
    c->cur = c->stream;
    return SPADA_OK;
}

constexpr size_t LDS_MAX = 160 * 1024;

// persistent grid for the LDS bitmap kernels: as many workgroups as the LDS footprint admits per CU
uint32_t bm_grid(uint32_t rows, size_t lds)
{
    const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, LDS_MAX / lds));
    return std::min<uint32_t>(rows, 256u * per_cu);
}

float ev_ms(spada_ctx *c, int a, int b)
{
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev[a], c->ev[b]) != hipSuccess) return 0.f;
    return ms;
}

void dev_free(spada_dev_csr *m)
{
    if (!m) return;
    if (m->ptr) (void)hipFree(m->ptr);
    if (m->idx) (void)hipFree(m->idx);
    if (m->val) (void)hipFree(m->val);
    if (m->rowid) (void)hipFree(m->rowid);
    delete m;
}

int dev_upload(spada_ctx *c, const spada_csr_view *m, spada_dev_csr **out)
{
    if (m->cols >= 0xFFFFFFFFull)
        return fail(SPADA_ERR_UNSUPPORTED, "cols = %llu does not fit the 32-bit device column index",
                    (unsigned long long)m->cols);
    if (m->rows >= 0xFFFFFFFFull) return fail(SPADA_ERR_UNSUPPORTED, "rows = %llu >= 2^32", (unsigned long long)m->rows);
    auto d = std::make_unique<spada_dev_csr>();
    d->rows = m->rows;
    d->cols = m->cols;
    d->nnz = m->nnz;
    std::vector<uint32_t> idx32(m->nnz), rid32(m->nnz);
    for (uint64_t q = 0; q < m->nnz; ++q) idx32[q] = (uint32_t)m->indices[q];
    for (uint64_t r = 0; r < m->rows; ++r)
        for (uint64_t q = m->indptr[r]; q < m->indptr[r + 1]; ++q) rid32[q] = (uint32_t)r;
    HIP_TRY(hipMalloc((void **)&d->ptr, (m->rows + 1) * 8));
    if (hipMalloc((void **)&d->idx, std::max<uint64_t>(m->nnz, 1) * 4) != hipSuccess ||
        hipMalloc((void **)&d->val, std::max<uint64_t>(m->nnz, 1) * 8) != hipSuccess ||
        hipMalloc((void **)&d->rowid, std::max<uint64_t>(m->nnz, 1) * 4) != hipSuccess) {
        dev_free(d.release());
        return fail(SPADA_ERR_OOM, "hipMalloc failed for a %llu-nnz matrix", (unsigned long long)m->nnz);
    }
    const auto copy_in = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(d->ptr, m->indptr, (m->rows + 1) * 8, hipMemcpyHostToDevice, c->stream));
        if (m->nnz) {
            HIP_TRY(hipMemcpyAsync(d->idx, idx32.data(), m->nnz * 4, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d->val, m->data, m->nnz * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(d->rowid, rid32.data(), m->nnz * 4, hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        return SPADA_OK;
    };
    if (int rc = copy_in()) {
        dev_free(d.release());
        return rc;
    }
    *out = d.release();
    return SPADA_OK;
}

// flat-batch kernel configurations: <BLOCK, LOG_T, NOUT, RMAX>; cap = 2 * flat_max, NOUT >= cap + flat_max
#ifndef SPADA_NF_LARGE   // default: four 256-thread workgroups per CU, 2048-slot tables (A/B: -7 % numeric time on the regular surrogates vs two 512-thread workgroups with 4096-slot tables = SPADA_NF_LARGE)
constexpr int NF_LOG_T = 11, NF_NOUT = 1536, NF_RMAX = 128;
constexpr uint32_t NUM_FLAT_MAX = 512, NUM_FLAT_CAP = 1024;
#define NF_CFG_LIST(X) X(256, 2)
#else
constexpr int NF_LOG_T = 12, NF_NOUT = 3072, NF_RMAX = 256;
constexpr uint32_t NUM_FLAT_MAX = 1024, NUM_FLAT_CAP = 2048;
#define NF_CFG_LIST(X) X(256, 4) X(512, 2) X(1024, 1)
#endif
#ifndef SPADA_SF_LARGE   /* default: 256-thread workgroups, 4096-key tables (A/B: -17..21 % symbolic time) */
constexpr int SF_RMAX = 128;
#define SF_CFG_LIST(X) X(256, 2)
#else
constexpr int SF_RMAX = 256;
#define SF_CFG_LIST(X) X(256, 4) X(512, 2) X(1024, 1)
#endif
static_assert(NF_NOUT >= NUM_FLAT_CAP + NUM_FLAT_MAX, "a batch weighs less than cap + flat_max");
static_assert((1u << SYM_FLAT_LOG_T) * 3 >= (SYM_FLAT_CAP + SYM_FLAT_MAX) * 4, "symbolic table load <= 0.75");

uint32_t flat_grid(uint64_t nb_upper, size_t lds)
{
    const uint64_t per_cu = std::max<size_t>(1, std::min<size_t>(8, LDS_MAX / lds));
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nb_upper, 256ull * per_cu * 4));
}

int run_numeric(spada_ctx *c, uint64_t *d_ptr, uint32_t *d_idx, double *d_val)
{
    c->cur = c->stream;   // an earlier call may have failed between a fork and its join
    if (c->dbg.p && c->dbg_g != 2) HIP_TRY(hipMemsetAsync(c->dbg.p, 0, 64 * 16 * 8, c->stream));
    HIP_TRY(hipEventRecord(c->ev[EV_NUM_BEGIN], c->stream));
    HIP_TRY(hipMemcpyAsync(d_ptr, c->cptr.p, ((size_t)c->nrows + 1) * 8, hipMemcpyDeviceToDevice, c->stream));
    const uint32_t *cnt = c->h_num_counts;
    uint32_t off[SPADA_N_BINS + 1];
    off[0] = 0;
    for (int b = 0; b < SPADA_N_BINS; ++b) off[b + 1] = off[b] + (b == BIN_EMPTY || b == BIN_FLAT ? 0u : cnt[b]);
    int rc;
    Counters *dc = c->counters.as<Counters>();
    // the dequeue cursors of the huge-row kernels: a numeric call may be repeated after one symbolic call
    HIP_TRY(hipMemsetAsync(dc->queue, 0, sizeof dc->queue, c->stream));
    // spill slabs are (re)allocated and zeroed on the engine stream BEFORE the fork event, so the side stream sees them
    if (cnt[NUM2_BIN_SPILL] && !c->bm_fits && (rc = ensure_spill(c, cnt[NUM2_BIN_SPILL], true))) return rc;
    HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
    // heaviest bins first; every bin on its own stream
    if (cnt[NUM2_BIN_SPILL]) {
        const uint32_t nsp = cnt[NUM2_BIN_SPILL];
        if (c->bm_fits) {
            if ((rc = fork_to(c, NUM2_BIN_SPILL))) return rc;
            const size_t lds = bm_lds_bytes(c->B->cols, 0);
            hipLaunchKernelGGL(k_num_bitmap<false>, dim3(bm_grid(nsp, lds)), dim3(BM_BLOCK), lds, c->cur, c->a_view(),
                               c->B->view(), c->r0, c->num_rows.as<uint32_t>() + off[NUM2_BIN_SPILL], nsp, c->B->cols, 0u,
                               c->cptr.as<uint64_t>(), d_idx, d_val, &dc->queue[0]);
        } else {
            if ((rc = fork_to(c, NUM2_BIN_SPILL))) return rc;
            const uint64_t words = ((c->B->cols + 31) / 32 + 3) & ~3ull;
            const uint32_t grid = (uint32_t)std::min<uint64_t>(nsp, c->spill_slabs);
            hipLaunchKernelGGL(k_num_spill, dim3(grid), dim3(SPILL_BLOCK), 0, c->cur, c->a_view(), c->B->view(), c->r0,
                               c->num_rows.as<uint32_t>() + off[NUM2_BIN_SPILL], nsp, c->bitmaps.as<uint32_t>(),
                               c->slabs.as<uint32_t>(), words, c->cptr.as<uint64_t>(), d_idx, d_val);
        }
        HIP_TRY(hipGetLastError());
        if ((rc = join_from(c, NUM2_BIN_SPILL))) return rc;
    }
    if (cnt[NUM2_BIN_BMV]) {
        const uint32_t nb = cnt[NUM2_BIN_BMV];
        if ((rc = fork_to(c, NUM2_BIN_BMV))) return rc;
        const size_t lds = bm_lds_bytes(c->B->cols, c->bm_vcap);
        hipLaunchKernelGGL(k_num_bitmap<true>, dim3(bm_grid(nb, lds)), dim3(BM_BLOCK), lds, c->cur, c->a_view(),
                           c->B->view(), c->r0, c->num_rows.as<uint32_t>() + off[NUM2_BIN_BMV], nb, c->B->cols, c->bm_vcap,
                           c->cptr.as<uint64_t>(), d_idx, d_val, &dc->queue[1]);
        HIP_TRY(hipGetLastError());
        if ((rc = join_from(c, NUM2_BIN_BMV))) return rc;
    }
#define NUM_BIN(BIN, G, LT)                                                            \
    if (cnt[BIN]) {                                                                    \
        if ((rc = fork_to(c, BIN))) return rc;                                         \
        if ((rc = launch_num<G, LT>(c, off[BIN], cnt[BIN], d_idx, d_val))) return rc;  \
        if ((rc = join_from(c, BIN))) return rc;                                       \
    }
    if (c->flat_on && c->flat_big) {
        // A/B (measured 20 % slower than the per-row kernels below): rows above the mid class through the flat kernel, one row per
        // 1024-thread workgroup, 8192-slot table (list mode)
        for (int bin : {NUM2_BIN_6K, NUM2_BIN_2K})
            if (cnt[bin]) {
                if ((rc = fork_to(c, bin))) return rc;
                constexpr size_t lds = num_flat_lds<1024, 1, 13, 6144, 128>();
                hipLaunchKernelGGL((k_num_flat<1024, 1, 13, 6144, 128, true, 1>), dim3(flat_grid(cnt[bin], lds)), dim3(1024), lds,
                                   c->cur, c->A->ptr, c->A->val, c->B->idx, c->B->val, c->eb0.as<uint64_t>(),
                                   c->elen.as<uint32_t>(), c->r0, c->nrows, c->row_bin.as<uint8_t>(),
                                   c->row_kmin.as<uint32_t>(), c->row_kmax.as<uint32_t>(), c->cptr.as<uint64_t>(),
                                   c->batch_num.as<uint32_t>(), &dc->num_counts[bin], c->colbits, d_idx, d_val,
                                   (unsigned long long *)nullptr, c->num_rows.as<uint32_t>() + off[bin], (uint32_t)bin);
                HIP_TRY(hipGetLastError());
                if ((rc = join_from(c, bin))) return rc;
            }
    } else {
        NUM_BIN(NUM2_BIN_6K, 1024, 13)
        NUM_BIN(NUM2_BIN_2K, 256, 12)
    }
#undef NUM_BIN
    if (cnt[BIN_FLAT] && c->accumulator == SPADA_ACC_SORT_MERGE) {
        // sort-merge accumulator: the rows of the symbolic (product-weighted) batches
        if ((rc = fork_to(c, BIN_FLAT))) return rc;
        constexpr size_t lds = num_sm_lds<1024, 1, SF_RMAX>();
        HIP_TRY(hipEventRecord(c->ev[EV_NFLAT_0], c->cur));
        hipLaunchKernelGGL((k_num_sortmerge<1024, 1, SF_RMAX>), dim3(flat_grid(c->h_nb_sym, lds)), dim3(1024), lds, c->cur,
                           c->A->ptr, c->A->val, c->B->idx, c->B->val, c->eb0.as<uint64_t>(), c->elen.as<uint32_t>(), c->r0,
                           c->nrows, c->row_bin.as<uint8_t>(), c->cptr.as<uint64_t>(), c->batch_sym.as<uint32_t>(),
                           &dc->nb_sym, c->colbits, d_idx, d_val);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev[EV_NFLAT_1], c->cur));
        if ((rc = join_from(c, BIN_FLAT))) return rc;
    } else if (cnt[BIN_FLAT]) {
        if ((rc = fork_to(c, BIN_FLAT))) return rc;
#define LAUNCH_NUM_FLAT(BL, EP, LS, RP)                                                                                             \
    {                                                                                                                        \
        const size_t lds = num_flat_lds<BL, EP, NF_LOG_T, NF_NOUT, NF_RMAX>() + c->lds_pad;                                  \
        hipLaunchKernelGGL((k_num_flat<BL, EP, NF_LOG_T, NF_NOUT, NF_RMAX, LS, RP>), dim3(flat_grid(nf_batches, lds)),                \
                           dim3(BL), lds, c->cur, c->A->ptr, c->A->val, c->B->idx, c->B->val, c->eb0.as<uint64_t>(),          \
                           c->elen.as<uint32_t>(), c->r0, c->nrows, c->row_bin.as<uint8_t>(), c->row_kmin.as<uint32_t>(),     \
                           c->row_kmax.as<uint32_t>(), c->cptr.as<uint64_t>(), c->batch_num.as<uint32_t>(), nf_nb,            \
                           c->colbits, d_idx, d_val, nf_dbg, nf_list, nf_bin);                                               \
    }
#ifndef SPADA_NF_LARGE
#define NUM_FLAT_DISPATCH(LS, RP) LAUNCH_NUM_FLAT(256, 2, LS, RP)
#else
#define NUM_FLAT_DISPATCH(LS, RP)                          \
    if (c->flat_cfg == 0) LAUNCH_NUM_FLAT(256, 4, LS, RP)  \
    else if (c->flat_cfg == 1) LAUNCH_NUM_FLAT(512, 2, LS, RP) \
    else LAUNCH_NUM_FLAT(1024, 1, LS, RP)
#endif
        HIP_TRY(hipEventRecord(c->ev[EV_NFLAT_0], c->cur));
        {
            const uint64_t nf_batches = c->h_counters->nb_num;
            const uint32_t *nf_nb = &dc->nb_num, *nf_list = nullptr;
            const uint32_t nf_bin = BIN_FLAT;
            unsigned long long *nf_dbg = c->dbg_g == 1 ? c->dbg.as<unsigned long long>() : nullptr;
            NUM_FLAT_DISPATCH(false, 1)
        }
        HIP_TRY(hipEventRecord(c->ev[EV_NFLAT_1], c->cur));
        HIP_TRY(hipGetLastError());
        if ((rc = join_from(c, BIN_FLAT))) return rc;
    }
    for (int pass = 0; pass < 2; ++pass) {   // mid rows from their row lists: one row per batch, or two of the lower half
        const int bin = pass == 0 ? NUM2_BIN_MID : NUM2_BIN_MID2;
        if (!cnt[bin]) continue;
        if ((rc = fork_to(c, bin))) return rc;
        const uint64_t nf_batches = (cnt[bin] + pass) / (pass + 1);
        const uint32_t *nf_nb = &dc->num_counts[bin], *nf_list = c->num_rows.as<uint32_t>() + off[bin];
        const uint32_t nf_bin = (uint32_t)bin;
        unsigned long long *nf_dbg = nullptr;
        if (pass == 0) HIP_TRY(hipEventRecord(c->ev[EV_NMID_0], c->cur));
        if (pass == 0) { NUM_FLAT_DISPATCH(true, 1) } else { NUM_FLAT_DISPATCH(true, 2) }
        HIP_TRY(hipGetLastError());
        if (pass == 0) HIP_TRY(hipEventRecord(c->ev[EV_NMID_1], c->cur));
        if ((rc = join_from(c, bin))) return rc;
    }
#undef LAUNCH_NUM_FLAT
#define LAUNCH_MERGE(BIN, PM)                                                                                               \
    if (cnt[BIN]) {                                                                                                        \
        if ((rc = fork_to(c, BIN))) return rc;                                                                             \
        constexpr size_t lds = 4 * ((num_merge_wave_bytes<PM>() + 15) & ~(size_t)15);                                      \
        const uint32_t grid = (uint32_t)std::min<uint64_t>((cnt[BIN] + 3) / 4, 256ull * (LDS_MAX / lds) * 4);               \
        hipLaunchKernelGGL(k_num_merge<PM>, dim3(grid), dim3(256), lds, c->cur, c->A->ptr, c->A->val, c->B->idx, c->B->val,  \
                           c->eb0.as<uint64_t>(), c->elen.as<uint32_t>(), c->r0, c->num_rows.as<uint32_t>() + off[BIN],     \
                           cnt[BIN], c->cptr.as<uint64_t>(), d_idx, d_val);                                               \
        HIP_TRY(hipGetLastError());                                                                                        \
        if ((rc = join_from(c, BIN))) return rc;                                                                           \
    }
    LAUNCH_MERGE(NUM2_BIN_MERGE_L, 1024)
    LAUNCH_MERGE(NUM2_BIN_MERGE_S, 512)
#undef LAUNCH_MERGE
    if (cnt[BIN_COPY]) {
        if ((rc = fork_to(c, BIN_COPY))) return rc;
        const uint32_t grid = std::min<uint32_t>((c->nrows + 255) / 256, 256u * 8 * 4);
        hipLaunchKernelGGL(k_num_copy2, dim3(grid), dim3(256), 0, c->cur, c->A->ptr, c->A->val, c->B->idx, c->B->val,
                           c->eb0.as<uint64_t>(), c->elen.as<uint32_t>(), c->r0, c->nrows, c->row_bin.as<uint8_t>(),
                           c->cptr.as<uint64_t>(), d_idx, d_val);
        HIP_TRY(hipGetLastError());
        if ((rc = join_from(c, BIN_COPY))) return rc;
    }
    HIP_TRY(hipEventRecord(c->ev[EV_NUM_END], c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stats.ms_numeric = c->stats.ms_numeric_call = ev_ms(c, EV_NUM_BEGIN, EV_NUM_END);
    c->stats.ms_num_flat = cnt[BIN_FLAT] ? ev_ms(c, EV_NFLAT_0, EV_NFLAT_1) : 0.0;
    c->stats.ms_num_mid = cnt[NUM2_BIN_MID] ? ev_ms(c, EV_NMID_0, EV_NMID_1) : 0.0;
    c->stats.workspace_bytes = c->ws_bytes;
    return SPADA_OK;
}


// ---- task pipeline ---------------------------------------------------------------------------------------------------
float tev_ms(spada_ctx *c, int a, int b)
{
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->tev[a], c->tev[b]) != hipSuccess) return 0.f;
    return ms;
}

template <int MODE>
void launch_task(spada_ctx *c, const TaskArgs &g)
{
    hipLaunchKernelGGL(k_task<MODE>, dim3(c->n_cu * 4), dim3(TK_BLOCK), task_lds(), c->stream, g);
}

TaskArgs task_args(spada_ctx *c, uint64_t *cptr, uint32_t *d_idx, double *d_val, uint64_t capacity)
{
    TaskArgs g;
    g.aptr = c->A->ptr;
    g.aval = c->A->val;
    g.bidx = c->B->idx;
    g.bval = c->B->val;
    g.eb0 = c->eb0.as<uint64_t>();
    g.elen = c->elen.as<uint32_t>();
    g.r0 = c->r0;
    g.nrows = c->nrows;
    g.colbits = c->colbits;
    g.row_cls = c->row_bin.as<uint8_t>();
    g.row_kmin = c->row_kmin.as<uint32_t>();
    g.row_kmax = c->row_kmax.as<uint32_t>();
    g.tasks = c->t_tasks.as<TaskDesc>();
    g.scr_col = c->t_scrcol.as<uint32_t>();
    g.scr_val = c->t_scrval.as<double>();
    g.cptr = cptr;
    g.range_out = c->t_rangeout.as<uint64_t>();
    g.status = c->t_status.as<unsigned long long>();
    g.ctr = c->t_ctr.as<TaskCounters>();
    g.c_idx = d_idx;
    g.c_val = d_val;
    g.capacity = capacity;
    return g;
}

// Row statistics, BIG-row expansion, task list and the task kernel in MODE_COUNT (cptr = the context's C.indptr) or MODE_FUSED
// (cptr / d_idx / d_val = the caller's buffers).  Nothing is read back before the end; workspaces whose size depends on the
// data (tasks, range descriptors, scratch) keep their capacity from earlier calls, the kernels refuse to overrun them, and one
// more run with the sizes they report follows when that happened.
int task_pipeline(spada_ctx *c, int mode, uint64_t *cptr, uint32_t *d_idx, double *d_val, uint64_t capacity)
{
    const spada_dev_csr *a = c->A, *b = c->B;
    const uint32_t n = c->nrows;
    hipStream_t s = c->stream;
    int rc;
    (void)hipGetLastError();   // an error an earlier call already reported must not be seen by the launch checks below
    const size_t n1 = (size_t)n + 1;
    if ((rc = c->row_nprod.ensure(n1 * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->row_bin.ensure(n1, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->row_kmin.ensure(n1 * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->row_kmax.ensure(n1 * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->t_rowP.ensure(n1 * 8, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->t_rowm.ensure(n1 * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->t_rowt.ensure(n1 * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->t_rowtmp.ensure(n1 * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->t_big.ensure(n1 * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->cptr.ensure(n1 * 8, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->t_ctr.ensure(sizeof(TaskCounters), false, s, &c->ws_bytes))) return rc;
    if ((rc = c->eb0.ensure(std::max<uint64_t>(a->nnz, 1) * 8, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->elen.ensure(std::max<uint64_t>(a->nnz, 1) * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->efl.ensure(std::max<uint64_t>(a->nnz, 1) * 8, false, s, &c->ws_bytes))) return rc;
    const uint32_t ntiles = std::max<uint32_t>((n + CUT_TILE - 1) / CUT_TILE, 1);
    if ((rc = c->t_tiles.ensure(((size_t)ntiles + 2) * 4, false, s, &c->ws_bytes))) return rc;
    // composite hash keys of a batch: (local row << colbits) | column
    uint32_t cb = 0;
    while (cb < 32 && (1ull << cb) < b->cols) ++cb;
    c->colbits = cb;
    const uint32_t rmax = cb >= 32 ? 1u : (uint32_t)std::min<uint64_t>((1ull << (32 - cb)) - 1, TK_RMAX);
    if (cptr == nullptr) cptr = c->cptr.as<uint64_t>();
    TaskCounters *dc = c->t_ctr.as<TaskCounters>();
    if (!c->t_cap_tasks) {
        c->t_cap_tasks = n / 4 + 4096;
        c->t_cap_tmp = 4096;
        c->t_cap_scr = 1u << 20;
    }
    c->stats.pipeline_runs = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        c->t_cap_tasks = std::max<uint64_t>(c->t_cap_tasks, (uint64_t)n / 4 + 4096);
        if ((rc = c->t_tasks.ensure(c->t_cap_tasks * sizeof(TaskDesc), false, s, &c->ws_bytes))) return rc;
        if ((rc = c->t_status.ensure(c->t_cap_tasks * 8, false, s, &c->ws_bytes))) return rc;
        if ((rc = c->t_rangeout.ensure(c->t_cap_tasks * 8, false, s, &c->ws_bytes))) return rc;
        if ((rc = c->t_tmp.ensure(c->t_cap_tmp * sizeof(TaskDesc), false, s, &c->ws_bytes))) return rc;
        if ((rc = c->t_scrcol.ensure(c->t_cap_scr * 4, false, s, &c->ws_bytes))) return rc;
        if ((rc = c->t_scrval.ensure(c->t_cap_scr * 8, false, s, &c->ws_bytes))) return rc;
        // capacities the kernels may rely on (DevBuf over-allocates; use what was asked for)
        const uint32_t cap_tasks = (uint32_t)std::min<uint64_t>(c->t_cap_tasks, 0xFFFFFFF0u);
        const uint32_t cap_tmp = (uint32_t)std::min<uint64_t>(c->t_cap_tmp, 0xFFFFFFF0u);
        ++c->stats.pipeline_runs;
        HIP_TRY(hipEventRecord(c->tev[0], s));
        HIP_TRY(hipMemsetAsync(dc, 0, sizeof(TaskCounters), s));
        if (n) {
            HIP_TRY(hipMemsetAsync(c->t_rowP.p, 0, (size_t)n * 8, s));
            HIP_TRY(hipMemsetAsync(c->row_kmin.p, 0xFF, (size_t)n * 4, s));
            HIP_TRY(hipMemsetAsync(c->row_kmax.p, 0, (size_t)n * 4, s));
            const uint32_t gent = (uint32_t)std::min<uint64_t>((a->nnz + 255) / 256 + 1, (uint64_t)c->n_cu * 8 * 4);
            hipLaunchKernelGGL(k_entry_stats, dim3(gent), dim3(256), 0, s, a->ptr, a->idx, a->rowid, b->ptr, b->idx, c->r0, n,
                               c->eb0.as<uint64_t>(), c->elen.as<uint32_t>(), c->t_rowP.as<unsigned long long>(),
                               c->row_kmin.as<uint32_t>(), c->row_kmax.as<uint32_t>());
            hipLaunchKernelGGL(k_row_class, dim3(std::min<uint32_t>((n + 255) / 256, c->n_cu)), dim3(256), 0, s, a->ptr, c->r0,
                               n, rmax, c->t_rowP.as<unsigned long long>(), c->row_nprod.as<uint32_t>(), c->row_bin.as<uint8_t>(),
                               c->t_rowm.as<uint32_t>(), c->t_big.as<uint32_t>(), dc);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(c->tev[1], s));
        if (n) {
            hipLaunchKernelGGL(k_big_expand, dim3(c->n_cu * 5), dim3(TK_BLOCK), BX_LDS, s, a->ptr, a->val, b->idx, b->val,
                               c->eb0.as<uint64_t>(), c->elen.as<uint32_t>(), c->r0, c->t_big.as<uint32_t>(),
                               c->row_nprod.as<uint32_t>(), c->row_kmin.as<uint32_t>(), c->row_kmax.as<uint32_t>(),
                               c->t_rowm.as<uint32_t>(), c->t_rowtmp.as<uint32_t>(), c->t_tmp.as<TaskDesc>(), cap_tmp,
                               c->t_scrcol.as<uint32_t>(), c->t_scrval.as<double>(), c->t_cap_scr, dc);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(c->tev[2], s));
        if (n) {
            hipLaunchKernelGGL(k_cut1, dim3(ntiles), dim3(256), 0, s, c->row_bin.as<uint8_t>(), c->row_nprod.as<uint32_t>(),
                               c->t_rowm.as<uint32_t>(), n, rmax, c->t_tiles.as<uint32_t>(), c->t_rowt.as<uint32_t>());
            hipLaunchKernelGGL(k_cut2, dim3(1), dim3(256), 0, s, c->t_tiles.as<uint32_t>(), ntiles, cap_tasks, dc);
            hipLaunchKernelGGL(k_cut3, dim3(ntiles), dim3(256), 0, s, c->row_bin.as<uint8_t>(), c->t_rowt.as<uint32_t>(),
                               c->t_rowtmp.as<uint32_t>(), n, c->t_tiles.as<uint32_t>(),
                               c->t_tmp.as<TaskDesc>(), c->t_tasks.as<TaskDesc>(), cap_tasks, dc);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemsetAsync(c->t_status.p, 0, (size_t)cap_tasks * 8, s));
        HIP_TRY(hipEventRecord(c->tev[3], s));
        if (n) {
            const TaskArgs g = task_args(c, cptr, d_idx, d_val, capacity);
            if (mode == MODE_COUNT) launch_task<MODE_COUNT>(c, g);
            else launch_task<MODE_FUSED>(c, g);
            HIP_TRY(hipGetLastError());
        } else {
            HIP_TRY(hipMemsetAsync(cptr, 0, 8, s));
        }
        HIP_TRY(hipEventRecord(c->tev[4], s));
        HIP_TRY(hipMemcpyAsync(c->h_tctr, dc, sizeof(TaskCounters), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        TaskCounters &h = *c->h_tctr;
        h.nprod = 0;
        for (int k = 0; k < N_CLS; ++k) h.nprod += h.cls_prod[k];
        h.nprod_big = h.cls_prod[CLS_BIG];
        if (!h.abort_flag) break;
        if (attempt == 2) return fail(SPADA_ERR_HIP, "task pipeline: workspaces still too small after two retries (flag %u)", h.abort_flag);
        if (h.abort_flag & 4u) return fail(SPADA_ERR_UNSUPPORTED, "a row of C has 2^32 or more products");
        c->t_cap_scr = std::max<uint64_t>(c->t_cap_scr, h.nprod_big + h.nprod_big / 16 + 1024);
        c->t_cap_tmp = std::max<uint64_t>(c->t_cap_tmp, (uint64_t)h.tmp_cursor + h.tmp_cursor / 16 + 1024);
        c->t_cap_tasks = std::max<uint64_t>(c->t_cap_tasks, (uint64_t)h.need_tasks + h.need_tasks / 16 + 1024);
    }
    const TaskCounters &h = *c->h_tctr;
    c->nnz_c = n ? h.nnz_c : 0;
    spada_stats &st = c->stats;
    st.rows = n;
    st.a_nnz = h.a_nnz;
    st.b_nnz = b->nnz;
    st.nprod = h.nprod;
    st.c_nnz = c->nnz_c;
    st.bytes_read = ((uint64_t)n + 1) * 8 + h.a_nnz * 12 + h.a_nnz * 16 + h.nprod * 12;
    st.bytes_write = ((uint64_t)n + 1) * 8 + c->nnz_c * 12;
    st.ms_row_stats = tev_ms(c, 0, 1);
    st.ms_big_expand = tev_ms(c, 1, 2);
    st.ms_cut = tev_ms(c, 2, 3);
    st.ms_task = tev_ms(c, 3, 4);
    for (int k = 0; k < N_CLS; ++k) {
        st.cls_rows[k] = st.num_bin_rows[k] = st.sym_bin_rows[k] = h.cls_rows[k];
        st.cls_prod[k] = st.num_bin_prod[k] = st.sym_bin_prod[k] = h.cls_prod[k];
    }
    if (SPADA_TASK_DBG)
        std::fprintf(stderr, "[task dbg] tasks %u  loop cycles/WG %.0f  accumulate %.1f%%  chain %.1f%%  emit %.1f%%  windows/task %.2f  spins/task %.2f  waits/task %.2f  mean distance of the awaited task %.1f\n",
                     h.ntasks, (double)h.dbg[3] / (c->n_cu * 4.0), 100.0 * h.dbg[4] / std::max<double>(1, h.dbg[3]),
                     100.0 * h.dbg[0] / std::max<double>(1, h.dbg[3]), 100.0 * h.dbg[5] / std::max<double>(1, h.dbg[3]),
                     (double)h.dbg[1] / std::max(1u, h.ntasks), (double)h.dbg[2] / std::max(1u, h.ntasks),
                     (double)h.dbg[7] / std::max(1u, h.ntasks), (double)h.dbg[6] / std::max<double>(1, h.dbg[7]));
    if (SPADA_TASK_DBG && h.dbg[15])
        std::fprintf(stderr, "[batch dbg] %llu batches, cycles each: descriptor %.0f | rows+scan+clear %.0f | walk %.0f | counts+publish %.0f | "
                     "emit (LDS, look-back wait, stores) %.0f | copy rows %.0f | ticket %.0f\n", h.dbg[15], (double)h.dbg[8] / h.dbg[15],
                     (double)h.dbg[9] / h.dbg[15], (double)h.dbg[10] / h.dbg[15], (double)h.dbg[11] / h.dbg[15],
                     (double)h.dbg[12] / h.dbg[15], (double)h.dbg[13] / h.dbg[15], (double)h.dbg[14] / h.dbg[15]);
    st.n_tasks = h.ntasks;
    st.multi_pass_tasks = h.multi_pass_tasks;
    st.scratch_products = h.nprod_big;
    st.spill_rows = h.n_big;
    st.workspace_bytes = c->ws_bytes;
    if (mode == MODE_COUNT) {
        st.ms_symbolic_call = tev_ms(c, 0, 4);
        st.ms_symbolic = st.ms_task;
    } else {
        st.ms_fused_call = tev_ms(c, 0, 4);
        st.ms_numeric = st.ms_task;
    }
    return SPADA_OK;
}

int task_numeric(spada_ctx *c, uint64_t *d_ptr, uint32_t *d_idx, double *d_val)
{
    hipStream_t s = c->stream;
    HIP_TRY(hipEventRecord(c->tev[0], s));
    HIP_TRY(hipMemcpyAsync(d_ptr, c->cptr.p, ((size_t)c->nrows + 1) * 8, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemsetAsync(c->t_ctr.as<TaskCounters>()->ticket, 0, sizeof(TaskCounters::ticket), s));
    if (c->nrows) {
        const TaskArgs g = task_args(c, c->cptr.as<uint64_t>(), d_idx, d_val, c->nnz_c);
        launch_task<MODE_NUMERIC>(c, g);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(c->tev[4], s));
    HIP_TRY(hipStreamSynchronize(s));
    c->stats.ms_numeric = c->stats.ms_numeric_call = c->stats.ms_task = tev_ms(c, 0, 4);
    c->stats.workspace_bytes = c->ws_bytes;
    return SPADA_OK;
}

}  // namespace

extern "C" {

int spada_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int spada_create(const spada_options *opts, spada_ctx **out)
{
    if (!out) return fail(SPADA_ERR_INVALID, "spada_create: null out");
    *out = nullptr;
    spada_options o{sizeof(spada_options), -1, SPADA_ACC_LDS_HASH, 0};
    if (opts) {
        if (opts->struct_size != sizeof(spada_options))
            return fail(SPADA_ERR_INVALID, "spada_options.struct_size = %u, expected %zu", opts->struct_size,
                        sizeof(spada_options));
        o = *opts;
    }
    if (o.accumulator != SPADA_ACC_LDS_HASH && o.accumulator != SPADA_ACC_SORT_MERGE)
        return fail(SPADA_ERR_INVALID, "unknown accumulator %d", o.accumulator);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return fail(SPADA_ERR_NO_DEVICE, "no HIP device visible: this engine has no CPU path");
    int dev = o.device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= n) return fail(SPADA_ERR_NO_DEVICE, "device %d requested but only %d visible", dev, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SPADA_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 (MI355X) only", dev,
                    prop.gcnArchName);
    HIP_TRY(hipSetDevice(dev));
    // a failure below releases whatever was created so far (streams, events, pinned memory) through spada_destroy
    std::unique_ptr<spada_ctx, void (*)(spada_ctx *)> c(new spada_ctx, spada_destroy);
    c->device = dev;
    c->accumulator = o.accumulator;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
    {
        // the one-workgroup-per-CU kernels (large / huge rows) need a whole CU's LDS: give their streams priority, otherwise
        // the many small workgroups of the flat kernels keep every CU partly occupied and starve them until the end
        int prio_lo = 0, prio_hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        const bool use_prio = !(std::getenv("SPADA_NO_PRIO") && std::getenv("SPADA_NO_PRIO")[0] == '1');
        for (int k = 0; k < SPADA_N_BINS; ++k) {
            const bool big = k == NUM2_BIN_6K || k == NUM2_BIN_2K || k == NUM2_BIN_BMV || k == NUM2_BIN_SPILL;
            HIP_TRY(hipStreamCreateWithPriority(&c->side[k], hipStreamNonBlocking, big && use_prio ? prio_hi : 0));   // 0 = default priority
        }
    }
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (auto &e : c->ev_join) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->cur = c->stream;
    if (const char *e = std::getenv("SPADA_SERIAL_BINS")) c->serial_bins = e[0] == '1';
    if (const char *e = std::getenv("SPADA_DBG_G")) {
        c->dbg_g = atoi(e);
        int rc0 = c->dbg.ensure(64 * 16 * 8, true, c->stream, &c->ws_bytes);
        if (rc0) return rc0;
    }
    HIP_TRY(hipHostMalloc((void **)&c->h_counters, sizeof(Counters), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&c->h_tctr, sizeof(TaskCounters), hipHostMallocDefault));
    for (auto &e : c->tev) HIP_TRY(hipEventCreate(&e));
    c->n_cu = (uint32_t)std::max(1, prop.multiProcessorCount);
    if (const char *e = std::getenv("SPADA_PIPELINE")) c->use_tasks = std::strcmp(e, "legacy") != 0;
    if (o.accumulator == SPADA_ACC_SORT_MERGE) c->use_tasks = false;   // the sort-merge accumulator still runs on the per-bin kernels
    HIP_TRY(hipHostMalloc((void **)&c->h_u64, 64, hipHostMallocDefault));
    int rc;
    if ((rc = allow_lds(k_sym_hash<512, 14>, sym_lds<512, 14>()))) return rc;
    if ((rc = allow_lds(k_sym_hash<1024, 15>, sym_lds<1024, 15>()))) return rc;
    if ((rc = allow_lds(k_num_hash<256, 12>, num_lds<256, 12>()))) return rc;
    if ((rc = allow_lds(k_num_hash<1024, 13>, num_lds<1024, 13>()))) return rc;
#define ALLOW_FLAT(BL, EP)                                                                                                   \
    if ((rc = allow_lds(k_sym_flat<BL, EP, SYM_FLAT_LOG_T, SF_RMAX, false, 1>, sym_flat_lds<BL, EP, SYM_FLAT_LOG_T, SF_RMAX>()))) return rc; \
    if ((rc = allow_lds(k_sym_flat<BL, EP, SYM_FLAT_LOG_T, SF_RMAX, true, 1>, sym_flat_lds<BL, EP, SYM_FLAT_LOG_T, SF_RMAX>()))) return rc; \
    if ((rc = allow_lds(k_sym_flat<BL, EP, SYM_FLAT_LOG_T, SF_RMAX, true, 2>, sym_flat_lds<BL, EP, SYM_FLAT_LOG_T, SF_RMAX>()))) return rc;

    SF_CFG_LIST(ALLOW_FLAT)
#undef ALLOW_FLAT
#define ALLOW_NFLAT(BL, EP)                                                                                                  \
    if ((rc = allow_lds(k_num_flat<BL, EP, NF_LOG_T, NF_NOUT, NF_RMAX, false, 1>, LDS_MAX))) return rc; \
    if ((rc = allow_lds(k_num_flat<BL, EP, NF_LOG_T, NF_NOUT, NF_RMAX, true, 1>, LDS_MAX))) return rc; \
    if ((rc = allow_lds(k_num_flat<BL, EP, NF_LOG_T, NF_NOUT, NF_RMAX, true, 2>, LDS_MAX))) return rc;
    NF_CFG_LIST(ALLOW_NFLAT)
#undef ALLOW_NFLAT
    if ((rc = allow_lds(k_num_sortmerge<1024, 1, SF_RMAX>, num_sm_lds<1024, 1, SF_RMAX>()))) return rc;
    if (const char *e = std::getenv("SPADA_FLAT_CFG")) c->flat_cfg = atoi(e);
    if (const char *e = std::getenv("SPADA_SORT_HUGE")) c->sort_huge = e[0] == '1';
    if (const char *e = std::getenv("SPADA_FLAT_BIG")) c->flat_big = e[0] == '1';
    if (const char *e = std::getenv("SPADA_LDS_PAD")) c->lds_pad = (size_t)atol(e);
    if (const char *e = std::getenv("SPADA_MERGE")) c->merge_on = e[0] == '1';
    if ((rc = allow_lds(k_num_merge<512>, 4 * ((num_merge_wave_bytes<512>() + 15) & ~(size_t)15)))) return rc;
    if ((rc = allow_lds(k_num_merge<1024>, 4 * ((num_merge_wave_bytes<1024>() + 15) & ~(size_t)15)))) return rc;
    if ((rc = allow_lds(k_num_flat<1024, 1, 13, 6144, 128, true, 1>, num_flat_lds<1024, 1, 13, 6144, 128>()))) return rc;
    if ((rc = allow_lds(k_task<MODE_COUNT>, task_lds()))) return rc;
    if ((rc = allow_lds(k_task<MODE_NUMERIC>, task_lds()))) return rc;
    if ((rc = allow_lds(k_task<MODE_FUSED>, task_lds()))) return rc;
    if ((rc = allow_lds(k_big_expand, BX_LDS))) return rc;
    if ((rc = allow_lds(k_sym_bitmap, LDS_MAX))) return rc;
    if ((rc = allow_lds(k_num_bitmap<true>, LDS_MAX))) return rc;
    if ((rc = allow_lds(k_num_bitmap<false>, LDS_MAX))) return rc;
    *out = c.release();
    return SPADA_OK;
}

void spada_destroy(spada_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    dev_free(c->hA);
    if (c->hB != c->hA) dev_free(c->hB);
    for (DevBuf *b : {&c->row_nprod, &c->row_nnzc, &c->row_bin, &c->sym_rows, &c->num_rows, &c->counters, &c->cptr,
                      &c->tile_sums, &c->bitmaps, &c->slabs, &c->own_idx, &c->own_val, &c->own_ptr, &c->wide_idx,
                      &c->eb0, &c->elen, &c->efl, &c->row_kmin, &c->row_kmax, &c->batch_sym, &c->batch_num, &c->tile_w, &c->dbg,
                      &c->t_rowP, &c->t_rowm, &c->t_rowt, &c->t_rowtmp, &c->t_big, &c->t_tiles, &c->t_tmp, &c->t_tasks, &c->t_status, &c->t_rangeout,
                      &c->t_scrcol, &c->t_scrval, &c->t_ctr})
        b->release();
    if (c->h_tctr) (void)hipHostFree(c->h_tctr);
    for (auto &e : c->tev)
        if (e) (void)hipEventDestroy(e);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    if (c->h_u64) (void)hipHostFree(c->h_u64);
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_join)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (auto &st : c->side)
        if (st) (void)hipStreamDestroy(st);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int spada_dev_csr_upload(spada_ctx *c, const spada_csr_view *m, spada_dev_csr **out)
{
    if (!c) return fail(SPADA_ERR_STATE, "spada_dev_csr_upload: no engine context (no GPU?)");
    if (!out) return fail(SPADA_ERR_INVALID, "spada_dev_csr_upload: null out");
    *out = nullptr;
    int rc = spada_csr_validate(m);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    return dev_upload(c, m, out);
}

void spada_dev_csr_free(spada_ctx *c, spada_dev_csr *m)
{
    if (c) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (c->A == m || c->B == m) c->have_symbolic = false;
    }
    dev_free(m);
}

int spada_dev_spgemm_symbolic(spada_ctx *c, const spada_dev_csr *a, const spada_dev_csr *b, uint64_t row_begin,
                              uint64_t row_end, uint64_t *nnz_c)
{
    if (!c) return fail(SPADA_ERR_STATE, "spada_dev_spgemm_symbolic: no engine context (no GPU?)");
    if (!a || !b || !nnz_c) return fail(SPADA_ERR_INVALID, "spada_dev_spgemm_symbolic: null argument");
    if (a->cols != b->rows)
        return fail(SPADA_ERR_INVALID, "inner dimensions differ: A is %llux%llu, B is %llux%llu", (unsigned long long)a->rows,
                    (unsigned long long)a->cols, (unsigned long long)b->rows, (unsigned long long)b->cols);
    if (row_begin > row_end || row_end > a->rows) return fail(SPADA_ERR_INVALID, "bad row range");
    HIP_TRY(hipSetDevice(c->device));
    c->have_symbolic = false;
    c->cur = c->stream;
    c->A = a;
    c->B = b;
    c->r0 = row_begin;
    c->nrows = (uint32_t)(row_end - row_begin);
    c->nnz_c = 0;
    std::memset(&c->stats, 0, sizeof c->stats);
    if (c->use_tasks) {
        const int rc_t = task_pipeline(c, MODE_COUNT, nullptr, nullptr, nullptr, 0);
        if (rc_t) return rc_t;
        *nnz_c = c->nnz_c;
        c->have_symbolic = true;
        return SPADA_OK;
    }
    {
        const size_t base = bm_lds_bytes(b->cols, 0);
        c->bm_fits = base <= LDS_MAX;
        c->bm_vcap = 0;
        if (c->bm_fits && LDS_MAX - base >= 8192 * 8) c->bm_vcap = (uint32_t)std::min<size_t>(16384, (LDS_MAX - base) / 8);
    }
    const uint32_t n = c->nrows;
    hipStream_t s = c->stream;
    int rc;
    const size_t n1 = (size_t)n + 1;
    if ((rc = c->row_nprod.ensure(n1 * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->row_nnzc.ensure(n1 * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->row_bin.ensure(n1, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->row_kmin.ensure(n1 * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->row_kmax.ensure(n1 * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->sym_rows.ensure(n1 * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->num_rows.ensure(n1 * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->cptr.ensure(n1 * 8, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->counters.ensure(sizeof(Counters), false, s, &c->ws_bytes))) return rc;
    if ((rc = c->eb0.ensure(std::max<uint64_t>(a->nnz, 1) * 8, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->elen.ensure(std::max<uint64_t>(a->nnz, 1) * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->efl.ensure(std::max<uint64_t>(a->nnz, 1) * 8, false, s, &c->ws_bytes))) return rc;
    // batch b starts at row batch_first[b]; b <= sum of weights / cap <= rows (a row weighs at most cap)
    if ((rc = c->batch_sym.ensure((n1 + 4) * 4, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->batch_num.ensure((n1 + 4) * 4, false, s, &c->ws_bytes))) return rc;
    const uint32_t ntiles = std::max<uint32_t>((n + SCAN_TILE - 1) / SCAN_TILE, 1);
    if ((rc = c->tile_sums.ensure(((size_t)ntiles + 2) * 8, false, s, &c->ws_bytes))) return rc;
    if ((rc = c->tile_w.ensure(((size_t)ntiles + 2) * 8, false, s, &c->ws_bytes))) return rc;
    Counters *dc = c->counters.as<Counters>();

    // composite hash keys of the flat batches: (local row << colbits) | column
    {
        uint32_t cb = 0;
        while (cb < 32 && (1ull << cb) < b->cols) ++cb;
        c->colbits = cb;
        const uint64_t lim = cb >= 32 ? 1 : std::min<uint64_t>((1ull << (32 - cb)) - 1, 256);
        c->rmax_eff = (uint32_t)lim;
        c->flat_on = lim >= 4;
        c->num_flat_max = c->flat_on ? NUM_FLAT_MAX : 0;
    }
    const uint32_t rmax = std::max<uint32_t>(c->rmax_eff, 4);
    const uint32_t rmax_s = std::min<uint32_t>(rmax, SF_RMAX);
    const CutParams cut_sym{SYM_FLAT_CAP, (SYM_FLAT_CAP + rmax_s - 1) / rmax_s, 0, 0, 0, 0};
    const bool sort_merge = c->accumulator == SPADA_ACC_SORT_MERGE && c->flat_on;
    const uint32_t rmax_n = std::min<uint32_t>(rmax, NF_RMAX);
    const CutParams cut_num{NUM_FLAT_CAP, (NUM_FLAT_CAP + rmax_n - 1) / rmax_n, c->num_flat_max, c->bm_vcap,
                            sort_merge ? SYM_FLAT_MAX : 0u, c->merge_on && c->colbits <= 21 ? 1024u : 0u};

    if (c->dbg.p && c->dbg_g == 2) HIP_TRY(hipMemsetAsync(c->dbg.p, 0, 64 * 16 * 8, s));
    HIP_TRY(hipEventRecord(c->ev[EV_SYM_BEGIN], s));
    HIP_TRY(hipMemsetAsync(dc, 0, sizeof(Counters), s));
    const uint32_t g256 = (n + 255) / 256, gsc = (n + 256 * SC_ITEMS - 1) / (256 * SC_ITEMS);
    if (n) {
        const uint32_t gent = (uint32_t)std::min<uint64_t>((a->nnz + 255) / 256 + 1, 256u * 8 * 8);
        hipLaunchKernelGGL(k_entry_desc, dim3(gent), dim3(256), 0, s, a->ptr, a->idx, b->ptr, b->idx, c->r0, n,
                           c->eb0.as<uint64_t>(), c->elen.as<uint32_t>(), c->efl.as<uint2>());
        hipLaunchKernelGGL(k_row_stats2, dim3(std::min<uint32_t>(g256, 2048)), dim3(256), 0, s, a->ptr, c->elen.as<uint32_t>(),
                           c->efl.as<uint2>(), c->r0, n, c->row_nprod.as<uint32_t>(), c->row_nnzc.as<uint32_t>(),
                           c->row_bin.as<uint8_t>(), c->row_kmin.as<uint32_t>(), c->row_kmax.as<uint32_t>(), dc->sym_counts,
                           dc->totals, dc->sym_prod, c->flat_on ? 1 : 0);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(c->ev[EV_STATS], s));
    HIP_TRY(hipMemcpyAsync(c->h_counters, dc, sizeof(Counters), hipMemcpyDeviceToHost, s));
    if (n) {
        hipLaunchKernelGGL(k_bin_scatter2, dim3(gsc), dim3(256), 0, s, c->row_bin.as<uint8_t>(), n, dc->sym_counts,
                           dc->sym_cursor, c->sym_rows.as<uint32_t>());
        hipLaunchKernelGGL(k_cut_tile_sums<0>, dim3(ntiles), dim3(SCAN_BLOCK), 0, s, a->ptr, c->r0, n,
                           c->row_nprod.as<uint32_t>(), c->row_nnzc.as<uint32_t>(), c->row_bin.as<uint8_t>(), cut_sym,
                           c->tile_sums.as<uint64_t>(), c->tile_w.as<uint64_t>());
        hipLaunchKernelGGL(k_cut_scan_tiles, dim3(1), dim3(SCAN_BLOCK), 0, s, c->tile_sums.as<uint64_t>(),
                           c->tile_w.as<uint64_t>(), ntiles, cut_sym.cap, 0, &dc->nb_sym);
        hipLaunchKernelGGL(k_cut_apply<0>, dim3(ntiles), dim3(SCAN_BLOCK), 0, s, a->ptr, c->r0, n,
                           c->row_nprod.as<uint32_t>(), c->row_nnzc.as<uint32_t>(), c->row_bin.as<uint8_t>(), cut_sym,
                           c->tile_sums.as<uint64_t>(), c->tile_w.as<uint64_t>(), ntiles, (uint64_t *)nullptr,
                           (uint8_t *)nullptr, (uint32_t *)nullptr, (unsigned long long *)nullptr,
                           c->batch_sym.as<uint32_t>());
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(c->ev[EV_BINNED], s));
    HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(c->h_sym_counts, c->h_counters->sym_counts, sizeof c->h_sym_counts);
    std::memcpy(c->h_sym_prod, c->h_counters->sym_prod, sizeof c->h_sym_prod);
    const uint64_t nprod = c->h_counters->totals[0], a_nnz = c->h_counters->totals[1];

    {
        const uint32_t *cnt = c->h_sym_counts;
        uint32_t off[SPADA_N_BINS + 1];
        off[0] = 0;
        for (int k = 0; k < SPADA_N_BINS; ++k) off[k + 1] = off[k] + (k == BIN_EMPTY || k == BIN_FLAT ? 0u : cnt[k]);
        if (cnt[SYM2_BIN_SPILL] && !c->bm_fits && (rc = ensure_spill(c, cnt[SYM2_BIN_SPILL], false))) return rc;
        HIP_TRY(hipEventRecord(c->ev_fork, s));
        if (cnt[SYM2_BIN_SPILL]) {
            const uint32_t nsp = cnt[SYM2_BIN_SPILL];
            if (c->bm_fits) {
                if ((rc = fork_to(c, SYM2_BIN_SPILL))) return rc;
                const size_t lds = bm_lds_bytes(b->cols, 0);
                hipLaunchKernelGGL(k_sym_bitmap, dim3(bm_grid(nsp, lds)), dim3(BM_BLOCK), lds, c->cur, c->a_view(), b->view(),
                                   c->r0, c->sym_rows.as<uint32_t>() + off[SYM2_BIN_SPILL], nsp, b->cols,
                                   c->row_nnzc.as<uint32_t>());
            } else {
                if ((rc = fork_to(c, SYM2_BIN_SPILL))) return rc;
                const uint64_t words = ((b->cols + 31) / 32 + 3) & ~3ull;
                const uint32_t grid = (uint32_t)std::min<uint64_t>(nsp, c->spill_slabs);
                hipLaunchKernelGGL(k_sym_spill, dim3(grid), dim3(SPILL_BLOCK), 0, c->cur, c->a_view(), b->view(), c->r0,
                                   c->sym_rows.as<uint32_t>() + off[SYM2_BIN_SPILL], nsp, c->bitmaps.as<uint32_t>(), words,
                                   c->row_nnzc.as<uint32_t>());
            }
            HIP_TRY(hipGetLastError());
            if ((rc = join_from(c, SYM2_BIN_SPILL))) return rc;
        }
#define SYM_BIN(BIN, G, LT)                                                   \
    if (cnt[BIN]) {                                                           \
        if ((rc = fork_to(c, BIN))) return rc;                                \
        if ((rc = launch_sym<G, LT>(c, off[BIN], cnt[BIN]))) return rc;       \
        if ((rc = join_from(c, BIN))) return rc;                              \
    }
        SYM_BIN(SYM2_BIN_24K, 1024, 15)
        SYM_BIN(SYM2_BIN_8K, 512, 14)
#undef SYM_BIN
        if (cnt[BIN_FLAT]) {
            if ((rc = fork_to(c, BIN_FLAT))) return rc;
            const uint64_t nb_upper = (nprod + (uint64_t)n * cut_sym.minw) / cut_sym.cap + 1;
#define LAUNCH_SYM_FLAT(BL, EP, LS, RP)                                                                                             \
    {                                                                                                                        \
        constexpr size_t lds = sym_flat_lds<BL, EP, SYM_FLAT_LOG_T, SF_RMAX>();                                              \
        hipLaunchKernelGGL((k_sym_flat<BL, EP, SYM_FLAT_LOG_T, SF_RMAX, LS, RP>), dim3(flat_grid(sf_batches, lds)), dim3(BL), lds,    \
                           c->cur, a->ptr, b->idx, c->eb0.as<uint64_t>(), c->elen.as<uint32_t>(), c->r0, n,                   \
                           c->row_bin.as<uint8_t>(), c->batch_sym.as<uint32_t>(), sf_nb, c->colbits,                         \
                           c->row_nnzc.as<uint32_t>(), sf_list, sf_bin, sf_dbg);                                             \
    }
#ifndef SPADA_SF_LARGE   /* default: 256-thread workgroups, 4096-key tables (A/B: -17..21 % symbolic time) */
#define SYM_FLAT_DISPATCH(LS, RP) LAUNCH_SYM_FLAT(256, 2, LS, RP)
#else
#define SYM_FLAT_DISPATCH(LS, RP)                          \
    if (c->flat_cfg == 0) LAUNCH_SYM_FLAT(256, 4, LS, RP)  \
    else if (c->flat_cfg == 1) LAUNCH_SYM_FLAT(512, 2, LS, RP) \
    else LAUNCH_SYM_FLAT(1024, 1, LS, RP)
#endif
            HIP_TRY(hipEventRecord(c->ev[EV_SFLAT_0], c->cur));
            {
                const uint64_t sf_batches = nb_upper;
                const uint32_t *sf_nb = &dc->nb_sym, *sf_list = nullptr;
                const uint32_t sf_bin = BIN_FLAT;
                unsigned long long *sf_dbg = c->dbg_g == 2 ? c->dbg.as<unsigned long long>() : nullptr;
                SYM_FLAT_DISPATCH(false, 1)
            }
            HIP_TRY(hipEventRecord(c->ev[EV_SFLAT_1], c->cur));
            HIP_TRY(hipGetLastError());
            if ((rc = join_from(c, BIN_FLAT))) return rc;
        }
        for (int pass = 0; pass < 2; ++pass) {   // mid rows from their row lists: one row per batch, or two of the lower half
            const int bin = pass == 0 ? SYM2_BIN_MID : SYM2_BIN_MID2;
            if (!cnt[bin]) continue;
            if ((rc = fork_to(c, bin))) return rc;
            const uint64_t sf_batches = (cnt[bin] + pass) / (pass + 1);
            const uint32_t *sf_nb = &dc->sym_counts[bin], *sf_list = c->sym_rows.as<uint32_t>() + off[bin];
            const uint32_t sf_bin = (uint32_t)bin;
            unsigned long long *sf_dbg = nullptr;
            if (pass == 0) { SYM_FLAT_DISPATCH(true, 1) } else { SYM_FLAT_DISPATCH(true, 2) }
            HIP_TRY(hipGetLastError());
            if ((rc = join_from(c, bin))) return rc;
        }
#undef LAUNCH_SYM_FLAT
    }
    HIP_TRY(hipEventRecord(c->ev[EV_SYM], s));

    // nnz(C_i) -> cptr, numeric classification, numeric batch cut, per-row bin lists
    hipLaunchKernelGGL(k_cut_tile_sums<1>, dim3(ntiles), dim3(SCAN_BLOCK), 0, s, a->ptr, c->r0, n, c->row_nprod.as<uint32_t>(),
                       c->row_nnzc.as<uint32_t>(), c->row_bin.as<uint8_t>(), cut_num, c->tile_sums.as<uint64_t>(),
                       c->tile_w.as<uint64_t>());
    hipLaunchKernelGGL(k_cut_scan_tiles, dim3(1), dim3(SCAN_BLOCK), 0, s, c->tile_sums.as<uint64_t>(), c->tile_w.as<uint64_t>(),
                       ntiles, cut_num.cap, 1, &dc->nb_num);
    hipLaunchKernelGGL(k_cut_apply<1>, dim3(ntiles), dim3(SCAN_BLOCK), 0, s, a->ptr, c->r0, n, c->row_nprod.as<uint32_t>(),
                       c->row_nnzc.as<uint32_t>(), c->row_bin.as<uint8_t>(), cut_num, c->tile_sums.as<uint64_t>(),
                       c->tile_w.as<uint64_t>(), ntiles, c->cptr.as<uint64_t>(), c->row_bin.as<uint8_t>(), dc->num_counts,
                       dc->num_sums, c->batch_num.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    if (n) {
        hipLaunchKernelGGL(k_bin_scatter2, dim3(gsc), dim3(256), 0, s, c->row_bin.as<uint8_t>(), n, dc->num_counts,
                           dc->num_cursor, c->num_rows.as<uint32_t>());
        // largest-first order of the huge rows (k_sort_rows_desc) measured -85 us on the serialised bitmap kernel but
        // nothing on the concurrent pipeline (other bins fill the idle CUs) while costing 16 us here: opt-in
        if (c->sort_huge)
            for (int hb : {NUM2_BIN_SPILL, NUM2_BIN_BMV})
                hipLaunchKernelGGL(k_sort_rows_desc, dim3(1), dim3(1024), 0, s, c->num_rows.as<uint32_t>(), dc->num_counts,
                                   hb, c->row_nprod.as<uint32_t>());
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(c->h_counters, dc, sizeof(Counters), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(c->h_u64, c->cptr.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(c->ev[EV_SCAN], s));
    HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(c->h_num_counts, c->h_counters->num_counts, sizeof c->h_num_counts);
    c->nnz_c = c->h_u64[0];
    c->h_nb_sym = std::max<uint32_t>(c->h_counters->nb_sym, 1);
    *nnz_c = c->nnz_c;
    c->have_symbolic = true;

    spada_stats &st = c->stats;
    st.rows = n;
    st.a_nnz = a_nnz;
    st.b_nnz = b->nnz;
    st.nprod = nprod;
    st.c_nnz = c->nnz_c;
    st.bytes_read = ((uint64_t)n + 1) * 8 + a_nnz * 12 + a_nnz * 16 + nprod * 12;
    st.bytes_write = ((uint64_t)n + 1) * 8 + c->nnz_c * 12;
    st.ms_row_stats = ev_ms(c, EV_SYM_BEGIN, EV_STATS);
    st.ms_binning = ev_ms(c, EV_STATS, EV_BINNED);
    st.ms_symbolic = ev_ms(c, EV_BINNED, EV_SYM);
    st.ms_scan = ev_ms(c, EV_SYM, EV_SCAN);
    st.ms_symbolic_call = ev_ms(c, EV_SYM_BEGIN, EV_SCAN);
    for (int k = 0; k < SPADA_N_BINS; ++k) {
        st.sym_bin_rows[k] = c->h_sym_counts[k];
        st.num_bin_rows[k] = c->h_num_counts[k];
    }
    for (int k = 0; k < SPADA_N_BINS; ++k) {
        st.num_bin_prod[k] = c->h_counters->num_sums[k];
        st.num_bin_nnz[k] = c->h_counters->num_sums[SPADA_N_BINS + k];
        st.num_bin_entries[k] = c->h_counters->num_sums[2 * SPADA_N_BINS + k];
        st.sym_bin_prod[k] = c->h_sym_prod[k];
    }
    st.ms_sym_flat = c->h_sym_counts[BIN_FLAT] ? ev_ms(c, EV_SFLAT_0, EV_SFLAT_1) : 0.0;
    st.spill_rows = c->h_sym_counts[SYM2_BIN_SPILL] + c->h_num_counts[NUM2_BIN_SPILL] + c->h_num_counts[NUM2_BIN_BMV];
    st.workspace_bytes = c->ws_bytes;
    return SPADA_OK;
}

int spada_dev_spgemm_numeric(spada_ctx *c, void *d_c_indptr, void *d_c_indices, void *d_c_data)
{
    if (!c) return fail(SPADA_ERR_STATE, "spada_dev_spgemm_numeric: no engine context (no GPU?)");
    if (!c->have_symbolic) return fail(SPADA_ERR_STATE, "numeric phase called without a preceding symbolic phase");
    if (!d_c_indptr || (c->nnz_c && (!d_c_indices || !d_c_data)))
        return fail(SPADA_ERR_INVALID, "spada_dev_spgemm_numeric: null output pointer");
    HIP_TRY(hipSetDevice(c->device));
    if (c->use_tasks) return task_numeric(c, (uint64_t *)d_c_indptr, (uint32_t *)d_c_indices, (double *)d_c_data);
    return run_numeric(c, (uint64_t *)d_c_indptr, (uint32_t *)d_c_indices, (double *)d_c_data);
}

int spada_dev_spgemm_fused(spada_ctx *c, const spada_dev_csr *a, const spada_dev_csr *b, uint64_t row_begin, uint64_t row_end,
                           void *d_c_indptr, void *d_c_indices, void *d_c_data, uint64_t capacity, uint64_t *nnz_c)
{
    if (!c) return fail(SPADA_ERR_STATE, "spada_dev_spgemm_fused: no engine context (no GPU?)");
    if (!a || !b || !nnz_c || !d_c_indptr || (capacity && (!d_c_indices || !d_c_data)))
        return fail(SPADA_ERR_INVALID, "spada_dev_spgemm_fused: null argument");
    if (a->cols != b->rows)
        return fail(SPADA_ERR_INVALID, "inner dimensions differ: A is %llux%llu, B is %llux%llu", (unsigned long long)a->rows,
                    (unsigned long long)a->cols, (unsigned long long)b->rows, (unsigned long long)b->cols);
    if (row_begin > row_end || row_end > a->rows) return fail(SPADA_ERR_INVALID, "bad row range");
    if (!c->use_tasks) return fail(SPADA_ERR_UNSUPPORTED, "the one-pass entry point needs the task pipeline (LDS-hash accumulator)");
    HIP_TRY(hipSetDevice(c->device));
    c->have_symbolic = false;
    c->cur = c->stream;
    c->A = a;
    c->B = b;
    c->r0 = row_begin;
    c->nrows = (uint32_t)(row_end - row_begin);
    c->nnz_c = 0;
    std::memset(&c->stats, 0, sizeof c->stats);
    int rc = task_pipeline(c, MODE_FUSED, (uint64_t *)d_c_indptr, (uint32_t *)d_c_indices, (double *)d_c_data, capacity);
    if (rc) return rc;
    *nnz_c = c->nnz_c;
    if (c->h_tctr->cap_overflow) {
        // C.indptr is complete: keep it, so that a numeric call into large enough buffers can follow
        HIP_TRY(hipMemcpyAsync(c->cptr.p, d_c_indptr, ((size_t)c->nrows + 1) * 8, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->have_symbolic = true;    // task list, scratch and range offsets are those of a finished symbolic phase
        return fail(SPADA_ERR_CAPACITY, "nnz(C) = %llu exceeds the capacity of %llu entries", (unsigned long long)c->nnz_c,
                    (unsigned long long)capacity);
    }
    return SPADA_OK;
}

int spada_dev_spgemm_fused_owned(spada_ctx *c, const spada_dev_csr *a, const spada_dev_csr *b, uint64_t row_begin,
                                 uint64_t row_end, uint64_t capacity, void **d_c_indptr, void **d_c_indices, void **d_c_data,
                                 uint64_t *nnz_c)
{
    if (!c) return fail(SPADA_ERR_STATE, "spada_dev_spgemm_fused_owned: no engine context (no GPU?)");
    if (!a || !d_c_indptr || !d_c_indices || !d_c_data || !nnz_c) return fail(SPADA_ERR_INVALID, "null argument");
    if (row_begin > row_end || row_end > a->rows) return fail(SPADA_ERR_INVALID, "bad row range");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->own_ptr.ensure((size_t)(row_end - row_begin + 1) * 8, false, c->stream, &c->ws_bytes))) return rc;
    if ((rc = c->own_idx.ensure(std::max<uint64_t>(capacity, 1) * 4, false, c->stream, &c->ws_bytes))) return rc;
    if ((rc = c->own_val.ensure(std::max<uint64_t>(capacity, 1) * 8, false, c->stream, &c->ws_bytes))) return rc;
    *d_c_indptr = c->own_ptr.p;
    *d_c_indices = c->own_idx.p;
    *d_c_data = c->own_val.p;
    return spada_dev_spgemm_fused(c, a, b, row_begin, row_end, c->own_ptr.p, c->own_idx.p, c->own_val.p, capacity, nnz_c);
}

int spada_dev_spgemm_numeric_owned(spada_ctx *c, void **d_c_indptr, void **d_c_indices, void **d_c_data)
{
    if (!c) return fail(SPADA_ERR_STATE, "spada_dev_spgemm_numeric_owned: no engine context (no GPU?)");
    if (!c->have_symbolic) return fail(SPADA_ERR_STATE, "numeric phase called without a preceding symbolic phase");
    if (!d_c_indptr || !d_c_indices || !d_c_data) return fail(SPADA_ERR_INVALID, "null output pointer");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->own_ptr.ensure(((size_t)c->nrows + 1) * 8, false, c->stream, &c->ws_bytes))) return rc;
    if ((rc = c->own_idx.ensure(std::max<uint64_t>(c->nnz_c, 1) * 4, false, c->stream, &c->ws_bytes))) return rc;
    if ((rc = c->own_val.ensure(std::max<uint64_t>(c->nnz_c, 1) * 8, false, c->stream, &c->ws_bytes))) return rc;
    if (c->use_tasks) rc = task_numeric(c, c->own_ptr.as<uint64_t>(), c->own_idx.as<uint32_t>(), c->own_val.as<double>());
    else rc = run_numeric(c, c->own_ptr.as<uint64_t>(), c->own_idx.as<uint32_t>(), c->own_val.as<double>());
    if (rc) return rc;
    *d_c_indptr = c->own_ptr.p;
    *d_c_indices = c->own_idx.p;
    *d_c_data = c->own_val.p;
    return SPADA_OK;
}

int spada_dev_download_c(spada_ctx *c, const void *d_c_indptr, const void *d_c_indices, const void *d_c_data, uint64_t rows,
                         uint64_t nnz_c, uint64_t *c_indptr, uint64_t *c_indices, double *c_data)
{
    if (!c) return fail(SPADA_ERR_STATE, "spada_dev_download_c: no engine context (no GPU?)");
    if (!d_c_indptr || !c_indptr || (nnz_c && (!d_c_indices || !d_c_data || !c_indices || !c_data)))
        return fail(SPADA_ERR_INVALID, "spada_dev_download_c: null pointer");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c_indptr, d_c_indptr, (rows + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    if (nnz_c) {
        int rc = c->wide_idx.ensure(nnz_c * 8, false, c->stream, &c->ws_bytes);
        if (rc) return rc;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((nnz_c + 255) / 256, 4096);
        hipLaunchKernelGGL(k_widen_u32, dim3(grid), dim3(256), 0, c->stream, (const uint32_t *)d_c_indices, nnz_c,
                           c->wide_idx.as<uint64_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(c_indices, c->wide_idx.p, nnz_c * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(c_data, d_c_data, nnz_c * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPADA_OK;
}

int spada_spgemm_symbolic(spada_ctx *c, const spada_csr_view *a, const spada_csr_view *b, uint64_t *nnz_c)
{
    if (!c) return fail(SPADA_ERR_STATE, "spada_spgemm_symbolic: no engine context (no GPU?)");
    if (!a || !b || !nnz_c) return fail(SPADA_ERR_INVALID, "spada_spgemm_symbolic: null argument");
    HIP_TRY(hipSetDevice(c->device));
    c->have_symbolic = false;
    dev_free(c->hA);
    if (c->hB != c->hA) dev_free(c->hB);
    c->hA = c->hB = nullptr;
    int rc = spada_dev_csr_upload(c, a, &c->hA);
    if (rc) return rc;
    const bool same = a->indptr == b->indptr && a->indices == b->indices && a->data == b->data && a->rows == b->rows &&
                      a->cols == b->cols;
    if (same) c->hB = c->hA;
    else if ((rc = spada_dev_csr_upload(c, b, &c->hB))) return rc;
    return spada_dev_spgemm_symbolic(c, c->hA, c->hB, 0, a->rows, nnz_c);
}

int spada_spgemm_fused(spada_ctx *c, const spada_csr_view *a, const spada_csr_view *b, uint64_t capacity, uint64_t *c_indptr,
                       uint64_t *c_indices, double *c_data, uint64_t *nnz_c)
{
    if (!c) return fail(SPADA_ERR_STATE, "spada_spgemm_fused: no engine context (no GPU?)");
    if (!a || !b || !nnz_c || !c_indptr) return fail(SPADA_ERR_INVALID, "spada_spgemm_fused: null argument");
    HIP_TRY(hipSetDevice(c->device));
    c->have_symbolic = false;
    dev_free(c->hA);
    if (c->hB != c->hA) dev_free(c->hB);
    c->hA = c->hB = nullptr;
    int rc = spada_dev_csr_upload(c, a, &c->hA);
    if (rc) return rc;
    const bool same = a->indptr == b->indptr && a->indices == b->indices && a->data == b->data && a->rows == b->rows &&
                      a->cols == b->cols;
    if (same) c->hB = c->hA;
    else if ((rc = spada_dev_csr_upload(c, b, &c->hB))) return rc;
    void *dp, *di, *dv;
    rc = spada_dev_spgemm_fused_owned(c, c->hA, c->hB, 0, a->rows, capacity, &dp, &di, &dv, nnz_c);
    if (rc == SPADA_ERR_CAPACITY) {
        const std::string msg = spada_last_error();
        const int rc2 = spada_dev_download_c(c, dp, nullptr, nullptr, c->nrows, 0, c_indptr, nullptr, nullptr);
        if (rc2) return rc2;
        return fail(SPADA_ERR_CAPACITY, "%s", msg.c_str());
    }
    if (rc) return rc;
    return spada_dev_download_c(c, dp, di, dv, c->nrows, c->nnz_c, c_indptr, c_indices, c_data);
}

int spada_spgemm_numeric(spada_ctx *c, uint64_t *c_indptr, uint64_t *c_indices, double *c_data)
{
    if (!c) return fail(SPADA_ERR_STATE, "spada_spgemm_numeric: no engine context (no GPU?)");
    if (!c->have_symbolic || !c->hA) return fail(SPADA_ERR_STATE, "numeric phase called without a preceding symbolic phase");
    void *dp, *di, *dv;
    int rc = spada_dev_spgemm_numeric_owned(c, &dp, &di, &dv);
    if (rc) return rc;
    return spada_dev_download_c(c, dp, di, dv, c->nrows, c->nnz_c, c_indptr, c_indices, c_data);
}

// development aid (scripts/phase_timing.py): s_memtime stamps written by the kernels when SPADA_DBG_G is set
int spada_debug_read(spada_ctx *c, unsigned long long *out, uint64_t n)
{
    if (!c || !c->dbg.p) return fail(SPADA_ERR_STATE, "no debug buffer");
    HIP_TRY(hipMemcpy(out, c->dbg.p, std::min<uint64_t>(n, 64 * 16) * 8, hipMemcpyDeviceToHost));
    return SPADA_OK;
}

int spada_get_stats(const spada_ctx *c, spada_stats *out)
{
    if (!c || !out) return fail(SPADA_ERR_INVALID, "spada_get_stats: null argument");
    *out = c->stats;
    return SPADA_OK;
}

}  // extern "C"
