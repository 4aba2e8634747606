/*
 * oracle/spgemm_ref.c -- CPU restatement of spada-sim's multiply/merge arithmetic.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (libspada_spgemm.so)
 * never links, loads or calls anything in this file.
 *
 * PARITY STATUS
 *   - loader half (MatrixMarket -> CSR): pinned.  tests/golden/cari_loader.npz was produced
 *     by executing the reference's own embedded Python loader (src/py2rust.rs:64-79) in the
 *     build container (tests/golden/make_golden.py).
 *   - arithmetic half (this file): PARITY UNPINNED against the reference binary.  The
 *     reference is a Rust (nightly-2021-12-04) cycle simulator with 86 unvendored crates and
 *     cannot be compiled here (no cargo/rustc, no network); it ships no tests, golden vectors
 *     or known-answer outputs.  This restatement follows the reference source line by line
 *     (citations below) and is cross-checked against scipy (A @ B, pattern from boolean
 *     product) in tests/golden/make_golden.py; that is a cross-check, not a reference pin.
 *
 * What the reference computes (all citations into /root/reference/src):
 *   - product:        c.value = a.value * b.value, c.idx = [a.row, b.col]   simulator.rs:86-111
 *                                                                         adder_tree.rs:37-57
 *   - ordering:       per C row, products are sorted by column; the sort is stable
 *                     (slice::sort_by)                                      simulator.rs:143-171
 *   - accumulation:   runs of equal column are summed left to right with f64 `+=`; the
 *                     first element of a run is kept as is; nothing drops a 0.0 sum
 *                                                                         simulator.rs:199-230
 *                                                                         adder_tree.rs:73-83
 *   - merge of partial fibers: ascending unique columns, ties -> left    adder_tree.rs:145-188
 *   - result assembly: one fiber per A row, rows ascending; A rows without nonzeros, or
 *                     whose referenced B rows are all empty, give an empty fiber
 *                                                                         simulator.rs:1034-1062
 *   - compact statement of the same row-wise expand-scale-merge
 *                                                   storage_traffic_model.rs:1668-1697
 *   - operand choice: square -> B = A, otherwise B = A^T as sorted CSR     gemm.rs:41-53
 *   - layout:         data f64, indptr/indices usize (= uint64_t)          storage.rs:150-160
 *
 * Summation order: the reference adds the products of one output entry in an order set by
 * its K-windows and merge schedule (scheduler.rs:482-606, :381-480); this file adds them in
 * ascending k (A-nonzero order), which is what a stable sort by column followed by a
 * left-to-right run sum gives when the whole row is one group.  f64 addition order changes
 * results at the 1e-16 level; parity tolerance for values is 1e-9 relative (BASELINE.json).
 * oracle_spgemm_windowed (end of file) adds in the reference's window / pairwise-merge order instead; the
 * two differ by at most 3.6e-15 relative on cari and 6.6e-16 of sum|a b| on signed random cases
 * (tests/test_oracle_golden.py::test_order_faithful_variant_bounds_the_summation_order_effect).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    uint64_t col;
    double val;
} prod_t;

/* stable merge sort by column (simulator.rs:160 uses a stable sort) */
static void msort(prod_t *a, prod_t *tmp, size_t n)
{
    if (n < 2) return;
    if (n <= 16) { /* insertion sort is stable */
        for (size_t i = 1; i < n; ++i) {
            prod_t x = a[i];
            size_t j = i;
            while (j > 0 && a[j - 1].col > x.col) { a[j] = a[j - 1]; --j; }
            a[j] = x;
        }
        return;
    }
    size_t h = n / 2;
    msort(a, tmp, h);
    msort(a + h, tmp, n - h);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (a[j].col < a[i].col) ? a[j++] : a[i++]; /* ties -> left */
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, n * sizeof(prod_t));
}

/* number of products of A row i: sum over its nonzeros of nnz(B row k) */
static uint64_t row_nprod(uint64_t i, const uint64_t *a_indptr, const uint64_t *a_indices,
                          const uint64_t *b_indptr)
{
    uint64_t n = 0;
    for (uint64_t p = a_indptr[i]; p < a_indptr[i + 1]; ++p) {
        uint64_t k = a_indices[p];
        n += b_indptr[k + 1] - b_indptr[k];
    }
    return n;
}

uint64_t oracle_count_products(uint64_t a_rows, const uint64_t *a_indptr,
                               const uint64_t *a_indices, const uint64_t *b_indptr)
{
    uint64_t n = 0;
    for (uint64_t i = 0; i < a_rows; ++i) n += row_nprod(i, a_indptr, a_indices, b_indptr);
    return n;
}

/*
 * Sort-merge restatement, one C row at a time (single thread).
 * Phase 1 (c_indices == NULL): fills c_indptr[0..a_rows] and returns nnz(C).
 * Phase 2: c_indptr as produced by phase 1; fills c_indices / c_data.
 * Returns nnz(C), or UINT64_MAX on allocation failure.
 */
uint64_t oracle_spgemm_sortmerge(uint64_t a_rows,
                                 const uint64_t *a_indptr, const uint64_t *a_indices, const double *a_data,
                                 const uint64_t *b_indptr, const uint64_t *b_indices, const double *b_data,
                                 uint64_t *c_indptr, uint64_t *c_indices, double *c_data)
{
    uint64_t maxp = 0;
    for (uint64_t i = 0; i < a_rows; ++i) {
        uint64_t n = row_nprod(i, a_indptr, a_indices, b_indptr);
        if (n > maxp) maxp = n;
    }
    prod_t *buf = (prod_t *)malloc((maxp ? maxp : 1) * sizeof(prod_t));
    prod_t *tmp = (prod_t *)malloc((maxp ? maxp : 1) * sizeof(prod_t));
    if (!buf || !tmp) { free(buf); free(tmp); return UINT64_MAX; }

    const int fill = (c_indices != NULL);
    uint64_t nnz = 0;
    if (!fill) c_indptr[0] = 0;
    for (uint64_t i = 0; i < a_rows; ++i) {
        /* expand + scale (simulator.rs:100-101) */
        size_t n = 0;
        for (uint64_t p = a_indptr[i]; p < a_indptr[i + 1]; ++p) {
            uint64_t k = a_indices[p];
            double av = a_data[p];
            for (uint64_t q = b_indptr[k]; q < b_indptr[k + 1]; ++q) {
                buf[n].col = b_indices[q];
                buf[n].val = av * b_data[q];
                ++n;
            }
        }
        /* sort by column, stable (simulator.rs:160) */
        msort(buf, tmp, n);
        /* sum runs of equal column left to right, keep zeros (simulator.rs:209-220) */
        uint64_t out = fill ? c_indptr[i] : 0;
        uint64_t cnt = 0;
        size_t j = 0;
        while (j < n) {
            uint64_t col = buf[j].col;
            double acc = buf[j].val;
            ++j;
            while (j < n && buf[j].col == col) { acc += buf[j].val; ++j; }
            if (fill) { c_indices[out + cnt] = col; c_data[out + cnt] = acc; }
            ++cnt;
        }
        nnz += cnt;
        if (!fill) c_indptr[i + 1] = nnz;
    }
    free(buf);
    free(tmp);
    return nnz;
}

/*
 * Same result, bit for bit, with a dense sparse-accumulator (SPA) per thread instead of a
 * sort: first touch of a column assigns, later touches `+=` in ascending-k order, touched
 * columns are sorted before emission.  Rows are distributed over OpenMP threads.  This is
 * the variant bench.py times as `cpu_baseline` (kind "port").
 * Two-phase like oracle_spgemm_sortmerge.  n_threads <= 0 -> omp default.
 */
static int cmp_u64(const void *x, const void *y)
{
    uint64_t a = *(const uint64_t *)x, b = *(const uint64_t *)y;
    return (a > b) - (a < b);
}

uint64_t oracle_spgemm_spa(uint64_t a_rows, uint64_t b_cols,
                           const uint64_t *a_indptr, const uint64_t *a_indices, const double *a_data,
                           const uint64_t *b_indptr, const uint64_t *b_indices, const double *b_data,
                           uint64_t *c_indptr, uint64_t *c_indices, double *c_data, int n_threads)
{
    const int fill = (c_indices != NULL);
    int failed = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
    if (!fill) c_indptr[0] = 0;
#pragma omp parallel
    {
        double *acc = (double *)malloc((b_cols ? b_cols : 1) * sizeof(double));
        uint8_t *flag = (uint8_t *)calloc(b_cols ? b_cols : 1, 1);
        size_t cap = 1024;
        uint64_t *touched = (uint64_t *)malloc(cap * sizeof(uint64_t));
        if (!acc || !flag || !touched) {
#pragma omp atomic write
            failed = 1;
        } else {
#pragma omp for schedule(dynamic, 256)
            for (uint64_t i = 0; i < a_rows; ++i) {
                size_t nt = 0;
                for (uint64_t p = a_indptr[i]; p < a_indptr[i + 1]; ++p) {
                    uint64_t k = a_indices[p];
                    double av = a_data[p];
                    for (uint64_t q = b_indptr[k]; q < b_indptr[k + 1]; ++q) {
                        uint64_t col = b_indices[q];
                        double v = av * b_data[q];
                        if (!flag[col]) {
                            flag[col] = 1;
                            acc[col] = v;
                            if (nt == cap) {
                                cap *= 2;
                                touched = (uint64_t *)realloc(touched, cap * sizeof(uint64_t));
                            }
                            touched[nt++] = col;
                        } else {
                            acc[col] += v;
                        }
                    }
                }
                if (!fill) {
                    c_indptr[i + 1] = nt; /* per-row count; prefix-summed below */
                    for (size_t t = 0; t < nt; ++t) flag[touched[t]] = 0;
                } else {
                    qsort(touched, nt, sizeof(uint64_t), cmp_u64);
                    uint64_t out = c_indptr[i];
                    for (size_t t = 0; t < nt; ++t) {
                        uint64_t col = touched[t];
                        c_indices[out + t] = col;
                        c_data[out + t] = acc[col];
                        flag[col] = 0;
                    }
                }
            }
        }
        free(acc);
        free(flag);
        free(touched);
    }
    if (failed) return UINT64_MAX;
    if (!fill) {
        for (uint64_t i = 0; i < a_rows; ++i) c_indptr[i + 1] += c_indptr[i];
    }
    return c_indptr[a_rows];
}

/*
 * B = A^T as CSR with ascending column indices inside each row -- what
 * `mat.clone().transpose_into().to_csr()` yields (gemm.rs:46).  Counting sort over columns;
 * walking A in row order makes every output row ascending.
 * t_indptr has a_cols+1 entries; t_indices / t_data have nnz entries.
 */
void oracle_transpose_csr(uint64_t a_rows, uint64_t a_cols,
                          const uint64_t *indptr, const uint64_t *indices, const double *data,
                          uint64_t *t_indptr, uint64_t *t_indices, double *t_data)
{
    uint64_t nnz = indptr[a_rows];
    memset(t_indptr, 0, (a_cols + 1) * sizeof(uint64_t));
    for (uint64_t p = 0; p < nnz; ++p) t_indptr[indices[p] + 1]++;
    for (uint64_t c = 0; c < a_cols; ++c) t_indptr[c + 1] += t_indptr[c];
    uint64_t *cursor = (uint64_t *)malloc((a_cols ? a_cols : 1) * sizeof(uint64_t));
    memcpy(cursor, t_indptr, a_cols * sizeof(uint64_t));
    for (uint64_t i = 0; i < a_rows; ++i) {
        for (uint64_t p = indptr[i]; p < indptr[i + 1]; ++p) {
            uint64_t dst = cursor[indices[p]]++;
            t_indices[dst] = i;
            t_data[dst] = data[p];
        }
    }
    free(cursor);
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/*
 * Order-faithful variant: the SAME products, added in (a timing-independent restatement of) the reference's order instead of
 * ascending k.  The reference never forms a whole C row at once: with the shipped configuration (block height 1, lane_num 8)
 * Scheduler::next_window (scheduler.rs:482-606) hands the PE the A scalars of a row `lane_num` at a time; every window is
 * expanded, stably sorted by column and run-summed left to right (simulator.rs:143-171, :199-230) into a PARTIAL fiber of its
 * own (one psum address per window, scheduler.rs:548-550); the partial fibers of a row are then merged two at a time --
 * Scheduler::merge_task drains the two oldest (`psum_addrs.drain(..2)`, scheduler.rs:399-406), the PE multiplies both by 1.0
 * and merges them with equal columns added left + right (adder_tree.rs:73-83, :173-188: ties -> left), and the result joins
 * the back of the row's list -- until one fiber is left (simulator.rs:985-1006).  WHEN a merge is issued relative to later
 * windows depends on the simulated clock; this restatement issues all windows first and then merges first-in-first-out, which
 * fixes one of the orders the reference can produce.  Structure is identical to oracle_spgemm_sortmerge by construction;
 * values differ from it only by floating-point re-association (tests/test_oracle_golden.py records by how much).
 * Same two-phase calling convention.
 */
typedef struct {
    uint64_t *col;
    double *val;
    size_t n;
} fiber_t;

uint64_t oracle_spgemm_windowed(uint64_t a_rows, uint64_t lane_num,
                                const uint64_t *a_indptr, const uint64_t *a_indices, const double *a_data,
                                const uint64_t *b_indptr, const uint64_t *b_indices, const double *b_data,
                                uint64_t *c_indptr, uint64_t *c_indices, double *c_data)
{
    if (lane_num == 0) lane_num = 8;
    const int fill = (c_indices != NULL);
    uint64_t nnz = 0;
    if (!fill) c_indptr[0] = 0;
    for (uint64_t i = 0; i < a_rows; ++i) {
        const uint64_t a0 = a_indptr[i], a1 = a_indptr[i + 1];
        const size_t nwin = (size_t)((a1 - a0 + lane_num - 1) / lane_num);
        /* queue of partial fibers, oldest first; at most 2 * nwin entries are ever appended */
        fiber_t *q = (fiber_t *)calloc(2 * nwin + 1, sizeof(fiber_t));
        if (!q) return UINT64_MAX;
        size_t head = 0, tail = 0;
        for (size_t w = 0; w < nwin; ++w) {
            const uint64_t p0 = a0 + w * lane_num, p1 = (p0 + lane_num < a1) ? p0 + lane_num : a1;
            size_t n = 0;
            for (uint64_t p = p0; p < p1; ++p) n += (size_t)(b_indptr[a_indices[p] + 1] - b_indptr[a_indices[p]]);
            if (n == 0) continue;   /* no product, no psum address (simulator.rs:650) */
            prod_t *buf = (prod_t *)malloc(n * sizeof(prod_t)), *tmp = (prod_t *)malloc(n * sizeof(prod_t));
            fiber_t f;
            f.col = (uint64_t *)malloc(n * sizeof(uint64_t));
            f.val = (double *)malloc(n * sizeof(double));
            if (!buf || !tmp || !f.col || !f.val) return UINT64_MAX;
            size_t m = 0;
            for (uint64_t p = p0; p < p1; ++p) {
                const uint64_t k = a_indices[p];
                for (uint64_t t = b_indptr[k]; t < b_indptr[k + 1]; ++t) {
                    buf[m].col = b_indices[t];
                    buf[m].val = a_data[p] * b_data[t];
                    ++m;
                }
            }
            msort(buf, tmp, n);
            f.n = 0;
            for (size_t j = 0; j < n;) {
                const uint64_t col = buf[j].col;
                double acc = buf[j].val;
                ++j;
                while (j < n && buf[j].col == col) { acc += buf[j].val; ++j; }
                f.col[f.n] = col;
                f.val[f.n] = acc;
                ++f.n;
            }
            free(buf);
            free(tmp);
            q[tail++] = f;
        }
        while (tail - head > 1) {   /* merge the two oldest, append the result */
            fiber_t x = q[head], y = q[head + 1], z;
            head += 2;
            z.col = (uint64_t *)malloc((x.n + y.n) * sizeof(uint64_t));
            z.val = (double *)malloc((x.n + y.n) * sizeof(double));
            if (!z.col || !z.val) return UINT64_MAX;
            size_t ix = 0, iy = 0;
            z.n = 0;
            while (ix < x.n || iy < y.n) {
                if (iy >= y.n || (ix < x.n && x.col[ix] < y.col[iy])) { z.col[z.n] = x.col[ix]; z.val[z.n] = x.val[ix] * 1.0; ++ix; }
                else if (ix >= x.n || y.col[iy] < x.col[ix]) { z.col[z.n] = y.col[iy]; z.val[z.n] = y.val[iy] * 1.0; ++iy; }
                else { z.col[z.n] = x.col[ix]; z.val[z.n] = x.val[ix] * 1.0 + y.val[iy] * 1.0; ++ix; ++iy; }
                ++z.n;
            }
            free(x.col); free(x.val); free(y.col); free(y.val);
            q[tail++] = z;
        }
        uint64_t cnt = 0;
        if (tail > head) {
            cnt = q[head].n;
            if (fill) {
                memcpy(c_indices + c_indptr[i], q[head].col, cnt * sizeof(uint64_t));
                memcpy(c_data + c_indptr[i], q[head].val, cnt * sizeof(double));
            }
            free(q[head].col);
            free(q[head].val);
        }
        free(q);
        nnz += cnt;
        if (!fill) c_indptr[i + 1] = nnz;
    }
    return nnz;
}
