// gfx950 kernels, part 2: per-entry B-row descriptors, batch cutting, and the "flat batch" symbolic /
// numeric kernels that take every small and medium row of C.
//
// Why flat batches.  A per-row kernel spends its time in a chain of dependent memory hops (row id -> A row
// pointers -> A entries -> B row pointers -> B entries) while its LDS hash table, sized for the largest row of
// its bin, sits mostly empty; with 160 KiB of LDS per CU only a handful of rows are in flight.  Here consecutive
// rows of C are packed into batches of ~CAP outputs (numeric) / ~CAP products (symbolic): one workgroup owns
// one batch, all its rows share ONE open-addressing table keyed by (local row << colbits | column), every
// phase (gather, insert, bucket, scan, rank, store) is a long flat loop over the whole batch, the table is
// filled to 50-75 % whatever the individual row sizes are, the A entries of a batch are one contiguous stream
// and so is its slice of C.  Replaces, like part 1, the simulated multiply / sort / merge datapath
// (simulator.rs:86-230, scheduler.rs:381-606) -- only the result C is reproduced.
#pragma once
#include "spgemm_kernels.hip.hpp"

namespace spada {

// ---- bins of the v2 pipeline --------------------------------------------------------------------------------
// symbolic: 0 empty (P == 0) | 1 copy (one A nonzero: nnz = P) | 2 flat (P <= SYM_FLAT_MAX)
//           3 P <= 8192 (k_sym_hash<512,14>) | 4 P <= 24576 (k_sym_hash<1024,15>) | 5 bitmap / spill
// numeric : 0 empty | 1 copy | 2 flat (n <= num_flat_max) | 3 n <= 2048 and P <= 16384 (k_num_hash<256,12>)
//           4 n <= 6144 (k_num_hash<1024,13>) | 5 LDS bitmap with LDS values (n <= vcap) | 6 bitmap / spill
constexpr int BIN_EMPTY = 0, BIN_COPY = 1, BIN_FLAT = 2;
constexpr int SYM2_BIN_8K = 3, SYM2_BIN_24K = 4, SYM2_BIN_SPILL = 5, SYM2_BIN_MID = 6, SYM2_BIN_MID2 = 7;
constexpr int NUM2_BIN_2K = 3, NUM2_BIN_6K = 4, NUM2_BIN_BMV = 5, NUM2_BIN_SPILL = 6, NUM2_BIN_MID = 7;
constexpr int NUM2_BIN_MERGE_S = 8, NUM2_BIN_MERGE_L = 9;   // multiway merge, <= 512 / <= 1024 products
constexpr int NUM2_BIN_MID2 = 10;   // lower half of the mid class: two rows per list-mode batch
constexpr uint32_t MERGE_LMAX = 8, MERGE_PMIN = 96;
// "mid" rows: too large for a shared batch, small enough for the flat kernels' table -- one row per batch, taken
// from the bin's row list (list mode of k_sym_flat / k_num_flat)
#ifndef SPADA_SF_LARGE   /* default: 256-thread workgroups, 4096-key tables (A/B: -17..21 % symbolic time) */
constexpr uint32_t SYM_MID_MAX = 3072;
#else
constexpr uint32_t SYM_MID_MAX = 6144;   // products: 0.75 of the 8192-key symbolic table
#endif
#ifndef SPADA_NF_LARGE
constexpr uint32_t NUM_MID_MAX = 1536;
#else
constexpr uint32_t NUM_MID_MAX = 3072;   // outputs:  0.75 of the 4096-slot numeric table, = its bucket array
#endif

#ifndef SPADA_SF_LARGE   /* default: 256-thread workgroups, 4096-key tables (A/B: -17..21 % symbolic time) */
constexpr uint32_t SYM_FLAT_CAP = 2048, SYM_FLAT_MAX = 1024;
constexpr int SYM_FLAT_LOG_T = 12;
#else
constexpr uint32_t SYM_FLAT_CAP = 4096, SYM_FLAT_MAX = 2048;   // products per batch / per flat row
constexpr int SYM_FLAT_LOG_T = 13;                             // 8192 keys: load <= 0.75, typically 0.5
#endif

__host__ __device__ inline int sym2_bin_of(uint64_t P, uint32_t L)
{
    if (P == 0) return BIN_EMPTY;
    if (L == 1) return BIN_COPY;
    if (P <= SYM_FLAT_MAX) return BIN_FLAT;
    if (P <= SYM_MID_MAX / 2) return SYM2_BIN_MID2;   // any two of them fit one table
    if (P <= SYM_MID_MAX) return SYM2_BIN_MID;
    if (P <= 8192) return SYM2_BIN_8K;
    if (P <= 24576) return SYM2_BIN_24K;
    return SYM2_BIN_SPILL;
}
// sm_pmax > 0 selects the sort-merge accumulator: "flat" then means P <= sm_pmax (the rows of the symbolic batches)
// merge_pmax > 0 enables the multiway-merge class (rows with few, long B rows): L <= 32, MERGE_PMIN <= P <= merge_pmax
__host__ __device__ inline int num2_bin_of(uint32_t n, uint64_t P, uint32_t L, uint32_t flat_max, uint32_t vcap,
                                           uint32_t sm_pmax = 0, uint32_t merge_pmax = 0)
{
    if (n == 0) return BIN_EMPTY;
    if (L == 1) return BIN_COPY;
    if (merge_pmax && L <= MERGE_LMAX && P >= MERGE_PMIN && P <= merge_pmax) return P <= 512 ? NUM2_BIN_MERGE_S : NUM2_BIN_MERGE_L;
    if (sm_pmax ? P <= sm_pmax : (n <= flat_max && P <= (1u << 22))) return BIN_FLAT;
    if (flat_max && n <= NUM_MID_MAX / 2 && P <= (1u << 22)) return NUM2_BIN_MID2;   // any two of them fit one table
    if (flat_max && n <= NUM_MID_MAX && P <= (1u << 22)) return NUM2_BIN_MID;
    if (n <= 2048 && P <= 16384) return NUM2_BIN_2K;
    if (n <= 6144) return NUM2_BIN_6K;
    if (n <= vcap) return NUM2_BIN_BMV;
    return NUM2_BIN_SPILL;
}

// per-A-entry descriptor of the B row it selects, written once by k_row_stats2 and read by every walk
struct EntryDesc {
    const uint64_t *b0;    // first B entry of the row
    const uint32_t *len;   // its length
};

// ---- 1. entry descriptors + row statistics -------------------------------------------------------------------------
// k_entry_desc: one lane per A entry (coalesced A.indices; the irregular 16-byte B.indptr gather of the path, done
// exactly once, and the first / last column of the selected B row) -- pure throughput, no per-row chains.
// k_row_stats2: one lane per A row (rows longer than 16 nonzeros: the whole wave) over the now contiguous
// descriptors: products P_i, first / last possible column of C_i (they steer the order-preserving buckets of the
// numeric phase), symbolic bin, histogram.
__global__ __launch_bounds__(256) void k_entry_desc(const uint64_t *__restrict__ aptr, const uint32_t *__restrict__ aidx,
                                                    const uint64_t *__restrict__ bptr, const uint32_t *__restrict__ bidx,
                                                    uint64_t r0, uint32_t nrows, uint64_t *__restrict__ eb0,
                                                    uint32_t *__restrict__ elen, uint2 *__restrict__ efl)
{
    const uint64_t a_begin = aptr[r0], a_end = aptr[r0 + nrows];
    for (uint64_t q = a_begin + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < a_end; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t k = aidx[q];
        const uint64_t b0 = bptr[k], b1 = bptr[k + 1];
        eb0[q] = b0;
        elen[q] = (uint32_t)(b1 - b0);
        uint2 fl = make_uint2(0xFFFFFFFFu, 0u);
        if (b1 > b0) fl = make_uint2(bidx[b0], bidx[b1 - 1]);
        efl[q] = fl;
    }
}

__global__ __launch_bounds__(256) void k_row_stats2(const uint64_t *__restrict__ aptr, const uint32_t *__restrict__ elen,
                                                    const uint2 *__restrict__ efl, uint64_t r0, uint32_t nrows,
                                                    uint32_t *__restrict__ row_nprod, uint32_t *__restrict__ row_nnzc,
                                                    uint8_t *__restrict__ row_bin, uint32_t *__restrict__ row_kmin,
                                                    uint32_t *__restrict__ row_kmax, uint32_t *__restrict__ bin_counts,
                                                    unsigned long long *__restrict__ totals /* [0]=nprod [1]=a_nnz */,
                                                    unsigned long long *__restrict__ bin_prod, int flat_on)
{
    __shared__ uint32_t s_hist[SPADA_N_BINS];
    __shared__ unsigned long long s_tot[2], s_bp[SPADA_N_BINS];
    if (threadIdx.x < SPADA_N_BINS) s_bp[threadIdx.x] = 0;
    if (threadIdx.x < SPADA_N_BINS) s_hist[threadIdx.x] = 0;
    if (threadIdx.x < 2) s_tot[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    auto visit = [&](uint64_t q, uint64_t &part, uint32_t &mn, uint32_t &mx) {
        const uint2 fl = efl[q];
        part += elen[q];
        mn = min(mn, fl.x);
        mx = max(mx, fl.y);
    };
    uint64_t tot_p = 0, tot_l = 0;
    // grid-stride over row tiles: the per-bin histogram costs one global atomic per workgroup, not per tile
    // (a hot global word sustains ~90 atomics per microsecond)
    for (uint32_t tile = blockIdx.x * blockDim.x; tile < nrows; tile += gridDim.x * blockDim.x) {
        const uint32_t i = tile + threadIdx.x;
        uint64_t a0 = 0, a1 = 0, P = 0;
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
        if (i < nrows) {
            a0 = aptr[r0 + i];
            a1 = aptr[r0 + i + 1];
        }
        const uint32_t L = (uint32_t)(a1 - a0);
        const bool is_long = L > 16;
        if (!is_long)
            for (uint64_t q = a0; q < a1; q += 4) {   // four independent loads per round
                uint32_t l4[4];
                uint2 f4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint64_t qq = q + u < a1 ? q + u : a1 - 1;
                    l4[u] = elen[qq];
                    f4[u] = efl[qq];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (q + u < a1) {
                        P += l4[u];
                        kmin = min(kmin, f4[u].x);
                        kmax = max(kmax, f4[u].y);
                    }
            }
        unsigned long long mask = __ballot(is_long);
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const uint64_t sa0 = __shfl(a0, src), sa1 = __shfl(a1, src);
            uint64_t part = 0;
            uint32_t mn = 0xFFFFFFFFu, mx = 0;
            for (uint64_t q = sa0 + lane; q < sa1; q += 64) visit(q, part, mn, mx);
            part = wave_sum_u64(part);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mn = min(mn, (uint32_t)__shfl_xor(mn, o));
                mx = max(mx, (uint32_t)__shfl_xor(mx, o));
            }
            if (lane == src) {
                P = part;
                kmin = mn;
                kmax = mx;
            }
        }
        if (i < nrows) {
            int bin = sym2_bin_of(P, L);
            if ((bin == BIN_FLAT || bin == SYM2_BIN_MID || bin == SYM2_BIN_MID2) && !flat_on) bin = SYM2_BIN_8K;
            row_nprod[i] = P > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)P;
            row_bin[i] = (uint8_t)bin;
            row_kmin[i] = kmin;
            row_kmax[i] = kmax;
            if (bin == BIN_EMPTY) row_nnzc[i] = 0;
            if (bin == BIN_COPY) row_nnzc[i] = (uint32_t)P;   // one A nonzero: C row is a scaled copy of one B row
            atomicAdd(&s_hist[bin], 1u);
            if (bin > BIN_COPY) atomicAdd(&s_bp[bin], (unsigned long long)P);
        }
        tot_p += P;
        tot_l += L;
    }
    const uint64_t wp = wave_sum_u64(tot_p), wl = wave_sum_u64(tot_l);
    if (lane == 0) {
        atomicAdd(&s_tot[0], (unsigned long long)wp);
        atomicAdd(&s_tot[1], (unsigned long long)wl);
    }
    __syncthreads();
    if (threadIdx.x < SPADA_N_BINS && s_hist[threadIdx.x]) atomicAdd(&bin_counts[threadIdx.x], s_hist[threadIdx.x]);
    if (threadIdx.x < 2 && s_tot[threadIdx.x]) atomicAdd(&totals[threadIdx.x], s_tot[threadIdx.x]);
    if (threadIdx.x < SPADA_N_BINS && s_bp[threadIdx.x]) atomicAdd(&bin_prod[threadIdx.x], s_bp[threadIdx.x]);
}

// ---- 2. scans: nnz(C_i) -> cptr, numeric classification, and batch cutting ------------------------------------
// Batches: every row carries a weight w_i (flat rows: their size, at least `minw`; all other rows: `minw`, which
// bounds the rows of a batch by cap / minw).  With S_i the exclusive prefix sum of w, row i belongs to batch
// floor(S_i / cap); a batch therefore weighs less than cap + max flat weight.  Row i announces the first row
// of the next batch when its own interval [S_i, S_i + w_i) reaches the next multiple of cap.
struct CutParams {
    uint32_t cap, minw, flat_max /* numeric */, vcap /* numeric */, sm_pmax /* numeric, sort-merge accumulator */;
    uint32_t merge_pmax /* numeric: multiway-merge class, 0 = off */;
};

// MODE 0: symbolic cut (weights from row_nprod / row_bin)      MODE 1: numeric (nnzc -> cptr, classify, cut)
template <int MODE>
__device__ inline void scan_row_values(uint32_t i, uint32_t n, const uint64_t *aptr, uint64_t r0, const uint32_t *row_nprod,
                                       const uint32_t *row_nnzc, const uint8_t *row_bin_in, CutParams cp, uint32_t &nnz,
                                       uint32_t &w, int &bin)
{
    nnz = 0;
    w = 0;
    bin = 0;
    if (i >= n) return;
    if constexpr (MODE == 0) {
        bin = row_bin_in[i];
        const uint32_t P = row_nprod[i];
        w = bin == BIN_FLAT ? max(P, cp.minw) : cp.minw;
    } else {
        nnz = row_nnzc[i];
        const uint32_t L = (uint32_t)(aptr[r0 + i + 1] - aptr[r0 + i]);
        bin = num2_bin_of(nnz, row_nprod[i], L, cp.flat_max, cp.vcap, cp.sm_pmax, cp.merge_pmax);
        w = bin == BIN_FLAT ? max(nnz, cp.minw) : cp.minw;
    }
}

template <int MODE>
__global__ __launch_bounds__(SCAN_BLOCK) void k_cut_tile_sums(const uint64_t *__restrict__ aptr, uint64_t r0, uint32_t n,
                                                              const uint32_t *__restrict__ row_nprod,
                                                              const uint32_t *__restrict__ row_nnzc,
                                                              const uint8_t *__restrict__ row_bin, CutParams cp,
                                                              uint64_t *__restrict__ tile_nnz, uint64_t *__restrict__ tile_w)
{
    __shared__ uint64_t s_w[SCAN_BLOCK / 64];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint64_t sn = 0, sw = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        uint32_t nnz, w;
        int bin;
        scan_row_values<MODE>(base + j, n, aptr, r0, row_nprod, row_nnzc, row_bin, cp, nnz, w, bin);
        sn += nnz;
        sw += w;
    }
    uint64_t tot;
    block_exclusive_scan_u64(sw, s_w, &tot);
    if (threadIdx.x == 0) tile_w[blockIdx.x] = tot;
    if constexpr (MODE == 1) {
        __syncthreads();
        block_exclusive_scan_u64(sn, s_w, &tot);
        if (threadIdx.x == 0) tile_nnz[blockIdx.x] = tot;
    }
}

// single workgroup: exclusive scan of both tile arrays in place; totals -> [ntiles]; number of batches -> *nb
__global__ __launch_bounds__(SCAN_BLOCK) void k_cut_scan_tiles(uint64_t *__restrict__ tile_nnz, uint64_t *__restrict__ tile_w,
                                                               uint32_t ntiles, uint32_t cap, int mode,
                                                               uint32_t *__restrict__ nb)
{
    __shared__ uint64_t s_w[SCAN_BLOCK / 64];
    for (int pass = 0; pass < (mode == 1 ? 2 : 1); ++pass) {
        uint64_t *arr = pass == 0 ? tile_w : tile_nnz;
        uint64_t carry = 0;
        for (uint32_t b = 0; b < ntiles; b += SCAN_BLOCK) {
            const uint32_t i = b + threadIdx.x;
            uint64_t v = i < ntiles ? arr[i] : 0, tot;
            uint64_t ex = block_exclusive_scan_u64(v, s_w, &tot);
            if (i < ntiles) arr[i] = carry + ex;
            carry += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            arr[ntiles] = carry;
            if (pass == 0) *nb = (uint32_t)(carry / cap) + 1;
        }
        __syncthreads();
    }
}

template <int MODE>
__global__ __launch_bounds__(SCAN_BLOCK) void k_cut_apply(const uint64_t *__restrict__ aptr, uint64_t r0, uint32_t n,
                                                          const uint32_t *__restrict__ row_nprod,
                                                          const uint32_t *__restrict__ row_nnzc,
                                                          const uint8_t *__restrict__ row_bin_in, CutParams cp,
                                                          const uint64_t *__restrict__ tile_nnz,
                                                          const uint64_t *__restrict__ tile_w, uint32_t ntiles,
                                                          uint64_t *__restrict__ cptr /* n + 1, MODE 1 */,
                                                          uint8_t *__restrict__ row_bin_out /* MODE 1 */,
                                                          uint32_t *__restrict__ bin_counts /* MODE 1 */,
                                                          unsigned long long *__restrict__ bin_sums /* MODE 1: prod | nnz | entries */,
                                                          uint32_t *__restrict__ batch_first)
{
    __shared__ uint64_t s_w[SCAN_BLOCK / 64];
    __shared__ uint32_t s_hist[SPADA_N_BINS];
    __shared__ unsigned long long s_sum[3 * SPADA_N_BINS];
    if (threadIdx.x < SPADA_N_BINS) s_hist[threadIdx.x] = 0;
    if (threadIdx.x < 3 * SPADA_N_BINS) s_sum[threadIdx.x] = 0;
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t nnz[SCAN_ITEMS], w[SCAN_ITEMS];
    int bin[SCAN_ITEMS];
    uint64_t sn = 0, sw = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        scan_row_values<MODE>(base + j, n, aptr, r0, row_nprod, row_nnzc, row_bin_in, cp, nnz[j], w[j], bin[j]);
        sn += nnz[j];
        sw += w[j];
    }
    uint64_t tot;
    uint64_t exw = block_exclusive_scan_u64(sw, s_w, &tot) + tile_w[blockIdx.x];
    uint64_t exn = 0;
    if constexpr (MODE == 1) {
        __syncthreads();
        exn = block_exclusive_scan_u64(sn, s_w, &tot) + tile_nnz[blockIdx.x];
    }
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        const uint32_t i = base + j;
        if (i < n) {
            const uint64_t b_lo = exw / cp.cap, b_hi = (exw + w[j]) / cp.cap;
            if (b_hi > b_lo) batch_first[b_hi] = i + 1;   // w <= cap / 2: at most one multiple of cap is reached
            if constexpr (MODE == 1) {
                cptr[i] = exn;
                row_bin_out[i] = (uint8_t)bin[j];
                atomicAdd(&s_hist[bin[j]], 1u);
                if (bin[j] != BIN_EMPTY) {
                    atomicAdd(&s_sum[bin[j]], (unsigned long long)row_nprod[i]);
                    atomicAdd(&s_sum[SPADA_N_BINS + bin[j]], (unsigned long long)nnz[j]);
                    atomicAdd(&s_sum[2 * SPADA_N_BINS + bin[j]], (unsigned long long)(aptr[r0 + i + 1] - aptr[r0 + i]));
                }
            }
        }
        exw += w[j];
        exn += nnz[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        batch_first[0] = 0;
        if constexpr (MODE == 1) cptr[n] = tile_nnz[ntiles];
    }
    if constexpr (MODE == 1) {
        __syncthreads();
        if (threadIdx.x < SPADA_N_BINS && s_hist[threadIdx.x]) atomicAdd(&bin_counts[threadIdx.x], s_hist[threadIdx.x]);
        if (threadIdx.x < 3 * SPADA_N_BINS && s_sum[threadIdx.x]) atomicAdd(&bin_sums[threadIdx.x], s_sum[threadIdx.x]);
    }
}

// scatter the rows of the per-row bins (everything except empty / flat) into their lists; SC_ITEMS rows per
// thread so that the per-bin cursors see one global atomic per 4096 rows
constexpr int SC_ITEMS = 16;
__global__ __launch_bounds__(256) void k_bin_scatter2(const uint8_t *__restrict__ row_bin, uint32_t nrows,
                                                      const uint32_t *__restrict__ bin_counts,
                                                      uint32_t *__restrict__ bin_cursor, uint32_t *__restrict__ bin_rows)
{
    __shared__ uint32_t s_cnt[SPADA_N_BINS], s_base[SPADA_N_BINS];
    if (threadIdx.x < SPADA_N_BINS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (256 * SC_ITEMS) + threadIdx.x;
    int bin[SC_ITEMS];
    uint32_t local[SC_ITEMS];
#pragma unroll
    for (int j = 0; j < SC_ITEMS; ++j) {
        const uint32_t i = base + j * 256;
        bin[j] = -1;
        local[j] = 0;
        if (i < nrows) {
            const int b = row_bin[i];
            if (b != BIN_EMPTY && b != BIN_FLAT) {
                bin[j] = b;
                local[j] = atomicAdd(&s_cnt[b], 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < SPADA_N_BINS) {
        uint32_t off = 0;
        for (int u = 0; u < (int)threadIdx.x; ++u)
            if (u != BIN_EMPTY && u != BIN_FLAT) off += bin_counts[u];   // those two have no list
        uint32_t c = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = off + (c ? atomicAdd(&bin_cursor[threadIdx.x], c) : 0u);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SC_ITEMS; ++j)
        if (bin[j] >= 0) bin_rows[s_base[bin[j]] + local[j]] = base + j * 256;
}

// Largest-first order for the few huge rows: they run one row per persistent workgroup, so a static round robin leaves
// the CUs that drew the 40 k-product rows running long after the others (LPT scheduling: sort by products,
// descending, then dequeue dynamically).  One workgroup; lists longer than SORT_ROWS_MAX are left alone (with
// thousands of rows per CU the imbalance averages out).
constexpr int SORT_ROWS_MAX = 4096;
__global__ __launch_bounds__(1024) void k_sort_rows_desc(uint32_t *__restrict__ bin_rows, const uint32_t *__restrict__ bin_counts,
                                                         int bin, const uint32_t *__restrict__ row_nprod)
{
    __shared__ unsigned long long sk[SORT_ROWS_MAX];
    uint32_t off = 0;
    for (int u = 0; u < bin; ++u)
        if (u != BIN_EMPTY && u != BIN_FLAT) off += bin_counts[u];
    const uint32_t n = bin_counts[bin];
    if (n < 2 || n > (uint32_t)SORT_ROWS_MAX) return;
    uint32_t *list = bin_rows + off;
    uint32_t N = 2;
    while (N < n) N <<= 1;   // network size: next power of two
    for (uint32_t i = threadIdx.x; i < N; i += blockDim.x)
        sk[i] = i < n ? (((unsigned long long)(0xFFFFFFFFu - row_nprod[list[i]]) << 32) | list[i]) : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < N / 2; t += blockDim.x) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), x = i | j;
                const unsigned long long a = sk[i], c = sk[x];
                if ((a > c) == ((i & k) == 0)) {
                    sk[i] = c;
                    sk[x] = a;
                }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) list[i] = (uint32_t)sk[i];
}

// ---- 3. flat batches: shared pieces ---------------------------------------------------------------------------
constexpr uint32_t LR_NONE = 0xFFFFFFFFu;

// Within one wave the products handled by adjacent lanes belong to non-decreasing local rows; every run of
// equal `lr` adds its number of new keys to s_cnt[lr] with ONE LDS atomic (a per-lane atomic would serialise
// 64-fold on the same address).  Must be called by all lanes of the wave.
__device__ inline void segmented_count_add(uint32_t lr, bool isnew, uint32_t *s_cnt, int lane)
{
    const uint32_t prev = __shfl_up(lr, 1);
    const bool head = lane == 0 || prev != lr;
    const unsigned long long hm = __ballot(head), nm = __ballot(isnew);
    if (head && lr != LR_NONE) {
        const unsigned long long from = ~0ull << lane;                              // lanes >= this one
        const unsigned long long above = lane == 63 ? 0ull : (hm & (~0ull << (lane + 1)));
        const unsigned long long upto = above ? ((1ull << (__ffsll((long long)above) - 1)) - 1ull) : ~0ull;
        const uint32_t c = (uint32_t)__popcll(nm & from & upto);
        if (c) atomicAdd(&s_cnt[lr], c);
    }
}

// largest lr in [0, R) with s_re[lr] <= e   (s_re ascending, s_re[0] == 0, zero-length rows repeat a value)
template <int RMAX>
__device__ inline uint32_t row_of_entry(const uint32_t *s_re, uint32_t R, uint32_t e)
{
    uint32_t lo = 0;
#pragma unroll
    for (int step = RMAX / 2; step >= 1; step >>= 1)
        if (lo + step < R && s_re[lo + step] <= e) lo += step;
    return lo;
}

// ---- 4. the flat product walk ------------------------------------------------------------------------------------
// The A entries of the batch's flat rows are taken ECH = BLOCK * EPT at a time (each thread EPT consecutive
// entries: descriptor (begin, length) of the selected B row, the A value, the local row).  Entries that select an
// empty B row are dropped; one packed exclusive scan numbers the surviving entries and the products of the chunk
// densely.  Products are handled in windows of PWIN: every entry sets the bit of its first product in a window
// bitmap ("head bits"), one wave turns the word popcounts into prefix counts, and the owner entry of product p
// is   prefix[word(p)] + popcount(bits(word(p)) up to p) - 1   -- two broadcast LDS reads instead of a binary
// search.  Each wave takes 64 CONSECUTIVE products at a time (adjacent lanes read adjacent B entries, and the
// lanes of a wave hit the same or neighbouring entry records), U such segments per thread and round, so that U
// independent gathers and U independent first-probe LDS atomics are in flight.
// LDS scratch: entry records {pack = (begin - offset) mod 2^48 | local row << 48, a value} | bm u64[PWIN / 64] |
// bpre u32[PWIN / 64]
constexpr int FLAT_PWIN = 8192;
#ifndef SPADA_FLAT_U
#define SPADA_FLAT_U 4
#endif
#ifndef SPADA_FLAT_DENSE_RANK
#define SPADA_FLAT_DENSE_RANK 0   /* measured slower on every surrogate (two more barriers); kept for tightly clustered inputs */
#endif
#ifndef SPADA_FLAT_PREFETCH
#define SPADA_FLAT_PREFETCH 0
#endif
constexpr unsigned long long M48 = 0xFFFFFFFFFFFFull;

struct __attribute__((aligned(16))) EntryRecNum {
    uint64_t pack;
    double av;
};

template <int BLOCK, int EPT, bool NUMERIC>
__host__ __device__ constexpr size_t flat_walk_bytes()
{
    return (size_t)BLOCK * EPT * (NUMERIC ? 16 : 8) + (size_t)(FLAT_PWIN / 64) * 12 + 16;
}

template <int BLOCK, int EPT, int RMAX, bool NUMERIC, int U, class F>
__device__ inline void flat_walk(const uint32_t *s_re, const uint64_t *s_a0, uint32_t R, uint32_t E,
                                 const uint64_t *__restrict__ eb0, const uint32_t *__restrict__ elen,
                                 const double *__restrict__ aval, const uint32_t *__restrict__ bidx,
                                 const double *__restrict__ bval, unsigned char *scratch, uint32_t *hdr, F &&f,
                                 unsigned long long *wdbg = nullptr)
{
    uint32_t pbase = 0;   // products of the chunks already walked
#define WSTAMP(i) do { if (wdbg && threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); wdbg[i] += t_ - tprev; tprev = t_; } } while (0)
    unsigned long long tprev = wdbg ? __builtin_amdgcn_s_memtime() : 0;
    constexpr int ECH = BLOCK * EPT;
    constexpr int NW = BLOCK / 64;
    constexpr int PWORDS = FLAT_PWIN / 64;
    static_assert(PWORDS % 64 == 0 || PWORDS == 64 || PWORDS == 128, "prefix pass: whole words per lane");
    constexpr int WPL = PWORDS / 64;   // bitmap words per lane of the prefix wave
    EntryRecNum *w_ent = (EntryRecNum *)scratch;
    uint64_t *w_pack = (uint64_t *)scratch;
    unsigned long long *bm = (unsigned long long *)(scratch + (size_t)ECH * (NUMERIC ? 16 : 8));
    uint32_t *bpre = (uint32_t *)(bm + PWORDS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned long long lane_bit = 1ull << lane;
    for (uint32_t chunk = 0; chunk < E; chunk += ECH) {
        uint64_t b0[EPT];
        uint32_t len[EPT], lr[EPT], off[EPT];
        double av[EPT];
        {
            uint32_t ee[EPT];
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                ee[i] = chunk + tid * EPT + i;
                lr[i] = 0;
            }
#pragma unroll
            for (int step = RMAX / 2; step >= 1; step >>= 1) {   // EPT row searches in lock step
                uint32_t o[EPT];
#pragma unroll
                for (int i = 0; i < EPT; ++i) o[i] = lr[i] + step < R ? s_re[lr[i] + step] : 0xFFFFFFFFu;
#pragma unroll
                for (int i = 0; i < EPT; ++i) lr[i] += o[i] <= ee[i] ? step : 0;
            }
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                b0[i] = 0;
                len[i] = 0;
                av[i] = 0.0;
                if (ee[i] < E) {
                    const uint64_t a = s_a0[lr[i]] + (ee[i] - s_re[lr[i]]);
                    b0[i] = eb0[a];
                    len[i] = elen[a];
                    if constexpr (NUMERIC) av[i] = aval[a];
                }
            }
        }
        WSTAMP(0);
        // packed scan: (entries with products) << 32 | products
        unsigned long long mine = 0;
#pragma unroll
        for (int i = 0; i < EPT; ++i) mine += len[i] ? ((1ull << 32) | len[i]) : 0ull;
        unsigned long long tot64;
        unsigned long long ex64 = group_scan_excl_u64<BLOCK>(mine, tid, (unsigned long long *)(hdr + 4), &tot64);
        const uint32_t total = (uint32_t)tot64;
        WSTAMP(1);
        {
            uint32_t ci = (uint32_t)(ex64 >> 32), po = (uint32_t)ex64;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                off[i] = po;
                if (len[i]) {
                    const uint64_t pack = ((b0[i] - po) & M48) | ((uint64_t)lr[i] << 48);
                    if constexpr (NUMERIC) w_ent[ci] = EntryRecNum{pack, av[i]};
                    else w_pack[ci] = pack;
                    ++ci;
                    po += len[i];
                }
            }
        }
        for (uint32_t lo = 0; lo < total; lo += FLAT_PWIN) {
            const uint32_t hi = min(lo + (uint32_t)FLAT_PWIN, total);
            if (tid < PWORDS) bm[tid] = 0ull;
            if (tid == 0) hdr[0] = 0;
            __syncthreads();
            uint32_t before = 0;
#pragma unroll
            for (int i = 0; i < EPT; ++i)
                if (len[i]) {
                    if (off[i] >= lo && off[i] < hi) {
                        const uint32_t d = off[i] - lo;
                        atomicOr((uint32_t *)bm + (d >> 5), 1u << (d & 31));
                    }
                    before += off[i] < lo ? 1u : 0u;
                }
            if (lo) {   // entries whose first product lies before this window (uniform branch)
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
                if (lane == 0 && before) atomicAdd(&hdr[0], before);
            }
            __syncthreads();
            if (wave == 0) {
                uint32_t c[WPL], sum = 0;
#pragma unroll
                for (int k = 0; k < WPL; ++k) {
                    c[k] = (uint32_t)__popcll(bm[lane * WPL + k]);
                    sum += c[k];
                }
                uint32_t inc = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t t = __shfl_up(inc, o);
                    if (lane >= o) inc += t;
                }
                uint32_t run = hdr[0] + inc - sum;
#pragma unroll
                for (int k = 0; k < WPL; ++k) {
                    bpre[lane * WPL + k] = run;
                    run += c[k];
                }
            }
            __syncthreads();
            WSTAMP(2);
            for (uint32_t base = lo; base < hi; base += U * BLOCK) {   // uniform trip count over the workgroup
                uint32_t col[U], plr[U];
                double v[U];
                uint32_t pp[U], j[U];
                bool act[U];
                // lane l of a segment holds product seg + l, i.e. bit l of one bitmap word (lo, base and the segments are
                // multiples of 64): the word and its prefix are wave-uniform reads, the rank is a v_mbcnt pair.  Lanes past
                // the end fall back to product 0 of entry 0 (a valid address; their result is discarded).
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t seg = base + (u * NW + wave_u) * 64;      // wave-uniform
                    const uint32_t p = seg + lane;
                    act[u] = p < hi;
                    const uint32_t w = min((seg - lo) >> 6, (uint32_t)PWORDS - 1u);
                    const unsigned long long bits = bm[w];
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bits, 0u));
                    const uint32_t self = (bits & lane_bit) ? 1u : 0u;
                    j[u] = act[u] ? bpre[w] + below + self - 1u : 0u;
                    pp[u] = act[u] ? p : 0u;
                }
                uint64_t q[U];
                double a_[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint64_t pack;
                    a_[u] = 0.0;
                    if constexpr (NUMERIC) {
                        const EntryRecNum er = w_ent[j[u]];
                        pack = er.pack;
                        a_[u] = er.av;
                    } else {
                        pack = w_pack[j[u]];
                    }
                    q[u] = ((pack & M48) + pp[u]) & M48;
                    plr[u] = act[u] ? (uint32_t)(pack >> 48) : LR_NONE;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) col[u] = bidx[q[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    v[u] = 0.0;
                    if constexpr (NUMERIC) v[u] = a_[u] * bval[q[u]];   // simulator.rs:100-101
                }
                WSTAMP(3);
#pragma unroll
                for (int u = 0; u < U; ++u) pp[u] += pbase;
                f(col, plr, v, pp);
                WSTAMP(4);
            }
            __syncthreads();
        }
        pbase += total;
        WSTAMP(5);
    }
#undef WSTAMP
}

__device__ inline uint32_t compose_key(uint32_t lr, uint32_t col, uint32_t colbits)
{
    return colbits >= 32 ? col : ((lr << colbits) | col);
}

// per-row parameters of the ordered emission, one 16-byte LDS read per lookup
struct __attribute__((aligned(16))) RowEmit {
    uint32_t boff;    // first bucket (= first output slot inside the batch) of the row
    uint32_t n;       // nnz(C row)
    uint32_t kmin;    // smallest column that can occur
    float scale;      // n / (kmax - kmin + 1)
};

// ---- 5. symbolic, flat batches ---------------------------------------------------------------------------------
// LDS: 256 B hdr | keys u32[T] | s_re u32[RMAX + 1] | s_cnt u32[RMAX] | s_a0 u64[RMAX] | walk scratch
template <int BLOCK, int EPT, int LOG_T, int RMAX>
__host__ __device__ constexpr size_t sym_flat_lds()
{
    return 256 + ((size_t)4 << LOG_T) + (size_t)(RMAX + 1) * 4 + (size_t)RMAX * 4 + 8 + (size_t)RMAX * 8 +
           flat_walk_bytes<BLOCK, EPT, false>() + 16;
}

template <int BLOCK, int EPT, int LOG_T, int RMAX, bool LIST, int RPB>
__global__ __launch_bounds__(BLOCK, BLOCK / 128) void k_sym_flat(const uint64_t *__restrict__ aptr, const uint32_t *__restrict__ bidx,
                                                    const uint64_t *__restrict__ eb0, const uint32_t *__restrict__ elen,
                                                    uint64_t r0, uint32_t nrows, const uint8_t *__restrict__ row_bin,
                                                    const uint32_t *__restrict__ batch_first, const uint32_t *__restrict__ nb_ptr,
                                                    uint32_t colbits, uint32_t *__restrict__ row_nnzc,
                                                    const uint32_t *__restrict__ list, uint32_t want_bin,
                                                    unsigned long long *dbg = nullptr)
{
#define SSTAMP(i) do { if (dbg && threadIdx.x == 0 && b % 64 == 0 && b / 64 < 64) dbg[(b / 64) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    // LIST = false: batches of consecutive rows (batch_first); LIST = true: RPB consecutive rows of `list` per batch
    static_assert(RMAX <= BLOCK, "one thread per row of a batch");
    constexpr int T = 1 << LOG_T;
    constexpr int U = SPADA_FLAT_U;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    uint32_t *keys = (uint32_t *)(smem + 256);
    uint32_t *s_re = keys + T;
    uint32_t *s_cnt = s_re + RMAX + 1;
    uint64_t *s_a0 = (uint64_t *)(((uintptr_t)(s_cnt + RMAX) + 7) & ~(uintptr_t)7);
    unsigned char *scratch = (unsigned char *)(s_a0 + RMAX);
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t nl = *nb_ptr, G = gridDim.x;
    constexpr uint32_t rpb = RPB;
    const uint32_t nb = LIST ? (nl + rpb - 1) / rpb : nl;   // LIST: nl rows in the list, RPB of them per batch
    for (uint32_t b = blockIdx.x; b < nb; b += G) {
        uint32_t rb, re;   // rows [rb, re) of the matrix, or positions [rb, re) of the list
        if constexpr (LIST) {
            rb = b * rpb;
            re = min(rb + rpb, nl);
        } else {
            rb = batch_first[b];
            re = b + 1 < nb ? batch_first[b + 1] : nrows;
        }
        const uint32_t R = re - rb;   // <= RMAX by construction of the cut
        if (R == 0) continue;
        SSTAMP(0);
        uint32_t L = 0, rid = 0;
        bool flat = false;
        if ((uint32_t)tid < R) {
            rid = LIST ? list[rb + tid] : rb + tid;
            const uint64_t a0 = aptr[r0 + rid], a1 = aptr[r0 + rid + 1];
            flat = row_bin[rid] == want_bin;
            s_a0[tid] = a0;
            s_cnt[tid] = 0;
            if (flat) L = (uint32_t)(a1 - a0);
        }
        uint32_t E;
        const uint32_t ex = group_scan_excl<BLOCK>(L, tid, hdr + 2, &E);
        if ((uint32_t)tid < R) s_re[tid] = ex;
        if (tid == 0) s_re[R] = E;
        uint4 *k4 = (uint4 *)keys;
        for (int s = tid; s < T / 4; s += BLOCK) k4[s] = make_uint4(EMPTY_KEY, EMPTY_KEY, EMPTY_KEY, EMPTY_KEY);
        __syncthreads();
        SSTAMP(1);
        flat_walk<BLOCK, EPT, RMAX, false, U>(
            s_re, s_a0, R, E, eb0, elen, nullptr, bidx, nullptr, scratch, hdr, [&](uint32_t(&col)[U], uint32_t(&plr)[U], double(&)[U], uint32_t(&)[U]) {
                uint32_t key[U], h[U], old[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    key[u] = compose_key(plr[u], col[u], colbits);
                    h[u] = hash_slot<LOG_T>(key[u]);
                    old[u] = key[u];
                    if (plr[u] != LR_NONE) old[u] = atomicCAS(&keys[h[u]], EMPTY_KEY, key[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    bool isnew = old[u] == EMPTY_KEY;
                    if (!isnew && old[u] != key[u]) {
                        uint32_t hh = h[u];
                        for (;;) {
                            hh = (hh + 1) & (T - 1);
                            const uint32_t o = atomicCAS(&keys[hh], EMPTY_KEY, key[u]);
                            if (o == EMPTY_KEY) { isnew = true; break; }
                            if (o == key[u]) break;
                        }
                    }
                    segmented_count_add(plr[u], isnew, s_cnt, lane);
                }
            }, (dbg && b % 64 == 0 && b / 64 < 64) ? dbg + (b / 64) * 16 + 8 : nullptr);
        SSTAMP(2);
        if (flat) row_nnzc[rid] = s_cnt[tid];
        __syncthreads();
        SSTAMP(3);
    }
#undef SSTAMP
}

// ---- 6. numeric, flat batches -----------------------------------------------------------------------------------
// LDS: 256 B hdr | table: keys u32[T], vals f64[T]  (re-used after accumulation as lk u32[NOUT], lv f64[NOUT])
//      | region 2: bcnt u32[NOUT], aliased by the walk scratch (disjoint phases)
//      | rows: s_re u32[RMAX+1], s_row RowEmit[RMAX + 1], s_a0 u64[RMAX], s_out u64[RMAX]
template <int BLOCK, int EPT, int NOUT>
__host__ __device__ constexpr size_t num_flat_region2()
{
    return (flat_walk_bytes<BLOCK, EPT, true>() > (size_t)NOUT * 4 ? flat_walk_bytes<BLOCK, EPT, true>() : (size_t)NOUT * 4) + 16;
}
template <int BLOCK, int EPT, int LOG_T, int NOUT, int RMAX>
__host__ __device__ constexpr size_t num_flat_lds()
{
    return 256 + ((size_t)12 << LOG_T) + num_flat_region2<BLOCK, EPT, NOUT>() + (size_t)(RMAX + 1) * 4 + 16 +
           (size_t)(RMAX + 1) * 16 + (size_t)RMAX * 16 + 32;
}

template <int BLOCK, int EPT, int LOG_T, int NOUT, int RMAX, bool LIST, int RPB>
__global__ __launch_bounds__(BLOCK, BLOCK / 128) void k_num_flat(const uint64_t *__restrict__ aptr, const double *__restrict__ aval,
                                                    const uint32_t *__restrict__ bidx, const double *__restrict__ bval,
                                                    const uint64_t *__restrict__ eb0, const uint32_t *__restrict__ elen,
                                                    uint64_t r0, uint32_t nrows, const uint8_t *__restrict__ row_bin,
                                                    const uint32_t *__restrict__ row_kmin, const uint32_t *__restrict__ row_kmax,
                                                    const uint64_t *__restrict__ cptr, const uint32_t *__restrict__ batch_first,
                                                    const uint32_t *__restrict__ nb_ptr, uint32_t colbits,
                                                    uint32_t *__restrict__ c_idx, double *__restrict__ c_val,
                                                    unsigned long long *dbg, const uint32_t *__restrict__ list,
                                                    uint32_t want_bin)
{
#define STAMP(i) do { if (dbg && threadIdx.x == 0 && b % 64 == 0 && b / 64 < 64) dbg[(b / 64) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    static_assert(RMAX <= BLOCK, "one thread per row of a batch");
    static_assert(NOUT % BLOCK == 0, "flat scan length");
    static_assert((size_t)NOUT * 12 <= ((size_t)12 << LOG_T), "compacted (key, value) lists re-use the table");
    constexpr int T = 1 << LOG_T;
    constexpr int SPT = T / BLOCK;   // table slots per thread
    constexpr int U = SPADA_FLAT_U;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    uint32_t *keys = (uint32_t *)(smem + 256);
    double *vals = (double *)(keys + T);
    uint32_t *lk = keys;                                                              // keys in bucket order
    double *lv = (double *)(smem + 256 + (((size_t)NOUT * 4 + 7) & ~(size_t)7));      // and their values
    unsigned char *region2 = smem + 256 + ((size_t)12 << LOG_T);
    uint32_t *bcnt = (uint32_t *)region2;
    unsigned char *rows = region2 + ((num_flat_region2<BLOCK, EPT, NOUT>() + 15) & ~(size_t)15);
    RowEmit *s_row = (RowEmit *)rows;
    uint64_t *s_a0 = (uint64_t *)(s_row + RMAX + 1);
    uint64_t *s_out = s_a0 + RMAX;
    uint32_t *s_re = (uint32_t *)(s_out + RMAX);
    const int tid = threadIdx.x;
    const uint32_t nl = *nb_ptr, G = gridDim.x;
    constexpr uint32_t rpb = RPB;
    const uint32_t nb = LIST ? (nl + rpb - 1) / rpb : nl;   // LIST: nl rows in the list, RPB of them per batch
    const uint32_t colmask = colbits >= 32 ? 0xFFFFFFFFu : ((1u << colbits) - 1u);

    for (uint32_t b = blockIdx.x; b < nb; b += G) {
        uint32_t rb, re;   // rows [rb, re) of the matrix, or positions [rb, re) of the list
        if constexpr (LIST) {
            rb = b * rpb;
            re = min(rb + rpb, nl);
        } else {
            rb = batch_first[b];
            re = b + 1 < nb ? batch_first[b + 1] : nrows;
        }
        const uint32_t R = re - rb;
        if (R == 0) continue;
        STAMP(0);
        uint32_t L = 0, n = 0, kmin = 0, kmax = 0;
        if ((uint32_t)tid < R) {
            const uint32_t rid = LIST ? list[rb + tid] : rb + tid;
            const uint64_t a0 = aptr[r0 + rid], a1 = aptr[r0 + rid + 1];
            const uint64_t c0 = cptr[rid], c1 = cptr[rid + 1];
            s_a0[tid] = a0;
            s_out[tid] = c0;
            if (row_bin[rid] == want_bin) {
                L = (uint32_t)(a1 - a0);
                n = (uint32_t)(c1 - c0);
                kmin = row_kmin[rid];
                kmax = row_kmax[rid];
            }
        }
        uint32_t E, NO;
        if constexpr (LIST && RPB == 1) {
            // a single row: no scan, thread 0 publishes its two totals through the header
            if (tid == 0) {
                s_re[0] = 0;
                s_re[1] = L;
                s_row[0] = RowEmit{0u, n, kmin, (float)n / ((float)(kmax - kmin) + 1.0f)};
                hdr[40] = L;
                hdr[41] = n;
            }
        } else {
            // one packed scan: entries of the flat rows in the low, their outputs in the high 32 bits
            unsigned long long tot64;
            const unsigned long long ex64 = group_scan_excl_u64<BLOCK>(((unsigned long long)n << 32) | L, tid,
                                                                       (unsigned long long *)(hdr + 4), &tot64);
            const uint32_t exl = (uint32_t)ex64, exn = (uint32_t)(ex64 >> 32);
            if ((uint32_t)tid < R) {
                s_re[tid] = exl;
                s_row[tid] = RowEmit{exn, n, kmin, (float)n / ((float)(kmax - kmin) + 1.0f)};
            }
            if (tid == 0) {
                s_re[R] = (uint32_t)tot64;
                hdr[40] = (uint32_t)tot64;
                hdr[41] = (uint32_t)(tot64 >> 32);
            }
        }
        {
            uint4 *k4 = (uint4 *)keys;
            for (int s = tid; s < T / 4; s += BLOCK) k4[s] = make_uint4(EMPTY_KEY, EMPTY_KEY, EMPTY_KEY, EMPTY_KEY);
            double2 *v2 = (double2 *)vals;
            for (int s = tid; s < T / 2; s += BLOCK) v2[s] = make_double2(0.0, 0.0);
        }
        __syncthreads();
        E = hdr[40];
        NO = hdr[41];
        STAMP(1);

        // ---- expand - scale - accumulate --------------------------------------------------------------------
        flat_walk<BLOCK, EPT, RMAX, true, U>(
            s_re, s_a0, R, E, eb0, elen, aval, bidx, bval, region2, hdr, [&](uint32_t(&col)[U], uint32_t(&plr)[U], double(&v)[U], uint32_t(&)[U]) {
                uint32_t key[U], h[U], old[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    key[u] = compose_key(plr[u], col[u], colbits);
                    h[u] = hash_slot<LOG_T>(key[u]);
                    old[u] = key[u];
                    if (plr[u] != LR_NONE) old[u] = atomicCAS(&keys[h[u]], EMPTY_KEY, key[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (old[u] != EMPTY_KEY && old[u] != key[u]) {
                        for (;;) {
                            h[u] = (h[u] + 1) & (T - 1);
                            const uint32_t o = atomicCAS(&keys[h[u]], EMPTY_KEY, key[u]);
                            if (o == EMPTY_KEY || o == key[u]) break;
                        }
                    }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (plr[u] != LR_NONE) atomicAdd(&vals[h[u]], v[u]);   // simulator.rs:213-218 (order differs)
            }, (dbg && b % 64 == 0 && b / 64 < 64) ? dbg + (b / 64) * 16 + 8 : nullptr);
        STAMP(2);

        // ---- ordered emission -------------------------------------------------------------------------------
        // every occupied slot -> bucket = boff[lr] + floor((col - kmin) * n / span): monotone inside a row and
        // rows are laid out in order, so the bucket order IS the order of the batch's slice of C up to
        // permutations inside one bucket
        for (int s = tid; s < NOUT; s += BLOCK) bcnt[s] = 0;
        __syncthreads();
        uint32_t myk[SPT];
        uint16_t myb[SPT];
#pragma unroll
        for (int i = 0; i < SPT; ++i) myk[i] = keys[tid + i * BLOCK];
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            myb[i] = 0;
            if (myk[i] != EMPTY_KEY) {
                const uint32_t lr = colbits >= 32 ? 0u : (myk[i] >> colbits), col = myk[i] & colmask;
                const RowEmit rw = s_row[lr];
                uint32_t bk = (uint32_t)((float)(col - rw.kmin) * rw.scale);
                bk = bk < rw.n ? bk : rw.n - 1;
                myb[i] = (uint16_t)(rw.boff + bk);
                atomicAdd(&bcnt[myb[i]], 1u);
            }
        }
        double myv[SPT];
#pragma unroll
        for (int i = 0; i < SPT; ++i) myv[i] = vals[tid + i * BLOCK];
        __syncthreads();
        STAMP(3);
        group_exclusive_scan<BLOCK, NOUT>(bcnt, tid, hdr + 2);   // ends with a barrier: table fully read by now
        STAMP(4);
#pragma unroll
        for (int i = 0; i < SPT; ++i)
            if (myk[i] != EMPTY_KEY) {
                const uint32_t p = atomicAdd(&bcnt[myb[i]], 1u);   // afterwards bcnt[b] = end of bucket b
                lk[p] = myk[i];
                lv[p] = myv[i];
            }
        __syncthreads();
        STAMP(5);
        // Rank inside a bucket.  Buckets with fewer than 4 entries count smaller keys (<= 9 reads).  A larger bucket is
        // usually a cluster of neighbouring columns (site-local links, mesh neighbours): if its keys span at most
        // 32 * (m - 2) columns, the m words aux[lo .. hi) that belong to the bucket hold {min, max, bitmap of
        // (key - min)} and the rank is a prefix popcount -- O(span / 32) per entry instead of O(m).
        constexpr int OPT = NOUT / BLOCK;   // outputs per thread
        uint32_t *aux = (uint32_t *)(smem + 256 + (size_t)NOUT * 12);
        static_assert(!SPADA_FLAT_DENSE_RANK || ((size_t)12 << LOG_T) >= (size_t)NOUT * 16, "aux words live behind the (key, value) lists");
        uint32_t ok_[OPT], olo[OPT], ohi[OPT], obase[OPT];
        uint64_t oout[OPT];
        bool any_heavy = false;
#pragma unroll
        for (int w = 0; w < OPT; ++w) {
            const uint32_t p = tid + w * BLOCK;
            ok_[w] = p < NO ? lk[p] : EMPTY_KEY;
            olo[w] = ohi[w] = obase[w] = 0;
            oout[w] = 0;
        }
#pragma unroll
        for (int w = 0; w < OPT; ++w)
            if (ok_[w] != EMPTY_KEY) {
                const uint32_t lr = colbits >= 32 ? 0u : (ok_[w] >> colbits), col = ok_[w] & colmask;
                const RowEmit rw = s_row[lr];
                uint32_t bk = (uint32_t)((float)(col - rw.kmin) * rw.scale);
                bk = rw.boff + (bk < rw.n ? bk : rw.n - 1);
                olo[w] = bk ? bcnt[bk - 1] : 0u;
                ohi[w] = bcnt[bk];
                obase[w] = rw.boff;
                oout[w] = s_out[lr];
                if (SPADA_FLAT_DENSE_RANK && ohi[w] - olo[w] >= 4) {
                    any_heavy = true;
                    const uint32_t p = tid + w * BLOCK;
                    aux[p] = p == olo[w] ? 0xFFFFFFFFu : 0u;   // [lo] = min, [lo + 1] = max, the rest = bitmap words
                }
            }
        const bool heavy_block = SPADA_FLAT_DENSE_RANK ? __syncthreads_or(any_heavy) : false;
        uint32_t omn[OPT], ospan[OPT];
        if (heavy_block) {
#pragma unroll
            for (int w = 0; w < OPT; ++w)
                if (ohi[w] - olo[w] >= 4) {
                    atomicMin(&aux[olo[w]], ok_[w]);
                    atomicMax(&aux[olo[w] + 1], ok_[w]);
                }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < OPT; ++w) {
                omn[w] = 0;
                ospan[w] = 0xFFFFFFFFu;
                const uint32_t m = ohi[w] - olo[w];
                if (m >= 4) {
                    omn[w] = aux[olo[w]];
                    const uint32_t span = aux[olo[w] + 1] - omn[w];
                    if (span < 32u * (m - 2)) ospan[w] = span;
                }
            }
#pragma unroll
            for (int w = 0; w < OPT; ++w)
                if (ospan[w] != 0xFFFFFFFFu) {
                    const uint32_t d = ok_[w] - omn[w];
                    atomicOr(&aux[olo[w] + 2 + (d >> 5)], 1u << (d & 31));
                }
            __syncthreads();
        }
#pragma unroll
        for (int w = 0; w < OPT; ++w)
            if (ok_[w] != EMPTY_KEY) {
                uint32_t r = olo[w];
                if (heavy_block && ohi[w] - olo[w] >= 4 && ospan[w] != 0xFFFFFFFFu) {
                    const uint32_t d = ok_[w] - omn[w];
                    const uint32_t *bmw = aux + olo[w] + 2;
                    for (uint32_t q = 0; q < (d >> 5); ++q) r += __popc(bmw[q]);
                    r += __popc(bmw[d >> 5] & ((1u << (d & 31)) - 1u));
                } else {
                    for (uint32_t j = olo[w]; j < ohi[w]; ++j) r += (lk[j] < ok_[w]) ? 1u : 0u;
                }
                const uint64_t pos = oout[w] + (r - obase[w]);
                c_idx[pos] = ok_[w] & colmask;
                c_val[pos] = lv[tid + w * BLOCK];
            }
        __syncthreads();
        STAMP(6);
    }
#undef STAMP
}

// ---- 6b. numeric, rows with a single A nonzero: C row = a * B row (already ascending) ---------------------------------
// Matrix order, one lane per row: the row pointers, bins and C offsets of 64 consecutive rows are three coalesced
// loads, the (descriptor, value) of the single A entry one gather.  Short B rows are copied by their own lane
// (independent iterations, 4 in flight); rows longer than COPY_SHORT (A/B: 8 beats 24 and 64) are handed to the whole wave one after the
// other, 64 entries per step.
constexpr uint32_t COPY_SHORT = 8;
__global__ __launch_bounds__(256) void k_num_copy2(const uint64_t *__restrict__ aptr, const double *__restrict__ aval,
                                                   const uint32_t *__restrict__ bidx, const double *__restrict__ bval,
                                                   const uint64_t *__restrict__ eb0, const uint32_t *__restrict__ elen,
                                                   uint64_t r0, uint32_t nrows, const uint8_t *__restrict__ row_bin,
                                                   const uint64_t *__restrict__ cptr, uint32_t *__restrict__ c_idx,
                                                   double *__restrict__ c_val)
{
    const int lane = threadIdx.x & 63;
    for (uint32_t tile = blockIdx.x * blockDim.x; tile < nrows; tile += gridDim.x * blockDim.x) {
        const uint32_t i = tile + threadIdx.x;
        uint64_t b0 = 0, c0 = 0;
        uint32_t len = 0;
        double av = 0.0;
        if (i < nrows && row_bin[i] == BIN_COPY) {
            const uint64_t a = aptr[r0 + i];
            c0 = cptr[i];
            b0 = eb0[a];
            len = elen[a];
            av = aval[a];
        }
        if (len <= COPY_SHORT) {
            for (uint32_t t = 0; t < len; t += 4) {
                uint32_t k[4];
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t tt = t + u < len ? t + u : len - 1;
                    k[u] = bidx[b0 + tt];
                    v[u] = bval[b0 + tt];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (t + u < len) {
                        c_idx[c0 + t + u] = k[u];
                        c_val[c0 + t + u] = av * v[u];
                    }
            }
        }
        unsigned long long mask = __ballot(len > COPY_SHORT);
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const uint64_t sb0 = __shfl(b0, src), sc0 = __shfl(c0, src);
            const uint32_t slen = __shfl(len, src);
            const double sav = __shfl(av, src);
            for (uint32_t t = lane; t < slen; t += 64) {
                c_idx[sc0 + t] = bidx[sb0 + t];
                c_val[sc0 + t] = sav * bval[sb0 + t];
            }
        }
    }
}

// ---- 6c. numeric, rows with few but long B rows: multiway merge, one wavefront per row ------------------------------
// A C row whose A row has L <= 32 entries is the union of L sorted B rows: no hash table and no sort are needed, the
// reference's own answer -- a comparator merge tree followed by an adder (adder_tree.rs:145-188, :73-83) -- is also the
// cheapest one here.  One 64-lane wavefront per row, nothing but wave-level synchronisation: the products are
// expanded to LDS as keys (column << PB | product number), ceil(log2 L) merge-path rounds merge neighbouring runs
// pairwise (every lane produces a contiguous slice of the output: one diagonal search, then a sequential two-way
// merge), runs of equal column are added in ascending product number (= ascending k: bit-identical to a sequential
// CPU sort-merge) and stored.  ~5 LDS operations per product and round instead of ~50 for hash + ordered emission.
// LDS per wave: kA, kB u32[PMAX] | val f64[PMAX] | ebase u64[32] | eav f64[32] | off u32[34] | heads u64[PMAX/64] |
// hpre u32[PMAX/64]
template <int PMAX>
__host__ __device__ constexpr size_t num_merge_wave_bytes()
{
    return (size_t)PMAX * 16 + 32 * 16 + 34 * 4 + 8 + (size_t)(PMAX / 64) * 12 + 16;
}

template <int PMAX>
__global__ __launch_bounds__(256) void k_num_merge(const uint64_t *__restrict__ aptr, const double *__restrict__ aval,
                                                   const uint32_t *__restrict__ bidx, const double *__restrict__ bval,
                                                   const uint64_t *__restrict__ eb0, const uint32_t *__restrict__ elen,
                                                   uint64_t r0, const uint32_t *__restrict__ list, uint32_t n_list,
                                                   const uint64_t *__restrict__ cptr, uint32_t *__restrict__ c_idx,
                                                   double *__restrict__ c_val)
{
    constexpr int PB = PMAX == 1024 ? 10 : 9;
    static_assert(PMAX == 1024 || PMAX == 512, "product-number bits");
    constexpr uint32_t PMASK = (1u << PB) - 1u, MAXK = 0xFFFFFFFFu;
    constexpr int NCH = PMAX / 64;
    constexpr size_t WB = (num_merge_wave_bytes<PMAX>() + 15) & ~(size_t)15;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *base = smem + (size_t)wave * WB;
    uint32_t *kA = (uint32_t *)base, *kB = kA + PMAX;
    double *val = (double *)(kB + PMAX);
    uint64_t *ebase = (uint64_t *)(val + PMAX);
    double *eav = (double *)(ebase + 32);
    unsigned long long *heads = (unsigned long long *)(eav + 32);
    uint32_t *hpre = (uint32_t *)(heads + NCH);
    uint32_t *off = hpre + NCH;   // 34 entries: off[j] = first product of entry j, off[j >= L] = P
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    for (uint32_t slot = blockIdx.x * 4 + wave; slot < n_list; slot += gridDim.x * 4) {
        const uint32_t row = list[slot];
        const uint64_t a0 = aptr[r0 + row];
        const uint32_t L = (uint32_t)(aptr[r0 + row + 1] - a0);   // 2 .. 32
        const uint64_t c0 = cptr[row];
        uint64_t b0 = 0;
        uint32_t len = 0;
        double av = 0.0;
        if ((uint32_t)lane < L) {
            b0 = eb0[a0 + lane];
            len = elen[a0 + lane];
            av = aval[a0 + lane];
        }
        uint32_t inc = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        const uint32_t P = __shfl(inc, 63);   // <= PMAX by classification
        if (lane < 34) off[lane] = (uint32_t)lane < L ? inc - len : P;
        if (lane < 32) {
            ebase[lane] = b0 - (inc - len);
            eav[lane] = av;
        }
        wave_sync();
        // expand + scale (simulator.rs:86-111)
        for (uint32_t p = lane; p < P; p += 64) {
            uint32_t j = 0;
#pragma unroll
            for (uint32_t step = 16; step >= 1; step >>= 1)
                if (off[j + step] <= p) j += step;
            const uint64_t q = ebase[j] + p;
            kA[p] = (bidx[q] << PB) | p;
            val[p] = eav[j] * bval[q];
        }
        wave_sync();
        // merge tree (adder_tree.rs:145-188): runs of w lists -> runs of 2w lists
        uint32_t *src = kA, *dst = kB;
        const uint32_t VT = (P + 63) / 64;
        for (uint32_t w = 1; w < L; w <<= 1) {
            uint32_t o = lane * VT;
            const uint32_t oend = min(o + VT, P);
            while (o < oend) {
                uint32_t j = 0;   // first list of the pair that produces output o: a multiple of 2w
#pragma unroll
                for (uint32_t step = 16; step >= 1; step >>= 1)
                    if (step >= 2 * w && off[min(j + step, 32u)] <= o) j += step;
                const uint32_t s0 = off[j], s1 = off[min(j + w, 32u)], s2 = off[min(j + 2 * w, 32u)];
                const uint32_t nx = s1 - s0, ny = s2 - s1, k = o - s0;
                uint32_t lo = k > ny ? k - ny : 0u, hi = k < nx ? k : nx;
                while (lo < hi) {   // merge path: how many of the first k outputs come from the left run
                    const uint32_t mid = (lo + hi) >> 1;
                    if (src[s0 + mid] < src[s1 + (k - 1 - mid)]) lo = mid + 1;
                    else hi = mid;
                }
                uint32_t i = lo, jy = k - lo;
                const uint32_t lim = min(oend, s2);
                uint32_t x = i < nx ? src[s0 + i] : MAXK, y = jy < ny ? src[s1 + jy] : MAXK;
                while (o < lim) {   // branch-free two-way merge step: one LDS read per output
                    const bool tx = x < y;
                    dst[o] = tx ? x : y;
                    i += tx ? 1u : 0u;
                    jy += tx ? 0u : 1u;
                    const bool in = tx ? i < nx : jy < ny;
                    const uint32_t nxt = src[in ? (tx ? s0 + i : s1 + jy) : 0u];
                    x = tx ? (in ? nxt : MAXK) : x;
                    y = tx ? y : (in ? nxt : MAXK);
                    ++o;
                }
            }
            wave_sync();
            uint32_t *t = src;
            src = dst;
            dst = t;
        }
        // adder (adder_tree.rs:73-83): runs of equal column, added left to right
        const uint32_t nch = (P + 63) / 64;
        for (uint32_t ch = 0; ch < nch; ++ch) {
            const uint32_t p = ch * 64 + lane;
            const uint32_t cur = p < P ? src[p] : MAXK, prev = (p > 0 && p < P) ? src[p - 1] : MAXK;
            const bool head = p < P && (p == 0 || (cur >> PB) != (prev >> PB));
            const unsigned long long m = __ballot(head);
            if (lane == 0) heads[ch] = m;
        }
        wave_sync();
        {
            const uint32_t c = (uint32_t)lane < nch ? (uint32_t)__popcll(heads[lane]) : 0u;
            uint32_t ic = c;
#pragma unroll
            for (int o = 1; o < NCH; o <<= 1) {
                const uint32_t t = __shfl_up(ic, o);
                if (lane >= o) ic += t;
            }
            if (lane < NCH) hpre[lane] = ic - c;
        }
        wave_sync();
        for (uint32_t p = lane; p < P; p += 64) {
            const unsigned long long hw = heads[p >> 6];
            if (!((hw >> (p & 63)) & 1ull)) continue;
            const uint32_t rank = hpre[p >> 6] + (uint32_t)__popcll(hw & ((1ull << (p & 63)) - 1ull));
            const uint32_t key = src[p];
            double acc = val[key & PMASK];
            for (uint32_t q = p + 1; q < P && (src[q] >> PB) == (key >> PB); ++q) acc += val[src[q] & PMASK];
            c_idx[c0 + rank] = key >> PB;
            c_val[c0 + rank] = acc;
        }
        wave_sync();
    }
}

// ---- 7. numeric, flat batches, SORT-MERGE accumulator (SPADA_ACC_SORT_MERGE) ---------------------------------------
// The closest GPU analogue of what the reference's PE does to one group: collect the products
// (simulator.rs:86-111), sort them by column (SortingNetwork, simulator.rs:143-171), add runs of equal column left
// to right (MergeTree, simulator.rs:199-230).  One workgroup per symbolic batch (rows with P <= SYM_FLAT_MAX, less
// than SM_NP products per batch): every product is written to LDS as  key = (local row, column, product number)
// packed in 64 bits  +  value; a bitonic network sorts the keys; the first product of every run adds its run in
// ascending product number -- i.e. in ascending k, the order of the CPU restatement, so the values are
// bit-identical to a sequential CPU sort-merge.  Kept for the accumulator comparison of BASELINE.json configs[2]; the LDS-hash
// kernels are the fast path.
// LDS: 256 B hdr | sk u64[SM_NP] | sv f64[SM_NP] | heads u64[SM_NP / 64] | hpre u32[SM_NP / 64] | walk scratch |
//      rows: s_re, s_boff u32[RMAX + 1], s_a0, s_out u64[RMAX]
#ifndef SPADA_SF_LARGE   /* default: 256-thread workgroups, 4096-key tables (A/B: -17..21 % symbolic time) */
constexpr int SM_NP = 4096;
#else
constexpr int SM_NP = 8192;
#endif
static_assert(SM_NP >= SYM_FLAT_CAP + SYM_FLAT_MAX, "a symbolic batch holds fewer than cap + max products");

template <int BLOCK, int EPT, int RMAX>
__host__ __device__ constexpr size_t num_sm_lds()
{
    return 256 + (size_t)SM_NP * 16 + (size_t)(SM_NP / 64) * 12 + flat_walk_bytes<BLOCK, EPT, true>() + 16 +
           (size_t)(RMAX + 1) * 8 + (size_t)RMAX * 16 + 32;
}

template <int BLOCK, int EPT, int RMAX>
__global__ __launch_bounds__(BLOCK) void k_num_sortmerge(const uint64_t *__restrict__ aptr, const double *__restrict__ aval,
                                                         const uint32_t *__restrict__ bidx, const double *__restrict__ bval,
                                                         const uint64_t *__restrict__ eb0, const uint32_t *__restrict__ elen,
                                                         uint64_t r0, uint32_t nrows, const uint8_t *__restrict__ row_bin,
                                                         const uint64_t *__restrict__ cptr,
                                                         const uint32_t *__restrict__ batch_first,
                                                         const uint32_t *__restrict__ nb_ptr, uint32_t colbits,
                                                         uint32_t *__restrict__ c_idx, double *__restrict__ c_val)
{
    static_assert(RMAX <= BLOCK && BLOCK % 64 == 0, "one thread per row of a batch");
    constexpr int U = 4;
    constexpr int HW = SM_NP / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    unsigned long long *sk = (unsigned long long *)(smem + 256);
    double *sv = (double *)(sk + SM_NP);
    unsigned long long *heads = (unsigned long long *)(sv + SM_NP);
    uint32_t *hpre = (uint32_t *)(heads + HW);
    unsigned char *scratch = (unsigned char *)(hpre + HW);
    unsigned char *rows = scratch + ((flat_walk_bytes<BLOCK, EPT, true>() + 15) & ~(size_t)15);
    uint64_t *s_a0 = (uint64_t *)rows;
    uint64_t *s_out = s_a0 + RMAX;
    uint32_t *s_re = (uint32_t *)(s_out + RMAX);
    uint32_t *s_boff = s_re + RMAX + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nb = *nb_ptr;
    const uint32_t colmask = colbits >= 32 ? 0xFFFFFFFFu : ((1u << colbits) - 1u);
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint32_t rb = batch_first[b];
        const uint32_t re = b + 1 < nb ? batch_first[b + 1] : nrows;
        const uint32_t R = re - rb;
        if (R == 0) continue;
        uint32_t L = 0, n = 0;
        if ((uint32_t)tid < R) {
            const uint64_t a0 = aptr[r0 + rb + tid], a1 = aptr[r0 + rb + tid + 1];
            const uint64_t c0 = cptr[rb + tid], c1 = cptr[rb + tid + 1];
            s_a0[tid] = a0;
            s_out[tid] = c0;
            if (row_bin[rb + tid] == BIN_FLAT) {
                L = (uint32_t)(a1 - a0);
                n = (uint32_t)(c1 - c0);
            }
        }
        uint32_t E, NO;
        const uint32_t exl = group_scan_excl<BLOCK>(L, tid, hdr + 2, &E);
        __syncthreads();
        const uint32_t exn = group_scan_excl<BLOCK>(n, tid, hdr + 2, &NO);
        if ((uint32_t)tid < R) {
            s_re[tid] = exl;
            s_boff[tid] = exn;
        }
        if (tid == 0) s_re[R] = E;
        for (int s = tid; s < SM_NP; s += BLOCK) sk[s] = ~0ull;   // padding sorts to the end
        __syncthreads();
        // expand + scale: product number p of the batch -> sk[p], sv[p]
        flat_walk<BLOCK, EPT, RMAX, true, U>(s_re, s_a0, R, E, eb0, elen, aval, bidx, bval, scratch, hdr,
                                             [&](uint32_t(&col)[U], uint32_t(&plr)[U], double(&v)[U], uint32_t(&pp)[U]) {
#pragma unroll
                                                 for (int u = 0; u < U; ++u)
                                                     if (plr[u] != LR_NONE) {
                                                         const uint32_t key = compose_key(plr[u], col[u], colbits);
                                                         sk[pp[u]] = ((unsigned long long)key << 32) | pp[u];
                                                         sv[pp[u]] = v[u];
                                                     }
                                             });
        __syncthreads();
        // sort by (row, column, product number): bitonic network (simulator.rs:160 sorts each group by column)
        for (uint32_t k = 2; k <= (uint32_t)SM_NP; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = tid; t < (uint32_t)SM_NP / 2; t += BLOCK) {
                    const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // i has bit j clear
                    const uint32_t x = i | j;
                    const unsigned long long a = sk[i], c = sk[x];
                    const bool asc = (i & k) == 0;
                    if ((a > c) == asc) {
                        sk[i] = c;
                        sk[x] = a;
                    }
                }
                __syncthreads();
            }
        // runs of equal (row, column): head bits, their prefix counts, left-to-right sums (simulator.rs:209-220)
        for (uint32_t w = wave; w < (uint32_t)HW; w += BLOCK / 64) {
            const uint32_t p = w * 64 + lane;
            const unsigned long long cur = sk[p], prev = p ? sk[p - 1] : ~0ull;
            const bool head = cur != ~0ull && (p == 0 || (cur >> 32) != (prev >> 32));
            const unsigned long long m = __ballot(head);
            if (lane == 0) heads[w] = m;
        }
        __syncthreads();
        if (wave == 0) {
            constexpr int WPL = HW / 64;
            uint32_t c[WPL], sum = 0;
#pragma unroll
            for (int q = 0; q < WPL; ++q) {
                c[q] = (uint32_t)__popcll(heads[lane * WPL + q]);
                sum += c[q];
            }
            uint32_t inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            uint32_t run = inc - sum;
#pragma unroll
            for (int q = 0; q < WPL; ++q) {
                hpre[lane * WPL + q] = run;
                run += c[q];
            }
        }
        __syncthreads();
        for (uint32_t p = tid; p < (uint32_t)SM_NP; p += BLOCK) {
            const unsigned long long hw = heads[p >> 6];
            if (!((hw >> (p & 63)) & 1ull)) continue;
            const uint32_t rank = hpre[p >> 6] + (uint32_t)__popcll(hw & ((1ull << (p & 63)) - 1ull));
            const unsigned long long cur = sk[p];
            const uint32_t key = (uint32_t)(cur >> 32);
            double acc = sv[(uint32_t)cur];
            for (uint32_t q = p + 1; q < (uint32_t)SM_NP && (uint32_t)(sk[q] >> 32) == key; ++q) acc += sv[(uint32_t)sk[q]];
            const uint32_t lr = colbits >= 32 ? 0u : (key >> colbits);
            const uint64_t pos = s_out[lr] + (rank - s_boff[lr]);
            c_idx[pos] = key & colmask;
            c_val[pos] = acc;
        }
        __syncthreads();
    }
}

}  // namespace spada
