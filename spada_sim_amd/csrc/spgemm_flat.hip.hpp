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
constexpr int SYM2_BIN_8K = 3, SYM2_BIN_24K = 4, SYM2_BIN_SPILL = 5;
constexpr int NUM2_BIN_2K = 3, NUM2_BIN_6K = 4, NUM2_BIN_BMV = 5, NUM2_BIN_SPILL = 6;

constexpr uint32_t SYM_FLAT_CAP = 4096, SYM_FLAT_MAX = 2048;   // products per batch / per flat row
constexpr int SYM_FLAT_LOG_T = 13;                             // 8192 keys: load <= 0.75, typically 0.5

__host__ __device__ inline int sym2_bin_of(uint64_t P, uint32_t L)
{
    if (P == 0) return BIN_EMPTY;
    if (L == 1) return BIN_COPY;
    if (P <= SYM_FLAT_MAX) return BIN_FLAT;
    if (P <= 8192) return SYM2_BIN_8K;
    if (P <= 24576) return SYM2_BIN_24K;
    return SYM2_BIN_SPILL;
}
__host__ __device__ inline int num2_bin_of(uint32_t n, uint64_t P, uint32_t L, uint32_t flat_max, uint32_t vcap)
{
    if (n == 0) return BIN_EMPTY;
    if (L == 1) return BIN_COPY;
    if (n <= flat_max) return BIN_FLAT;
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

// ---- 1. row statistics + entry descriptors -------------------------------------------------------------------
// One lane per A row (rows longer than 16 nonzeros: the whole wave).  Per entry: B row begin / length (the
// irregular 16-byte gather of the path, done exactly once) and the row's first / last column, which bound the
// columns of C_i and later steer the order-preserving buckets of the numeric phase.
__global__ __launch_bounds__(256) void k_row_stats2(const uint64_t *__restrict__ aptr, const uint32_t *__restrict__ aidx,
                                                    const uint64_t *__restrict__ bptr, const uint32_t *__restrict__ bidx,
                                                    uint64_t r0, uint32_t nrows, uint64_t *__restrict__ eb0,
                                                    uint32_t *__restrict__ elen, uint32_t *__restrict__ row_nprod,
                                                    uint32_t *__restrict__ row_nnzc, uint8_t *__restrict__ row_bin,
                                                    uint32_t *__restrict__ row_kmin, uint32_t *__restrict__ row_kmax,
                                                    uint32_t *__restrict__ bin_counts,
                                                    unsigned long long *__restrict__ totals /* [0]=nprod [1]=a_nnz */,
                                                    int flat_on)
{
    __shared__ uint32_t s_hist[SPADA_N_BINS];
    __shared__ unsigned long long s_tot[2];
    if (threadIdx.x < SPADA_N_BINS) s_hist[threadIdx.x] = 0;
    if (threadIdx.x < 2) s_tot[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint64_t a0 = 0, a1 = 0, P = 0;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
    if (i < nrows) {
        a0 = aptr[r0 + i];
        a1 = aptr[r0 + i + 1];
    }
    const uint32_t L = (uint32_t)(a1 - a0);
    const bool is_long = L > 16;
    auto visit = [&](uint64_t q, uint64_t &part, uint32_t &mn, uint32_t &mx) {
        const uint32_t k = aidx[q];
        const uint64_t b0 = bptr[k], b1 = bptr[k + 1];
        eb0[q] = b0;
        elen[q] = (uint32_t)(b1 - b0);
        part += b1 - b0;
        if (b1 > b0) {
            mn = min(mn, bidx[b0]);
            mx = max(mx, bidx[b1 - 1]);
        }
    };
    if (!is_long)
        for (uint64_t q = a0; q < a1; ++q) visit(q, P, kmin, kmax);
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
        if (bin == BIN_FLAT && !flat_on) bin = SYM2_BIN_8K;
        row_nprod[i] = P > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)P;
        row_bin[i] = (uint8_t)bin;
        row_kmin[i] = kmin;
        row_kmax[i] = kmax;
        if (bin == BIN_EMPTY) row_nnzc[i] = 0;
        if (bin == BIN_COPY) row_nnzc[i] = (uint32_t)P;   // one A nonzero: C row is a scaled copy of one B row
        atomicAdd(&s_hist[bin], 1u);
    }
    uint64_t wp = wave_sum_u64(P), wl = wave_sum_u64((uint64_t)L);
    if (lane == 0) {
        atomicAdd(&s_tot[0], (unsigned long long)wp);
        atomicAdd(&s_tot[1], (unsigned long long)wl);
    }
    __syncthreads();
    if (threadIdx.x < SPADA_N_BINS && s_hist[threadIdx.x]) atomicAdd(&bin_counts[threadIdx.x], s_hist[threadIdx.x]);
    if (threadIdx.x < 2 && s_tot[threadIdx.x]) atomicAdd(&totals[threadIdx.x], s_tot[threadIdx.x]);
}

// ---- 2. scans: nnz(C_i) -> cptr, numeric classification, and batch cutting ------------------------------------
// Batches: every row carries a weight w_i (flat rows: their size, at least `minw`; all other rows: `minw`, which
// bounds the rows of a batch by cap / minw).  With S_i the exclusive prefix sum of w, row i belongs to batch
// floor(S_i / cap); a batch therefore weighs less than cap + max flat weight.  Row i announces the first row
// of the next batch when its own interval [S_i, S_i + w_i) reaches the next multiple of cap.
struct CutParams {
    uint32_t cap, minw, flat_max /* numeric */, vcap /* numeric */;
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
        bin = num2_bin_of(nnz, row_nprod[i], L, cp.flat_max, cp.vcap);
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
                                                          uint32_t *__restrict__ batch_first)
{
    __shared__ uint64_t s_w[SCAN_BLOCK / 64];
    __shared__ uint32_t s_hist[SPADA_N_BINS];
    if (threadIdx.x < SPADA_N_BINS) s_hist[threadIdx.x] = 0;
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
    }
}

// scatter the rows of the per-row bins (everything except empty / flat) into their lists
__global__ __launch_bounds__(256) void k_bin_scatter2(const uint8_t *__restrict__ row_bin, uint32_t nrows,
                                                      const uint32_t *__restrict__ bin_counts,
                                                      uint32_t *__restrict__ bin_cursor, uint32_t *__restrict__ bin_rows)
{
    __shared__ uint32_t s_cnt[SPADA_N_BINS], s_base[SPADA_N_BINS];
    if (threadIdx.x < SPADA_N_BINS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    int bin = -1;
    uint32_t local = 0;
    if (i < nrows) {
        bin = row_bin[i];
        if (bin == BIN_EMPTY || bin == BIN_FLAT) bin = -1;
        else local = atomicAdd(&s_cnt[bin], 1u);
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
    if (bin >= 0) bin_rows[s_base[bin] + local] = i;
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

// ---- 4. symbolic, flat batches ---------------------------------------------------------------------------------
// LDS: 128 B hdr | keys u32[T] | s_re u32[RMAX + 1] | s_cnt u32[RMAX] | s_a0 u64[RMAX] | w_b0 u64[BLOCK] |
//      w_off u32[BLOCK + 1] | w_lr u32[BLOCK]
template <int BLOCK, int LOG_T, int RMAX>
__host__ __device__ constexpr size_t sym_flat_lds()
{
    return 128 + ((size_t)4 << LOG_T) + (size_t)(RMAX + 1) * 4 + (size_t)RMAX * 4 + (size_t)RMAX * 8 + (size_t)BLOCK * 8 +
           (size_t)(BLOCK + 1) * 4 + (size_t)BLOCK * 4 + 32;
}

template <int BLOCK, int LOG_T, int RMAX>
__global__ __launch_bounds__(BLOCK) void k_sym_flat(const uint64_t *__restrict__ aptr, const uint32_t *__restrict__ bidx,
                                                    const uint64_t *__restrict__ eb0, const uint32_t *__restrict__ elen,
                                                    uint64_t r0, uint32_t nrows, const uint8_t *__restrict__ row_bin,
                                                    const uint32_t *__restrict__ batch_first, const uint32_t *__restrict__ nb_ptr,
                                                    uint32_t colbits, uint32_t *__restrict__ row_nnzc)
{
    static_assert(RMAX <= BLOCK, "one thread per row of a batch");
    constexpr int T = 1 << LOG_T;
    constexpr int U = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    uint32_t *keys = (uint32_t *)(smem + 128);
    uint32_t *s_re = keys + T;
    uint32_t *s_cnt = s_re + RMAX + 1;
    uint64_t *s_a0 = (uint64_t *)(((uintptr_t)(s_cnt + RMAX) + 7) & ~(uintptr_t)7);
    uint64_t *w_b0 = s_a0 + RMAX;
    uint32_t *w_off = (uint32_t *)(w_b0 + BLOCK);
    uint32_t *w_lr = w_off + BLOCK + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t nb = *nb_ptr;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const uint32_t rb = batch_first[b];
        const uint32_t re = b + 1 < nb ? batch_first[b + 1] : nrows;
        const uint32_t R = re - rb;   // <= RMAX by construction of the cut
        if (R == 0) continue;
        // rows of the batch: entry prefix of the flat ones
        uint32_t L = 0;
        if ((uint32_t)tid < R) {
            const uint64_t a0 = aptr[r0 + rb + tid], a1 = aptr[r0 + rb + tid + 1];
            s_a0[tid] = a0;
            s_cnt[tid] = 0;
            if (row_bin[rb + tid] == BIN_FLAT) L = (uint32_t)(a1 - a0);
        }
        uint32_t E;
        const uint32_t ex = group_scan_excl<BLOCK>(L, tid, hdr + 2, &E);
        if ((uint32_t)tid < R) s_re[tid] = ex;
        if (tid == 0) s_re[R] = E;
        for (int s = tid; s < T; s += BLOCK) keys[s] = EMPTY_KEY;
        __syncthreads();
        for (uint32_t chunk = 0; chunk < E; chunk += BLOCK) {
            const uint32_t e = chunk + tid;
            uint64_t b0 = 0;
            uint32_t len = 0, lr = 0;
            if (e < E) {
                lr = row_of_entry<RMAX>(s_re, R, e);
                const uint64_t a = s_a0[lr] + (e - s_re[lr]);
                b0 = eb0[a];
                len = elen[a];
            }
            uint32_t total;
            const uint32_t off = group_scan_excl<BLOCK>(len, tid, hdr + 2, &total);
            w_b0[tid] = b0;
            w_off[tid] = off;
            w_lr[tid] = lr;
            if (tid == BLOCK - 1) w_off[BLOCK] = total;
            __syncthreads();
            for (uint32_t base = 0; base < total; base += U * BLOCK) {
                uint32_t key[U], plr[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t p = base + u * BLOCK + tid;
                    key[u] = EMPTY_KEY;
                    plr[u] = LR_NONE;
                    if (p < total) {
                        int j = 0;
#pragma unroll
                        for (int step = BLOCK / 2; step >= 1; step >>= 1)
                            if (w_off[j + step] <= p) j += step;
                        const uint32_t c = bidx[w_b0[j] + (p - w_off[j])];
                        plr[u] = w_lr[j];
                        key[u] = (plr[u] << colbits) | c;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    bool isnew = false;
                    if (key[u] != EMPTY_KEY) {
                        uint32_t h = hash_slot<LOG_T>(key[u]);
                        for (;;) {
                            const uint32_t old = atomicCAS(&keys[h], EMPTY_KEY, key[u]);
                            if (old == EMPTY_KEY) { isnew = true; break; }
                            if (old == key[u]) break;
                            h = (h + 1) & (T - 1);
                        }
                    }
                    segmented_count_add(plr[u], isnew, s_cnt, lane);
                }
            }
            __syncthreads();
        }
        if ((uint32_t)tid < R && row_bin[rb + tid] == BIN_FLAT) row_nnzc[rb + tid] = s_cnt[tid];
        __syncthreads();
    }
}

// ---- 5. numeric, flat batches -----------------------------------------------------------------------------------
// LDS: 128 B hdr | table: keys u32[T], vals f64[T]  (re-used after accumulation as lk u32[NOUT], lv f64[NOUT])
//      | bcnt u32[NOUT]  (the walk scratch w_b0 / w_av / w_off / w_lr aliases it: disjoint phases)
//      | rows: s_re u32[RMAX+1], s_boff u32[RMAX+1], s_kmin u32[RMAX], s_scale f32[RMAX], s_a0 u64[RMAX], s_out u64[RMAX]
template <int BLOCK, int LOG_T, int NOUT, int RMAX>
__host__ __device__ constexpr size_t num_flat_lds()
{
    constexpr size_t walk = (size_t)BLOCK * 8 * 2 + (size_t)(BLOCK + 1) * 4 + (size_t)BLOCK * 4 + 16;
    constexpr size_t bc = (size_t)NOUT * 4;
    return 128 + ((size_t)12 << LOG_T) + (bc > walk ? bc : walk) + (size_t)(RMAX + 1) * 8 + (size_t)RMAX * 8 + (size_t)RMAX * 16 + 32;
}

template <int BLOCK, int LOG_T, int NOUT, int RMAX>
__global__ __launch_bounds__(BLOCK) void k_num_flat(const uint64_t *__restrict__ aptr, const double *__restrict__ aval,
                                                    const uint32_t *__restrict__ bidx, const double *__restrict__ bval,
                                                    const uint64_t *__restrict__ eb0, const uint32_t *__restrict__ elen,
                                                    uint64_t r0, uint32_t nrows, const uint8_t *__restrict__ row_bin,
                                                    const uint32_t *__restrict__ row_kmin, const uint32_t *__restrict__ row_kmax,
                                                    const uint64_t *__restrict__ cptr, const uint32_t *__restrict__ batch_first,
                                                    const uint32_t *__restrict__ nb_ptr, uint32_t colbits,
                                                    uint32_t *__restrict__ c_idx, double *__restrict__ c_val)
{
    static_assert(RMAX <= BLOCK, "one thread per row of a batch");
    static_assert(NOUT % BLOCK == 0, "flat scan length");
    static_assert((size_t)NOUT * 12 <= ((size_t)12 << LOG_T), "compacted (key, value) lists re-use the table");
    constexpr int T = 1 << LOG_T;
    constexpr int SPT = T / BLOCK;   // table slots per thread
    constexpr int U = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    uint32_t *keys = (uint32_t *)(smem + 128);
    double *vals = (double *)(keys + T);
    uint32_t *lk = keys;                       // after accumulation: keys in bucket order
    double *lv = (double *)(smem + 128 + (((size_t)NOUT * 4 + 7) & ~(size_t)7));   // and their values
    unsigned char *region2 = smem + 128 + ((size_t)12 << LOG_T);
    uint32_t *bcnt = (uint32_t *)region2;
    uint64_t *w_b0 = (uint64_t *)region2;
    double *w_av = (double *)(w_b0 + BLOCK);
    uint32_t *w_off = (uint32_t *)(w_av + BLOCK);
    uint32_t *w_lr = w_off + BLOCK + 1;
    constexpr size_t walk = (size_t)BLOCK * 8 * 2 + (size_t)(BLOCK + 1) * 4 + (size_t)BLOCK * 4 + 16;
    constexpr size_t r2 = ((size_t)NOUT * 4 > walk ? (size_t)NOUT * 4 : walk);
    uint32_t *s_re = (uint32_t *)(region2 + ((r2 + 7) & ~(size_t)7));
    uint32_t *s_boff = s_re + RMAX + 1;
    uint32_t *s_kmin = s_boff + RMAX + 1;
    float *s_scale = (float *)(s_kmin + RMAX);
    uint64_t *s_a0 = (uint64_t *)(((uintptr_t)(s_scale + RMAX) + 7) & ~(uintptr_t)7);
    uint64_t *s_out = s_a0 + RMAX;
    const int tid = threadIdx.x;
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
                const uint32_t kmin = row_kmin[rb + tid], kmax = row_kmax[rb + tid];
                s_kmin[tid] = kmin;
                s_scale[tid] = (float)n / ((float)(kmax - kmin) + 1.0f);
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
        if (tid == 0) {
            s_re[R] = E;
            s_boff[R] = NO;
        }
        for (int s = tid; s < T; s += BLOCK) {
            keys[s] = EMPTY_KEY;
            vals[s] = 0.0;
        }
        __syncthreads();

        // ---- expand - scale - accumulate --------------------------------------------------------------------
        for (uint32_t chunk = 0; chunk < E; chunk += BLOCK) {
            const uint32_t e = chunk + tid;
            uint64_t b0 = 0;
            uint32_t len = 0, lr = 0;
            double av = 0.0;
            if (e < E) {
                lr = row_of_entry<RMAX>(s_re, R, e);
                const uint64_t a = s_a0[lr] + (e - s_re[lr]);
                b0 = eb0[a];
                len = elen[a];
                av = aval[a];
            }
            uint32_t total;
            const uint32_t off = group_scan_excl<BLOCK>(len, tid, hdr + 2, &total);
            w_b0[tid] = b0;
            w_av[tid] = av;
            w_off[tid] = off;
            w_lr[tid] = lr << colbits;
            if (tid == BLOCK - 1) w_off[BLOCK] = total;
            __syncthreads();
            for (uint32_t p0 = tid; p0 < total; p0 += U * BLOCK) {
                uint32_t key[U];
                double v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t p = p0 + u * BLOCK;
                    key[u] = EMPTY_KEY;
                    v[u] = 0.0;
                    if (p < total) {
                        int j = 0;
#pragma unroll
                        for (int step = BLOCK / 2; step >= 1; step >>= 1)
                            if (w_off[j + step] <= p) j += step;
                        const uint64_t q = w_b0[j] + (p - w_off[j]);
                        key[u] = w_lr[j] | bidx[q];
                        v[u] = w_av[j] * bval[q];   // simulator.rs:100-101
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (key[u] != EMPTY_KEY) {
                        uint32_t h = hash_slot<LOG_T>(key[u]);
                        for (;;) {
                            const uint32_t old = atomicCAS(&keys[h], EMPTY_KEY, key[u]);
                            if (old == EMPTY_KEY || old == key[u]) break;
                            h = (h + 1) & (T - 1);
                        }
                        atomicAdd(&vals[h], v[u]);   // simulator.rs:213-218 (order differs)
                    }
            }
            __syncthreads();
        }

        // ---- ordered emission -------------------------------------------------------------------------------
        // every occupied slot -> bucket = s_boff[lr] + floor((col - kmin) * n / span): monotone inside a row and
        // rows are laid out in order, so the bucket order IS the order of the batch's slice of C up to
        // permutations inside one bucket
        for (int s = tid; s < NOUT; s += BLOCK) bcnt[s] = 0;
        __syncthreads();
        uint32_t myk[SPT];
        uint16_t myb[SPT];
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const uint32_t k = keys[tid + i * BLOCK];
            myk[i] = k;
            myb[i] = 0;
            if (k != EMPTY_KEY) {
                const uint32_t lr = colbits >= 32 ? 0u : (k >> colbits), col = k & colmask;
                const uint32_t nr = s_boff[lr + 1] - s_boff[lr];
                uint32_t bk = (uint32_t)((float)(col - s_kmin[lr]) * s_scale[lr]);
                bk = bk < nr ? bk : nr - 1;
                myb[i] = (uint16_t)(s_boff[lr] + bk);
                atomicAdd(&bcnt[myb[i]], 1u);
            }
        }
        double myv[SPT];
#pragma unroll
        for (int i = 0; i < SPT; ++i) myv[i] = vals[tid + i * BLOCK];
        __syncthreads();
        group_exclusive_scan<BLOCK, NOUT>(bcnt, tid, hdr + 2);   // ends with a barrier: table fully read by now
#pragma unroll
        for (int i = 0; i < SPT; ++i)
            if (myk[i] != EMPTY_KEY) {
                const uint32_t p = atomicAdd(&bcnt[myb[i]], 1u);   // afterwards bcnt[b] = end of bucket b
                lk[p] = myk[i];
                lv[p] = myv[i];
            }
        __syncthreads();
        for (uint32_t p = tid; p < NO; p += BLOCK) {
            const uint32_t k = lk[p];
            const uint32_t lr = colbits >= 32 ? 0u : (k >> colbits), col = k & colmask;
            const uint32_t nr = s_boff[lr + 1] - s_boff[lr];
            uint32_t bk = (uint32_t)((float)(col - s_kmin[lr]) * s_scale[lr]);
            bk = s_boff[lr] + (bk < nr ? bk : nr - 1);
            const uint32_t lo = bk ? bcnt[bk - 1] : 0u, hi = bcnt[bk];
            uint32_t r = lo;
            for (uint32_t j = lo; j < hi; ++j) r += (lk[j] < k) ? 1u : 0u;
            const uint64_t pos = s_out[lr] + (r - s_boff[lr]);
            c_idx[pos] = col;
            c_val[pos] = lv[p];
        }
        __syncthreads();
    }
}

}  // namespace spada
