// gfx950 (MI355X, CDNA4) kernels of the row-wise expand-scale-merge SpGEMM  C = A * B  on CSR, part 1: helpers and the
// one-workgroup-per-row kernels for LARGE rows (LDS hash: k_sym_hash / k_num_hash) and HUGE rows (column bitmap in
// LDS: k_*_bitmap, or in HBM: k_*_spill).  Small and mid rows -- the bulk -- take the flat-batch kernels of
// spgemm_flat.hip.hpp, which also holds the row statistics, the scans and the bin definitions.
//
// Together they replace what spada-sim simulates cycle by cycle (citations into /root/reference/src): window
// fetch of A scalars (scheduler.rs:482-606, storage.rs:279-323), B-fiber streaming (simulator.rs:892-953), the
// multiplier array (simulator.rs:86-111), sorting network + merge tree (simulator.rs:143-230), psum write-back /
// partial-fiber merging (simulator.rs:955-983, scheduler.rs:381-480, adder_tree.rs:145-188) and result assembly
// (simulator.rs:1034-1062).
//   * 64-wide wavefronts; per-row accumulators live in LDS: open-addressing hash (u32 key, f64 value), ds_cmpst for
//     the key, ds_add_f64 for the value; column compaction + ordering by a monotone bucket pass with LDS counters,
//     a group-wide scan and an in-bucket rank, then ascending stores to C.
//   * rows too large for an LDS table: column bitmap (LDS up to ~1.1 M columns, else HBM), prefix popcounts give
//     every column its final position, values by f64 atomics.
//   * no MFMA: this is irregular gather-accumulate, bounded by memory latency, LDS throughput and HBM bandwidth.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spada_ffi.h"

namespace spada {

constexpr uint32_t EMPTY_KEY = 0xFFFFFFFFu;
constexpr int WAVE = 64;

struct DevCsrView {
    const uint64_t *ptr;
    const uint32_t *idx;
    const double *val;
    // A only: per-entry descriptor of the selected B row (begin, length), written by k_row_stats2
    const uint64_t *eb0;
    const uint32_t *elen;
};

// bins: see spgemm_flat.hip.hpp (sym2_bin_of / num2_bin_of)
// ---- small helpers -----------------------------------------------------------------------------------
__device__ inline uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Synchronise the G lanes that share one row.  G >= 128 means the group is the whole workgroup.
// For G <= 64 the group lives inside one wavefront, which executes in lock step; only the LDS
// traffic has to be ordered.
template <int G>
__device__ inline void group_sync()
{
    if constexpr (G >= 128) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// Sum over the group.  `hdr` is a per-workgroup LDS word used when the group spans several waves.
template <int G>
__device__ inline uint32_t group_sum(uint32_t v, uint32_t *hdr)
{
    if constexpr (G <= 64) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    } else {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if (threadIdx.x == 0) *hdr = 0;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) atomicAdd(hdr, v);
        __syncthreads();
        return *hdr;
    }
}
template <int G>
__device__ inline uint32_t group_min(uint32_t v, uint32_t *hdr)
{
    if constexpr (G <= 64) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor(v, o));
        return v;
    } else {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor(v, o));
        __syncthreads();
        if (threadIdx.x == 0) *hdr = 0xFFFFFFFFu;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) atomicMin(hdr, v);
        __syncthreads();
        return *hdr;
    }
}
template <int G>
__device__ inline uint32_t group_max(uint32_t v, uint32_t *hdr)
{
    if constexpr (G <= 64) {
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor(v, o));
        return v;
    } else {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor(v, o));
        __syncthreads();
        if (threadIdx.x == 0) *hdr = 0;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) atomicMax(hdr, v);
        __syncthreads();
        return *hdr;
    }
}

// In-place exclusive scan of arr[0..N) by the G lanes of a group; N is a multiple of G.
// `wtot` = LDS scratch for per-wave totals (>= G/64 words), only used when G > 64.
template <int G, int N>
__device__ inline void group_exclusive_scan(uint32_t *arr, int gl, uint32_t *wtot)
{
    constexpr int PER = N / G;
    static_assert(N % G == 0, "scan length must be a multiple of the group size");
    uint32_t loc[PER];
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        loc[j] = arr[gl * PER + j];
        tot += loc[j];
    }
    // inclusive scan of `tot` across the lanes of the group
    uint32_t inc = tot;
    constexpr int W = G < 64 ? G : 64;
    const int wl = (G < 64) ? gl : (gl & 63);
#pragma unroll
    for (int o = 1; o < W; o <<= 1) {
        uint32_t t = __shfl_up(inc, o, W);
        if (wl >= o) inc += t;
    }
    uint32_t base = inc - tot;
    if constexpr (G > 64) {
        const int w = gl >> 6;
        __syncthreads();
        if (wl == 63) wtot[w] = inc;
        __syncthreads();
        uint32_t add = 0;
        for (int k = 0; k < w; ++k) add += wtot[k];
        base += add;
    }
    group_sync<G>();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        arr[gl * PER + j] = base;
        base += loc[j];
    }
    group_sync<G>();
}

template <int LOG_T>
__device__ inline uint32_t hash_slot(uint32_t col)
{
    return (col * 0x9E3779B1u) >> (32 - LOG_T);
}

// ---- 2. block-wide scan helpers (the scan kernels themselves: k_cut_* in spgemm_flat.hip.hpp) ---------------
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ inline uint64_t block_exclusive_scan_u64(uint64_t v, uint64_t *s_w /*[4]*/, uint64_t *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint64_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    uint64_t add = 0, tot = 0;
    for (int k = 0; k < SCAN_BLOCK / 64; ++k) {
        if (k < w) add += s_w[k];
        tot += s_w[k];
    }
    *total = tot;
    return inc - v + add;
}

// ---- 3. balanced walk over the products of one A row --------------------------------------------------------
// The G lanes of a group take the row's A nonzeros G at a time: lane j loads (k, a_ik, B.ptr[k], nnz(B_k)),
// an exclusive scan of the B row lengths gives every product of the chunk a dense index p, and the lanes
// then stride over p.  The owner of p is found by a branch-free binary search in LDS.  Adjacent lanes read
// adjacent B entries whenever they fall into the same B row (coalesced 4 B / 8 B loads), every lane has
// the same number of products whatever the row-length skew, and four independent gathers per lane are in
// flight before the first LDS atomic.
// LDS scratch per group: b0 u64[G] | av f64[G] (numeric only) | off u32[G + 2].
template <int G, bool NUMERIC>
__host__ __device__ constexpr size_t walk_scratch_bytes()
{
    return (size_t)G * 8 + (NUMERIC ? (size_t)G * 8 : 0) + ((size_t)G + 2) * 4;
}

// exclusive scan of one u32 per lane across the group; *total = group sum
template <int G>
__device__ inline uint32_t group_scan_excl(uint32_t v, int gl, uint32_t *wtot, uint32_t *total)
{
    constexpr int W = G < 64 ? G : 64;
    const int wl = (G < 64) ? gl : (gl & 63);
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < W; o <<= 1) {
        uint32_t t = __shfl_up(inc, o, W);
        if (wl >= o) inc += t;
    }
    if constexpr (G <= 64) {
        *total = __shfl(inc, W - 1, W);
        return inc - v;
    } else {
        const int w = gl >> 6;
        __syncthreads();
        if (wl == 63) wtot[w] = inc;
        __syncthreads();
        uint32_t add = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < G / 64; ++k) {
            const uint32_t t = wtot[k];
            if (k < w) add += t;
            tot += t;
        }
        *total = tot;
        return inc - v + add;
    }
}

// exclusive scan of one u64 per thread across a workgroup of G threads (G a multiple of 64); *total = sum
template <int G>
__device__ inline unsigned long long group_scan_excl_u64(unsigned long long v, int gl, unsigned long long *wtot,
                                                         unsigned long long *total)
{
    static_assert(G % 64 == 0 && G >= 64, "whole waves");
    const int wl = gl & 63, w = gl >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(inc, o);
        if (wl >= o) inc += t;
    }
    __syncthreads();
    if (wl == 63) wtot[w] = inc;
    __syncthreads();
    unsigned long long add = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < G / 64; ++k) {
        const unsigned long long t = wtot[k];
        if (k < w) add += t;
        tot += t;
    }
    *total = tot;
    return inc - v + add;
}

template <int G, bool NUMERIC, class F>
__device__ inline void walk_products(const DevCsrView &A, const DevCsrView &B, uint64_t a0, uint64_t a1, int gl,
                                     unsigned char *scratch, uint32_t *hdr, F &&f)
{
    uint64_t *s_b0 = (uint64_t *)scratch;
    double *s_av = (double *)(scratch + (size_t)G * 8);
    uint32_t *s_off = (uint32_t *)(scratch + (size_t)G * 8 + (NUMERIC ? (size_t)G * 8 : 0));
    constexpr int U = 4;
    for (uint64_t base = a0; base < a1; base += G) {
        const uint64_t a = base + gl;
        uint64_t b0 = 0;
        uint32_t len = 0;
        double av = 0.0;
        if (a < a1) {
            if constexpr (NUMERIC) av = A.val[a];
            b0 = A.eb0[a];
            len = A.elen[a];
        }
        const uint32_t maxlen = group_max<G>(len, hdr);
        uint32_t total;
        const uint32_t off = group_scan_excl<G>(len, gl, hdr + 2, &total);
        s_b0[gl] = b0;
        if constexpr (NUMERIC) s_av[gl] = av;
        s_off[gl] = off;
        if (gl == G - 1) s_off[G] = total;
        group_sync<G>();
        if (maxlen < (0xFFFFFFFFu / G)) {
            for (uint32_t p0 = gl; p0 < total; p0 += U * G) {
                uint32_t c[U], pp[U];
                double v[U];
                int j[U];
                // the U owner searches advance in lock step (U independent LDS reads per round), then the U gathers
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t p = p0 + u * G;
                    pp[u] = p < total ? p : total - 1;
                    j[u] = 0;
                }
#pragma unroll
                for (int step = G / 2; step >= 1; step >>= 1) {
                    uint32_t o[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) o[u] = s_off[j[u] + step];
#pragma unroll
                    for (int u = 0; u < U; ++u) j[u] += o[u] <= pp[u] ? step : 0;
                }
                uint64_t q[U];
#pragma unroll
                for (int u = 0; u < U; ++u) q[u] = s_b0[j[u]] + (pp[u] - s_off[j[u]]);
#pragma unroll
                for (int u = 0; u < U; ++u) c[u] = B.idx[q[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    v[u] = 0.0;
                    if constexpr (NUMERIC) v[u] = s_av[j[u]] * B.val[q[u]];   // simulator.rs:100-101
                    if (p0 + u * G >= total) c[u] = EMPTY_KEY;
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (c[u] != EMPTY_KEY) f(c[u], v[u]);
            }
        } else {
            // a B row so long that 32-bit product indices of the chunk could wrap: one B row at a time
            const int cnt = (int)((a1 - base) < (uint64_t)G ? (a1 - base) : (uint64_t)G);
            for (int j = 0; j < cnt; ++j) {
                const uint64_t jb0 = s_b0[j];
                const uint32_t jlen = s_off[j + 1] - s_off[j];   // exact modulo 2^32
                for (uint32_t t = gl; t < jlen; t += G) {
                    double v = 0.0;
                    if constexpr (NUMERIC) v = s_av[j] * B.val[jb0 + t];
                    f(B.idx[jb0 + t], v);
                }
            }
        }
        group_sync<G>();
    }
}

// ---- 4. symbolic: LDS hash set per row -------------------------------------------------------------------
// Group of G lanes per row, T-entry key table per row.  LDS: 128 B header | per row: keys u32[T] | walk scratch.
template <int G, int LOG_T>
__host__ __device__ constexpr size_t sym_row_bytes()
{
    return (((size_t)4 << LOG_T) + walk_scratch_bytes<G, false>() + 15) & ~(size_t)15;
}

template <int G, int LOG_T>
__global__ __launch_bounds__((G <= 64 ? 256 : G)) void k_sym_hash(DevCsrView A, DevCsrView B, uint64_t r0,
                                                                   const uint32_t *__restrict__ bin_rows, uint32_t n_bin_rows,
                                                                   uint32_t *__restrict__ row_nnzc)
{
    constexpr int T = 1 << LOG_T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;   // 128 B header: [0..1] reductions, [2..17] wave totals
    unsigned char *mine = smem + 128 + (size_t)(threadIdx.x / G) * sym_row_bytes<G, LOG_T>();
    uint32_t *keys = (uint32_t *)mine;
    const int gl = threadIdx.x % G;
    const uint32_t slot = blockIdx.x * ((G <= 64 ? 256 : G) / G) + threadIdx.x / G;
    const bool active = slot < n_bin_rows;
    for (int s = gl; s < T; s += G) keys[s] = EMPTY_KEY;
    group_sync<G>();
    uint32_t cnt = 0, row = 0;
    if (active) {
        row = bin_rows[slot];
        const uint64_t a0 = A.ptr[r0 + row], a1 = A.ptr[r0 + row + 1];
        walk_products<G, false>(A, B, a0, a1, gl, mine + ((size_t)4 << LOG_T), hdr, [&](uint32_t c, double) {
            uint32_t h = hash_slot<LOG_T>(c);
            for (;;) {
                const uint32_t old = atomicCAS(&keys[h], EMPTY_KEY, c);
                if (old == EMPTY_KEY) { ++cnt; break; }
                if (old == c) break;
                h = (h + 1) & (T - 1);
            }
        });
    }
    cnt = group_sum<G>(cnt, hdr);
    if (active && gl == 0) row_nnzc[row] = cnt;
}

// ---- 5. numeric: LDS hash accumulator + ordered emission ---------------------------------------------------
// LDS per row: keys u32[T] | vals f64[T] | cnt u32[T/2] | list u16[T] | walk scratch.
template <int G, int LOG_T>
__host__ __device__ constexpr size_t num_row_bytes()
{
    return (((size_t)16 << LOG_T) + walk_scratch_bytes<G, true>() + 15) & ~(size_t)15;
}

template <int G, int LOG_T>
__global__ __launch_bounds__((G <= 64 ? 256 : G)) void k_num_hash(DevCsrView A, DevCsrView B, uint64_t r0,
                                                                   const uint32_t *__restrict__ bin_rows, uint32_t n_bin_rows,
                                                                   const uint64_t *__restrict__ cptr,
                                                                   uint32_t *__restrict__ c_idx, double *__restrict__ c_val,
                                                                   unsigned long long *dbg,
                                                                   const uint32_t *__restrict__ row_kmin,
                                                                   const uint32_t *__restrict__ row_kmax)
{
#define STAMP(i) do { if (dbg && threadIdx.x == 0 && blockIdx.x % 16 == 0 && blockIdx.x / 16 < 64) dbg[(blockIdx.x / 16) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    STAMP(0);
    constexpr int T = 1 << LOG_T;
    constexpr int NB = T / 2;
    constexpr int BLOCK = G <= 64 ? 256 : G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;   // 128 B: [0..1] reductions, [2..17] wave totals
    unsigned char *mine = smem + 128 + (size_t)(threadIdx.x / G) * num_row_bytes<G, LOG_T>();
    uint32_t *keys = (uint32_t *)mine;
    double *vals = (double *)(mine + 4 * T);
    uint32_t *cnt = (uint32_t *)(mine + 12 * T);
    uint16_t *list = (uint16_t *)(mine + 14 * T);
    const int gl = threadIdx.x % G;
    const uint32_t slot = blockIdx.x * (BLOCK / G) + threadIdx.x / G;
    const bool active = slot < n_bin_rows;

    for (int s = gl; s < T; s += G) {
        keys[s] = EMPTY_KEY;
        vals[s] = 0.0;
    }
    for (int s = gl; s < NB; s += G) cnt[s] = 0;
    group_sync<G>();
    STAMP(1);

    uint32_t row = 0;
    if (active) {
        row = bin_rows[slot];
        const uint64_t a0 = A.ptr[r0 + row], a1 = A.ptr[r0 + row + 1];
        walk_products<G, true>(A, B, a0, a1, gl, mine + 16 * T, hdr, [&](uint32_t c, double v) {
            uint32_t h = hash_slot<LOG_T>(c);
            for (;;) {
                const uint32_t old = atomicCAS(&keys[h], EMPTY_KEY, c);
                if (old == EMPTY_KEY || old == c) break;
                h = (h + 1) & (T - 1);
            }
            atomicAdd(&vals[h], v);   // simulator.rs:213-218 (order differs)
        });
    }
    group_sync<G>();
    STAMP(2);

    // ---- ordered emission: monotone buckets over [kmin, kmax], scan, in-bucket rank ---------------------
    // each lane keeps its T/G table keys in registers for the three passes over the table
    constexpr int SPL = T / G;
    uint32_t myk[SPL];
#pragma unroll
    for (int i = 0; i < SPL; ++i) myk[i] = keys[gl + i * G];
    // smallest / largest column of the row = first / last column of the selected B rows (k_row_stats2): no reduction
    const uint32_t kmin = active ? row_kmin[row] : 0xFFFFFFFFu, kmax = active ? row_kmax[row] : 0u;
    STAMP(3);
    // an inactive group has an empty table (kmin > kmax)
    const float scale = (kmax >= kmin) ? (float)NB / ((float)(kmax - kmin) + 1.0f) : 0.0f;
    auto bucket = [&](uint32_t k) -> uint32_t {
        uint32_t b = (uint32_t)((float)(k - kmin) * scale);
        return b < (uint32_t)NB ? b : (uint32_t)NB - 1;
    };
#pragma unroll
    for (int i = 0; i < SPL; ++i)
        if (myk[i] != EMPTY_KEY) atomicAdd(&cnt[bucket(myk[i])], 1u);
    group_sync<G>();
    STAMP(4);
    group_exclusive_scan<G, NB>(cnt, gl, hdr + 2);
    STAMP(5);
#pragma unroll
    for (int i = 0; i < SPL; ++i)
        if (myk[i] != EMPTY_KEY) {
            const uint32_t p = atomicAdd(&cnt[bucket(myk[i])], 1u);   // afterwards cnt[b] = end of bucket b
            list[p] = (uint16_t)(gl + i * G);
        }
    group_sync<G>();
    STAMP(6);
    if (active) {
        const uint64_t c0 = cptr[row];
        const uint32_t n = (uint32_t)(cptr[row + 1] - c0);
        for (uint32_t p = gl; p < n; p += G) {
            const uint32_t s = list[p];
            const uint32_t k = keys[s];
            const uint32_t b = bucket(k);
            const uint32_t lo = b ? cnt[b - 1] : 0u, hi = cnt[b];
            uint32_t r = lo;
            for (uint32_t j = lo; j < hi; ++j) r += (keys[list[j]] < k) ? 1u : 0u;
            c_idx[c0 + r] = k;
            c_val[c0 + r] = vals[s];
        }
    }
    STAMP(7);
#undef STAMP
}

// ---- 7. spill path: rows whose accumulator does not fit LDS ---------------------------------------------------
// One persistent workgroup of 1024 lanes per row.  Per workgroup, in HBM: a column bitmap (1 bit per column
// of B) and a per-word prefix-popcount array; both are tiny next to 288 GB and stay L2 resident.  The bitmap
// gives the row's pattern in ascending order, its prefix popcounts give every column its final position in
// the C row, and the values are then accumulated with device-scope f64 atomics directly into that (compact,
// L2-resident) C row -- no dense accumulator, no sort, two walks over the row's products.
// Words are only ever modified by L2 atomics or by plain stores of this workgroup that are fenced before the
// next atomic phase; plain loads happen after an agent-scope acquire, so no stale L1 line is ever read.
constexpr int SPILL_BLOCK = 1024;
constexpr size_t SPILL_LDS = 128 + walk_scratch_bytes<SPILL_BLOCK, true>();

// This workgroup's plain stores are acknowledged by its XCD's L2 (the vector L1 is write-through) once vmcnt
// drains; the L2 atomics that follow hit the same L2, so no L2 write-back (`buffer_wbl2`, what __threadfence()
// would add, flushing every dirty line of the XCD while other workgroups stream C through it) is needed.
__device__ inline void stores_to_l2() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// pass A of both phases: set the bit of every product column; returns the touched word range [lo, hi]
__device__ inline void spill_mark(const DevCsrView &A, const DevCsrView &B, uint64_t a0, uint64_t a1,
                                  unsigned char *smem, uint32_t *hdr, uint32_t *bm, uint32_t *lo, uint32_t *hi)
{
    uint32_t wlo = 0xFFFFFFFFu, whi = 0;
    walk_products<SPILL_BLOCK, false>(A, B, a0, a1, threadIdx.x, smem + 128, hdr, [&](uint32_t c, double) {
        atomicOr(&bm[c >> 5], 1u << (c & 31));
        wlo = min(wlo, c >> 5);
        whi = max(whi, c >> 5);
    });
    *lo = group_min<SPILL_BLOCK>(wlo, hdr);
    *hi = group_max<SPILL_BLOCK>(whi, hdr + 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop L1 lines of bm that predate the atomics
    __syncthreads();
}

__global__ __launch_bounds__(SPILL_BLOCK) void k_sym_spill(DevCsrView A, DevCsrView B, uint64_t r0,
                                                           const uint32_t *__restrict__ bin_rows, uint32_t n_bin_rows,
                                                           uint32_t *__restrict__ bitmaps, uint64_t words_per_slab,
                                                           uint32_t *__restrict__ row_nnzc)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[SPILL_LDS];
    uint32_t *hdr = (uint32_t *)smem;
    uint32_t *bm = bitmaps + (uint64_t)blockIdx.x * words_per_slab;
    for (uint32_t slot = blockIdx.x; slot < n_bin_rows; slot += gridDim.x) {
        const uint32_t row = bin_rows[slot];
        const uint64_t a0 = A.ptr[r0 + row], a1 = A.ptr[r0 + row + 1];
        uint32_t lo, hi;
        spill_mark(A, B, a0, a1, smem, hdr, bm, &lo, &hi);
        uint32_t cnt = 0;
        if (lo != 0xFFFFFFFFu)
            for (uint32_t w = lo + threadIdx.x; w <= hi; w += SPILL_BLOCK) {
                cnt += __popc(bm[w]);
                bm[w] = 0;
            }
        cnt = group_sum<SPILL_BLOCK>(cnt, hdr);
        if (threadIdx.x == 0) row_nnzc[row] = cnt;
        stores_to_l2();    // the clears reach L2 before the next row's atomics
        __syncthreads();
    }
}

__global__ __launch_bounds__(SPILL_BLOCK) void k_num_spill(DevCsrView A, DevCsrView B, uint64_t r0,
                                                           const uint32_t *__restrict__ bin_rows, uint32_t n_bin_rows,
                                                           uint32_t *__restrict__ bitmaps, uint32_t *__restrict__ prefixes,
                                                           uint64_t words_per_slab, const uint64_t *__restrict__ cptr,
                                                           uint32_t *__restrict__ c_idx, double *__restrict__ c_val)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[SPILL_LDS];
    uint32_t *hdr = (uint32_t *)smem;    // [0..1] reductions, [2..17] wave totals
    uint32_t *bm = bitmaps + (uint64_t)blockIdx.x * words_per_slab;
    uint32_t *pre = prefixes + (uint64_t)blockIdx.x * words_per_slab;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int NW = SPILL_BLOCK / 64;
    for (uint32_t slot = blockIdx.x; slot < n_bin_rows; slot += gridDim.x) {
        const uint32_t row = bin_rows[slot];
        const uint64_t a0 = A.ptr[r0 + row], a1 = A.ptr[r0 + row + 1];
        const uint64_t c0 = cptr[row];
        const uint32_t n = (uint32_t)(cptr[row + 1] - c0);
        uint32_t lo, hi;
        spill_mark(A, B, a0, a1, smem, hdr, bm, &lo, &hi);
        // pass B: prefix popcounts over [lo, hi]; emit the column indices; zero the value row
        uint32_t running = 0;
        if (lo != 0xFFFFFFFFu) {
            for (uint32_t wbase = lo; wbase <= hi; wbase += SPILL_BLOCK) {
                const uint32_t w = wbase + threadIdx.x;
                uint32_t bits = (w <= hi) ? bm[w] : 0u;
                const uint32_t pc = __popc(bits);
                uint32_t inc = pc;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    uint32_t t = __shfl_up(inc, o);
                    if (lane >= o) inc += t;
                }
                __syncthreads();
                if (lane == 63) hdr[2 + wave] = inc;
                __syncthreads();
                uint32_t add = 0, tot = 0;
#pragma unroll
                for (int k2 = 0; k2 < NW; ++k2) {
                    const uint32_t t = hdr[2 + k2];
                    if (k2 < wave) add += t;
                    tot += t;
                }
                uint32_t out = running + add + (inc - pc);
                if (w <= hi) pre[w] = out;
                while (bits) {
                    const int b = __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    c_idx[c0 + out] = (w << 5) + (uint32_t)b;
                    ++out;
                }
                running += tot;
            }
        }
        for (uint32_t i = threadIdx.x; i < n; i += SPILL_BLOCK) c_val[c0 + i] = 0.0;
        stores_to_l2();    // zeros and prefixes are in L2 before any atomic of pass C
        __syncthreads();
        // pass C: accumulate every product at its final position
        walk_products<SPILL_BLOCK, true>(A, B, a0, a1, threadIdx.x, smem + 128, hdr, [&](uint32_t c, double v) {
            const uint32_t w = c >> 5;
            const uint32_t pos = pre[w] + __popc(bm[w] & ((1u << (c & 31)) - 1u));
            atomicAdd(&c_val[c0 + pos], v);
        });
        __syncthreads();
        if (lo != 0xFFFFFFFFu)
            for (uint32_t w = lo + threadIdx.x; w <= hi; w += SPILL_BLOCK) bm[w] = 0;
        stores_to_l2();
        __syncthreads();
    }
}

// ---- 8. large rows, LDS bitmap-rank path (matrices with <= ~1.1 M columns) ----------------------------------
// Same idea as the HBM spill path, but the row's column bitmap lives in LDS (cols/8 bytes: 128 KiB at 1 M columns,
// which fits the 160 KiB LDS of a CDNA4 CU): products set bits with ds_or, the popcount prefix gives each column
// its final position, values are accumulated at that position -- in an LDS f64 row when nnz(C_i) fits
// (LDS_VALS), else directly in the C row with device-scope atomics (compact, L2 resident).  No hash, no sort.
// LDS: 128 B hdr | walk scratch | coarse u32[W/8] | bitmap u32[W] | vals f64[vcap] (LDS_VALS only)
constexpr int BM_BLOCK = 512;

__host__ __device__ inline uint32_t bm_groups_per_thread(uint64_t cols)
{
    const uint64_t groups = (cols + 255) / 256;               // 8-word (256-column) groups
    return (uint32_t)((groups + BM_BLOCK - 1) / BM_BLOCK);
}
__host__ __device__ inline size_t bm_lds_bytes(uint64_t cols, uint32_t vcap)
{
    const size_t W = (size_t)bm_groups_per_thread(cols) * BM_BLOCK * 8;
    return 128 + ((walk_scratch_bytes<BM_BLOCK, true>() + 15) & ~(size_t)15) + (W / 8) * 4 + W * 4 + (size_t)vcap * 8;
}

__device__ inline uint32_t bm_block_scan(uint32_t v, uint32_t *wtot, uint32_t *total)
{
    return group_scan_excl<BM_BLOCK>(v, threadIdx.x, wtot, total);
}

__global__ __launch_bounds__(BM_BLOCK) void k_sym_bitmap(DevCsrView A, DevCsrView B, uint64_t r0,
                                                         const uint32_t *__restrict__ bin_rows, uint32_t n_bin_rows,
                                                         uint64_t cols, uint32_t *__restrict__ row_nnzc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    unsigned char *scratch = smem + 128;
    const uint32_t gpt = bm_groups_per_thread(cols);
    const uint32_t W = gpt * BM_BLOCK * 8;
    uint32_t *bm = (uint32_t *)(scratch + ((walk_scratch_bytes<BM_BLOCK, true>() + 15) & ~(size_t)15) + (W / 8) * 4);
    uint4 *bm4 = (uint4 *)bm;
    for (uint32_t i = threadIdx.x; i < W / 4; i += BM_BLOCK) bm4[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    for (uint32_t slot = blockIdx.x; slot < n_bin_rows; slot += gridDim.x) {
        const uint32_t row = bin_rows[slot];
        const uint64_t a0 = A.ptr[r0 + row], a1 = A.ptr[r0 + row + 1];
        walk_products<BM_BLOCK, false>(A, B, a0, a1, threadIdx.x, scratch, hdr,
                                       [&](uint32_t c, double) { atomicOr(&bm[c >> 5], 1u << (c & 31)); });
        __syncthreads();
        uint32_t cnt = 0;
        for (uint32_t i = threadIdx.x; i < W / 4; i += BM_BLOCK) {
            const uint4 w = bm4[i];
            cnt += __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w);
            bm4[i] = make_uint4(0, 0, 0, 0);
        }
        cnt = group_sum<BM_BLOCK>(cnt, hdr);
        if (threadIdx.x == 0) row_nnzc[row] = cnt;
        __syncthreads();
    }
}

template <bool LDS_VALS>
__global__ __launch_bounds__(BM_BLOCK) void k_num_bitmap(DevCsrView A, DevCsrView B, uint64_t r0,
                                                         const uint32_t *__restrict__ bin_rows, uint32_t n_bin_rows,
                                                         uint64_t cols, uint32_t vcap, const uint64_t *__restrict__ cptr,
                                                         uint32_t *__restrict__ c_idx, double *__restrict__ c_val,
                                                         uint32_t *__restrict__ queue /* zeroed: next row of the list */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    unsigned char *scratch = smem + 128;
    const uint32_t gpt = bm_groups_per_thread(cols);
    const uint32_t W = gpt * BM_BLOCK * 8;
    uint32_t *coarse = (uint32_t *)(scratch + ((walk_scratch_bytes<BM_BLOCK, true>() + 15) & ~(size_t)15));
    uint32_t *bm = coarse + W / 8;
    uint4 *bm4 = (uint4 *)bm;
    double *vals = (double *)(bm + W);
    for (uint32_t i = threadIdx.x; i < W / 4; i += BM_BLOCK) bm4[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    for (;;) {
        // the list is sorted by products, largest first: dequeue dynamically (one device-scope atomic per row)
        if (threadIdx.x == 0) hdr[31] = atomicAdd(queue, 1u);
        __syncthreads();
        const uint32_t slot = hdr[31];
        __syncthreads();
        if (slot >= n_bin_rows) break;
        const uint32_t row = bin_rows[slot];
        const uint64_t a0 = A.ptr[r0 + row], a1 = A.ptr[r0 + row + 1];
        const uint64_t c0 = cptr[row];
        const uint32_t n = (uint32_t)(cptr[row + 1] - c0);
        // pass A: pattern
        walk_products<BM_BLOCK, false>(A, B, a0, a1, threadIdx.x, scratch, hdr,
                                       [&](uint32_t c, double) { atomicOr(&bm[c >> 5], 1u << (c & 31)); });
        __syncthreads();
        // pass B: every thread owns gpt consecutive 8-word groups; prefix popcounts; column indices out
        uint32_t mine = 0;
        const uint32_t g0 = threadIdx.x * gpt;
        for (uint32_t g = g0; g < g0 + gpt; ++g) {
            const uint4 lo4 = bm4[2 * g], hi4 = bm4[2 * g + 1];
            mine += __popc(lo4.x) + __popc(lo4.y) + __popc(lo4.z) + __popc(lo4.w) + __popc(hi4.x) + __popc(hi4.y) +
                    __popc(hi4.z) + __popc(hi4.w);
        }
        uint32_t total;
        uint32_t pos = bm_block_scan(mine, hdr + 2, &total);
        for (uint32_t g = g0; g < g0 + gpt; ++g) {
            coarse[g] = pos;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t bits = bm[g * 8 + j];
                while (bits) {
                    const int b = __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    c_idx[c0 + pos] = ((g * 8 + j) << 5) + (uint32_t)b;
                    ++pos;
                }
            }
        }
        if constexpr (LDS_VALS) {
            for (uint32_t i = threadIdx.x; i < n; i += BM_BLOCK) vals[i] = 0.0;
        } else {
            for (uint32_t i = threadIdx.x; i < n; i += BM_BLOCK) c_val[c0 + i] = 0.0;
            stores_to_l2();
        }
        __syncthreads();
        // pass C: every product lands at its final position
        walk_products<BM_BLOCK, true>(A, B, a0, a1, threadIdx.x, scratch, hdr, [&](uint32_t c, double v) {
            const uint32_t w = c >> 5, g = w >> 3, j = w & 7;
            const uint4 lo4 = bm4[2 * g], hi4 = bm4[2 * g + 1];
            const uint32_t ww[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
            uint32_t p = coarse[g];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t m = (uint32_t)t < j ? 0xFFFFFFFFu : ((uint32_t)t == j ? ((1u << (c & 31)) - 1u) : 0u);
                p += __popc(ww[t] & m);
            }
            if constexpr (LDS_VALS) atomicAdd(&vals[p], v);
            else atomicAdd(&c_val[c0 + p], v);
        });
        __syncthreads();
        if constexpr (LDS_VALS)
            for (uint32_t i = threadIdx.x; i < n; i += BM_BLOCK) c_val[c0 + i] = vals[i];
        for (uint32_t i = threadIdx.x; i < W / 4; i += BM_BLOCK) bm4[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
}

// widen u32 column indices to the ABI's u64 (usize)
__global__ __launch_bounds__(256) void k_widen_u32(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = in[i];
}

}  // namespace spada
