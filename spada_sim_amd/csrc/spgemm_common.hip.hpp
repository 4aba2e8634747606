// gfx950 (MI355X, CDNA4) device helpers shared by the kernels of the task pipeline (spgemm_task.hip.hpp): wave / workgroup
// scans and reductions, the LDS hash slot, and the FLAT PRODUCT WALK -- the balanced expansion of the products of a run of
// A entries that every kernel of the path uses (row-wise expand step of the reference: scheduler.rs:482-606 window fetch,
// simulator.rs:892-953 B-fiber streaming, simulator.rs:86-111 multiplier).  64-wide wavefronts throughout; no MFMA: this path
// is irregular gather-accumulate, bounded by memory latency, LDS throughput and HBM bandwidth.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spada_ffi.h"

namespace spada {

constexpr uint32_t EMPTY_KEY = 0xFFFFFFFFu;

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

// Workgroup barrier that orders LDS accesses only: loads from and stores to global memory that are in flight stay in flight
// (__syncthreads() is a fence over all address spaces and may wait for them).
__device__ inline void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
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

// ---- scans on DPP (data-parallel primitives of the vector ALU: one instruction per step, no LDS round trip) -------------------
// inclusive sum over the 64 lanes of a wave: row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then lane 15 of rows 0 and 2 to
// rows 1 and 3 (row_bcast:15), then lane 31 to the upper half (row_bcast:31)
__device__ inline uint32_t wave_scan_incl_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
// exclusive sum over the 256 threads; ONE barrier.  `slot` = four LDS words (16-byte aligned) that no other scan of the same
// barrier interval uses: consecutive scans take different slots, so none has to wait for the readers of the one before.
__device__ inline uint32_t block_scan_excl_dpp(uint32_t v, uint32_t *slot, uint32_t *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_scan_incl_u32(v);
    if (lane == 63) slot[w] = inc;
    __syncthreads();
    const uint4 t = *(const uint4 *)slot;
    *total = t.x + t.y + t.z + t.w;
    return inc - v + (w > 0 ? t.x : 0u) + (w > 1 ? t.y : 0u) + (w > 2 ? t.z : 0u);
}

// The same for a workgroup of NW waves, in two halves, so that several scans can share ONE barrier: scan_part() before the barrier
// (wave scan; the wave's total goes to slot[wave]), scan_done() after it (the totals of the waves before this one are added).
// `slot` = NW words, 16-byte aligned, not used by another scan of the same barrier interval.
// (`tid` = the caller's thread number: a caller that keeps it opaque per task -- spgemm_batch.hip.hpp -- passes its own)
__device__ inline uint32_t scan_part(uint32_t v, uint32_t *slot, uint32_t tid)
{
    const uint32_t inc = wave_scan_incl_u32(v);
    if ((tid & 63u) == 63u) slot[tid >> 6] = inc;
    return inc;
}
// (`nw`: only the first nw waves have called scan_part -- the others hold zeros and have left their slots alone)
template <int NW>
__device__ inline uint32_t scan_done(uint32_t inc, uint32_t v, const uint32_t *slot, uint32_t *total, uint32_t tid, uint32_t nw = NW)
{
    static_assert(NW <= 16, "the waves' totals fit one row of 16 lanes");
    // the waves' totals scanned once more, by every wave for itself: lane k < NW takes total k, four DPP steps inside the row of 16
    // lanes, and the two numbers a wave needs -- the totals before it, the total of all -- are scalar reads of that register
    // (every thread adding up NW words with compares and selects took 30 vector instructions per scan, five scans per task)
    const uint32_t lane = tid & 63u, w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    uint32_t p = lane < nw ? slot[lane] : 0u;
    p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x111, 0xf, 0xf, false);
    p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x112, 0xf, 0xf, false);
    p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x114, 0xf, 0xf, false);
    if (NW > 8) p += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x118, 0xf, 0xf, false);
    *total = (uint32_t)__builtin_amdgcn_readlane((int)p, NW - 1);
    const uint32_t before = w ? (uint32_t)__builtin_amdgcn_readlane((int)p, (int)(w - 1u)) : 0u;
    return inc - v + before;
}
template <int NW>
__device__ inline uint32_t block_scan_excl_dpp_n(uint32_t v, uint32_t *slot, uint32_t *total, uint32_t tid)
{
    const uint32_t inc = scan_part(v, slot, tid);
    __syncthreads();
    return scan_done<NW>(inc, v, slot, total, tid);
}

// in-place exclusive scan of arr[0 .. 4 * 256) by a workgroup of 256 threads (four consecutive elements each); ends with a barrier
__device__ inline void block_exclusive_scan4_dpp(uint32_t *arr, uint32_t *slot)
{
    const int tid = threadIdx.x;
    const uint32_t a = arr[tid * 4 + 0], b = arr[tid * 4 + 1], c = arr[tid * 4 + 2], d = arr[tid * 4 + 3];
    uint32_t total;
    const uint32_t ex = block_scan_excl_dpp(a + b + c + d, slot, &total);
    arr[tid * 4 + 0] = ex;
    arr[tid * 4 + 1] = ex + a;
    arr[tid * 4 + 2] = ex + a + b;
    arr[tid * 4 + 3] = ex + a + b + c;
    __syncthreads();
}

constexpr uint32_t LR_NONE = 0xFFFFFFFFu;

// ---- the flat product walk -----------------------------------------------------------------------------------------
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
constexpr int FLAT_U = 4;   // segments of 64 products per thread and round (2: +0 ... 2 %, 8: +3 ... 6 %)
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

// `trim(b0, len)` may narrow the (begin, length) descriptors of the EPT entries a thread has just loaded (NoTrim: whole B rows).
struct NoTrim {
    template <int EPT>
    __device__ inline void operator()(uint64_t (&)[EPT], uint32_t (&)[EPT]) const {}
};

template <int BLOCK, int EPT, int RMAX, bool NUMERIC, int U, class F, class Trim = NoTrim>
__device__ inline void flat_walk(const uint32_t *s_re, const uint64_t *s_a0, uint32_t R, uint32_t E,
                                 const uint64_t *__restrict__ eb0, const uint32_t *__restrict__ elen,
                                 const double *__restrict__ aval, const uint32_t *__restrict__ bidx,
                                 const double *__restrict__ bval, unsigned char *scratch, uint32_t *hdr, F &&f, Trim trim = Trim())
{
    uint32_t pbase = 0;   // products of the chunks already walked
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
            trim(b0, len);
        }
        // packed scan: (entries with products) << 32 | products
        unsigned long long mine = 0;
#pragma unroll
        for (int i = 0; i < EPT; ++i) mine += len[i] ? ((1ull << 32) | len[i]) : 0ull;
        unsigned long long tot64;
        unsigned long long ex64 = group_scan_excl_u64<BLOCK>(mine, tid, (unsigned long long *)(hdr + 4), &tot64);
        const uint32_t total = (uint32_t)tot64;
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
            for (uint32_t base = lo; base < hi; base += U * BLOCK) {   // uniform trip count over the workgroup
                uint32_t col[U], plr[U];
                double v[U];
                uint32_t pp[U], j[U];
                bool act[U];
                // lane l of a segment holds product seg + l, i.e. bit l of one bitmap word (lo, base and the segments are
                // multiples of 64): the word and its prefix are wave-uniform reads, the rank is a v_mbcnt pair.  Lanes past
                // the end fall back to product 0 of entry 0 (a valid address; their result is discarded).
                // (the 2 U bitmap reads are issued together, then the U entry records, then the gathers: asm pins keep the
                // compiler from sinking each read into the branch of its own `act` and waiting for them one by one)
                unsigned long long bits[U];
                uint32_t bp[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t seg = base + (u * NW + wave_u) * 64;      // wave-uniform
                    const uint32_t w = min((seg - lo) >> 6, (uint32_t)PWORDS - 1u);
                    bits[u] = bm[w];
                    bp[u] = bpre[w];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) asm volatile("" : "+v"(bits[u]), "+v"(bp[u]));
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t p = base + (u * NW + wave_u) * 64 + lane;
                    act[u] = p < hi;
                    const uint32_t below =
                        __builtin_amdgcn_mbcnt_hi((uint32_t)(bits[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bits[u], 0u));
                    const uint32_t self = (bits[u] & lane_bit) ? 1u : 0u;
                    j[u] = act[u] ? bp[u] + below + self - 1u : 0u;
                    pp[u] = act[u] ? p : 0u;
                }
                uint64_t q[U], pack[U];
                double a_[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    a_[u] = 0.0;
                    if constexpr (NUMERIC) {
                        const EntryRecNum er = w_ent[j[u]];
                        pack[u] = er.pack;
                        a_[u] = er.av;
                    } else {
                        pack[u] = w_pack[j[u]];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    asm volatile("" : "+v"(pack[u]));
                    if constexpr (NUMERIC) asm volatile("" : "+v"(a_[u]));
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    q[u] = ((pack[u] & M48) + pp[u]) & M48;
                    plr[u] = act[u] ? (uint32_t)(pack[u] >> 48) : LR_NONE;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) col[u] = bidx[q[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    v[u] = 0.0;
                    if constexpr (NUMERIC) v[u] = a_[u] * bval[q[u]];   // simulator.rs:100-101
                }
#pragma unroll
                for (int u = 0; u < U; ++u) pp[u] += pbase;
                f(col, plr, v, pp);
            }
            __syncthreads();
        }
        pbase += total;
    }
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

// widen u32 column indices to the ABI's u64 (usize)
__global__ __launch_bounds__(256) void k_widen_u32(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = in[i];
}

}  // namespace spada
