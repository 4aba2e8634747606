// gfx950 kernels of the task pipeline, part 2 of 5: the BIG-ROW STAGE -- rows with more products than one task takes are cut into parts
// (k_big_parts), their products counted over column buckets (k_big_hist), the buckets packed into column ranges and the row left to DIRECT range
// tasks or SPILLED (k_big_plan); the cut table of the direct rows (big_cuts_body) and the scatter of the spilled rows (big_scatter_body) run
// inside k_after_plan (spgemm_prekernel.hip.hpp).  What it replaces: the partial-fiber merging of rows that exceed a PE (scheduler.rs:381-480,
// adder_tree.rs:145-188) and the psum spill to DRAM (storage.rs:599-658).
#pragma once
#include "spgemm_defs.hip.hpp"

namespace spada {

// ---- 2. BIG rows: histogram, column ranges, spill of the largest rows into HBM scratch --------------------------------------
// A BIG row (more products than one task's table takes) becomes column-RANGE tasks.  It is first cut into PARTS of ~BX_PART
// products (whole A entries), so that a row with a million products is handled by hundreds of workgroups and the largest row
// does not set the time:
//   k_big_parts   one wave per BIG row: running sum of its entries' B-row lengths; a new part starts wherever
//                 floor(prefix / BX_PART) changes.  The row's parts are consecutive records (+ one sentinel)
//   k_big_hist    one workgroup per part: its products (column indices only) counted in BX_NB column buckets of width
//                 2^wshift over [kmin, kmax] of the row (LDS), stored per part
//   k_big_plan    one workgroup per row: bucket counts of the row = sum over its parts; buckets grouped into column RANGES -- a
//                 bucket with more products than the limit is a range of its own, the others are packed greedily into ranges
//                 of at most `limit` (2040) products, i.e. a light range fits one task's table whatever its outputs
//                 are; a heavy range holds at most 2^wshift distinct columns and is split further by the task itself if
//                 both exceed the table (k_task, multi-pass).  Range descriptors go to `tmp` (bump allocated),
//                 their number to row_m[row].  Then the row is either left to DIRECT range tasks, which find their products
//                 in B themselves, or SPILLED: its slice of the scratch arrays is bump allocated and the counts of every part
//                 are turned into cursors (exclusive prefix over buckets, then over the parts before it)
//   k_big_scatter one workgroup per part of a spilled row: walks its products again and stores (column, a * b) at the bucket's
//                 cursor: afterwards the scratch slice of every range is contiguous
// k_cut3 copies the range descriptors into the task list in row order.
constexpr uint32_t LB_PAUSE_MAX = 2;   // a task polls its status word with a pause that grows by this many steps (0 .. 12: within 1 %)
constexpr uint32_t BX_PART_SHIFT = 13, BX_PART_SHIFT_HUGE = 16;   // products per part of a BIG row: 8192 (4 K / 16 K / 32 K: within 3 % on R-MAT 16, + 5 % on web);
                                                                  // 64 K when the call before on the context had a billion products in BIG rows (spada_engine.hip)
constexpr int CUT_ITEMS = 4, CUT_TILE = 256 * CUT_ITEMS;   // (tiles of 1024 rows: 2048 leaves too few workgroups on the smaller inputs,
                                                           // 512 cuts too many batches at tile borders -- +11 % tasks on the stencil input)
constexpr uint32_t BX_NOPART = 0xFFFFFFFFu;
#ifndef HIST_BY_ENTRY
#define HIST_BY_ENTRY 1
#endif
constexpr uint32_t HIST_ENTRY_MAX = 64, HIST_ENTRY_LEN = 256;   // k_big_hist: parts of at most .. entries of at least .. products on average are walked entry by entry
constexpr uint32_t PLAN_WALK_MAX = 16;   // k_big_plan: rows of at most .. x limit products walk their ranges one after the other (a search by the whole workgroup per range)
constexpr uint32_t PLAN_UNROLL = 4;   // part records of a row whose histograms k_big_plan has in flight together
constexpr uint32_t BX_MARK = 0x80000000u;   // a cursor word of k_big_plan that names the bucket holding the cursor instead (k_big_scatter)
constexpr uint32_t BX_RUN = 8;   // consecutive part records per workgroup (k_big_scatter)
struct BigPart {
    uint32_t slot;      // position of the row in big_rows; BX_NOPART: sentinel / unused record
    uint32_t p_begin;   // products of the row before the part (sort-merge: product numbers)
    uint64_t e_begin;   // first A entry of the part; the part ends where the next record begins
};
struct BigSlot {
    uint64_t scr_base;  // first product of the row in the scratch arrays
    uint32_t ok;        // 0: a workspace was too small, nothing of the row is written
    uint32_t direct;    // 1: the row is not spilled, its range tasks walk B themselves (k_big_plan)
    uint32_t part_begin, part_count;   // records of the row: parts[part_begin .. part_begin + part_count], the last a sentinel
    uint64_t cut_base;  // direct rows with at most BT_EMAX entries: first word of the row's (ranges + 1) x entries cut table
};

__device__ inline uint32_t big_wshift(uint32_t kmin, uint32_t kmax)
{
    uint32_t w = 0;
    while (((kmax - kmin) >> w) >= (uint32_t)BX_NB) ++w;
    return w;
}

constexpr int BP_EPL = 8;     // entries per lane and step
constexpr int BP_ROWS = 16;   // rows per workgroup and round: their records are allocated with ONE device atomic per array (a
                              // single hot word sustains ~88 atomics / us: one per row would cost more than the kernel's work)
// ranges of a row with P products, upper bound: a light range is closed when the next bucket does not fit, so two consecutive
// ones hold more than `lim` products together; a heavy bucket (more than `lim` products) ends the range before it and is one itself
// -- plus, for a row whose buckets are wider than the table (`wide`), up to BX_SUB_MAX descriptors for each of its at most
// P / lim heavy buckets (below: column sub-ranges)
constexpr uint32_t BX_SUB_MAX = 8;
constexpr uint32_t BX_ARENAS = 16;
constexpr uint32_t BX_CUT_ITEM = 256;   // (range, entry) pairs -- binary searches -- per work item of k_big_cuts: one per thread
__host__ __device__ inline uint32_t big_max_ranges(uint32_t P, uint32_t lim, bool wide)
{
    return 2u * (P / lim) + 2u * (P / (lim + 1u)) + 3u + (wide ? (BX_SUB_MAX - 1u) * (P / lim) : 0u);
}

__global__ __launch_bounds__(BP_ROWS * 64) void k_big_parts(const uint64_t *__restrict__ aptr, const uint32_t *__restrict__ elen, uint64_t r0,
                                                   const uint32_t *__restrict__ big_rows, const uint32_t *__restrict__ row_nprod,
                                                   const uint32_t *__restrict__ row_kmin, const uint32_t *__restrict__ row_kmax,
                                                   uint32_t allow_sub, uint32_t psh /* log2 of the products per part */,
                                                   BigPart *__restrict__ parts, uint32_t part_cap,
                                                   uint32_t *__restrict__ row_tmp, uint32_t tmp_cap, BigSlot *__restrict__ slots,
                                                   TaskCounters *__restrict__ ctr)
{
    const uint32_t lim = ctr->prod_limit;
    __shared__ uint32_t s_pbase[BP_ROWS], s_tbase[BP_ROWS];
    // (a row whose histogram buckets are wider than the table may get several descriptors per heavy bucket: k_big_plan)
    auto wide_row = [&](uint32_t row) { return allow_sub && (1ull << big_wshift(row_kmin[row], row_kmax[row])) > (unsigned long long)TK_NOUT; };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nbig = ctr->n_big;
    for (uint32_t s0 = blockIdx.x * BP_ROWS; s0 < nbig; s0 += gridDim.x * BP_ROWS) {
        __syncthreads();
        if (wave == 0) {   // records of the round's rows: parts (+ sentinel) and range descriptors (upper bound)
            const uint32_t sl = s0 + lane;
            const bool have = lane < BP_ROWS && sl < nbig;
            const uint32_t P = have ? row_nprod[big_rows[sl]] : 0u;
            const bool good = have && P != 0xFFFFFFFFu;
            const uint32_t np = good ? (P >> psh) + 2u : 0u, nt = good ? big_max_ranges(P, lim, wide_row(big_rows[sl])) : 0u;
            uint32_t ip = np, it = nt;
#pragma unroll
            for (int o = 1; o < BP_ROWS; o <<= 1) {
                const uint32_t a = __shfl_up(ip, o), b = __shfl_up(it, o);
                if (lane >= o) {
                    ip += a;
                    it += b;
                }
            }
            uint32_t bp = 0, bt = 0;
            if (lane == BP_ROWS - 1) {
                bp = atomicAdd(&ctr->n_parts, ip);
                bt = atomicAdd(&ctr->tmp_cursor, it);
            }
            bp = __shfl(bp, BP_ROWS - 1);
            bt = __shfl(bt, BP_ROWS - 1);
            if (lane < BP_ROWS) {
                s_pbase[lane] = bp + ip - np;
                s_tbase[lane] = bt + it - nt;
            }
        }
        __syncthreads();
        for (int rr = wave; rr < BP_ROWS; rr += (int)(blockDim.x >> 6)) {   // (one wave per row of the round: sixteen waves since round 6, a row's walk is a chain of dependent steps)
        const uint32_t slot = s0 + rr;
        if (slot >= nbig) break;
        const uint32_t row = big_rows[slot];
        const uint32_t P = row_nprod[row];
        if (P == 0xFFFFFFFFu) {   // 2^32 or more products in one row: 32-bit counters would wrap
            if (lane == 0) atomicOr(&ctr->abort_flag, 4u);
            continue;
        }
        const uint64_t a0 = aptr[r0 + row], a1 = aptr[r0 + row + 1];
        // every window [w 2^psh, (w + 1) 2^psh) of the running product count that contains the first product of some entry
        // starts a part: at most `ub` of them (an entry with an empty B row may sit at prefix P itself)
        const uint32_t ub = (P >> psh) + 1;
        const uint32_t base = s_pbase[rr], tbase = s_tbase[rr];
        const bool fits = (unsigned long long)base + ub + 1 <= part_cap;
        if (lane == 0) {
            slots[slot] = BigSlot{0ull, 0u, 0u, base, ub, 0ull};
            row_tmp[row] = tbase;
            if (!fits) atomicOr(&ctr->abort_flag, 16u);
            if ((unsigned long long)tbase + big_max_ranges(P, lim, wide_row(row)) > tmp_cap) atomicOr(&ctr->abort_flag, 1u);
        }
        if (!fits) continue;
        uint32_t carry = 0, nstart = 0;   // products / parts before this step
        uint32_t prev_win = 0xFFFFFFFFu;  // window of the entry before this step (none: the first entry starts a part)
        uint32_t nlen[BP_EPL];   // the lengths of the step after this one: loaded a step ahead (a hub row is a hundred dependent steps of one wave)
#pragma unroll
        for (int i = 0; i < BP_EPL; ++i) nlen[i] = a0 + (uint64_t)lane * BP_EPL + i < a1 ? elen[a0 + (uint64_t)lane * BP_EPL + i] : 0u;
        for (uint64_t q0 = a0; q0 < a1; q0 += 64 * BP_EPL) {
            const uint64_t q = q0 + (uint64_t)lane * BP_EPL;
            uint32_t len[BP_EPL], sum = 0;
#pragma unroll
            for (int i = 0; i < BP_EPL; ++i) {
                len[i] = nlen[i];
                sum += len[i];
            }
            if (q0 + 64 * BP_EPL < a1) {
                const uint64_t qn = q + 64 * BP_EPL;
#pragma unroll
                for (int i = 0; i < BP_EPL; ++i) nlen[i] = qn + i < a1 ? elen[qn + i] : 0u;
            }
            uint32_t inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            uint32_t ex = carry + inc - sum;
            // window of the last entry of the lane before (lanes past the end repeat the last window: no new start there)
            const uint32_t my_last = (ex + sum - len[BP_EPL - 1]) >> psh;   // (entries past the end have length 0)
            uint32_t pw = __shfl_up(my_last, 1);
            if (lane == 0) pw = prev_win;
            uint32_t w[BP_EPL], exi[BP_EPL], cnt = 0;
#pragma unroll
            for (int i = 0; i < BP_EPL; ++i) {
                exi[i] = ex;
                w[i] = ex >> psh;
                ex += len[i];
            }
            uint32_t startmask = 0, p = pw;
#pragma unroll
            for (int i = 0; i < BP_EPL; ++i) {
                if (q + i < a1 && w[i] != p) {
                    startmask |= 1u << i;
                    ++cnt;
                }
                if (q + i < a1) p = w[i];
            }
            uint32_t cinc = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(cinc, o);
                if (lane >= o) cinc += t;
            }
            uint32_t k = nstart + cinc - cnt;
#pragma unroll
            for (int i = 0; i < BP_EPL; ++i)
                if (startmask & (1u << i)) parts[base + k++] = BigPart{slot, exi[i], q + i};
            nstart += __shfl(cinc, 63);
            carry += __shfl(inc, 63);
            // window of the last entry of the step
            const uint64_t last_q = min(q0 + 64 * BP_EPL, a1) - 1;
            const int ll = (int)((last_q - q0) / BP_EPL), li = (int)((last_q - q0) % BP_EPL);
            uint32_t wl = 0;
#pragma unroll
            for (int i = 0; i < BP_EPL; ++i) wl = li == i ? w[i] : wl;
            prev_win = __shfl(wl, ll);
        }
        for (uint32_t k = nstart + lane; k <= ub; k += 64) parts[base + k] = BigPart{BX_NOPART, P, a1};
        }
    }
}

// Runs of equal buckets in a wave.  The lanes of a wave hold consecutive products, i.e. (mostly) consecutive entries of ONE sorted B
// row: on a skewed input -- the popular columns of an R-MAT graph -- dozens of neighbouring lanes fall into the same bucket and an
// LDS atomic per lane serialises on one address.  The first lane of every run speaks for the run: `head`, the run's length, and
// for every lane the lane of its head.  key = 0xFFFFFFFF marks a lane without a product (such lanes form runs that add nothing).
// Returns false -- and nothing else -- when no two neighbouring lanes share a bucket (meshes: the check costs three instructions,
// the run bookkeeping a dozen and a cross-lane read).
__device__ inline bool wave_runs(uint32_t key, bool &head, uint32_t &len, uint32_t &head_lane)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
    head = lane == 0 || prev != key;
    const unsigned long long heads = __ballot(head);
    if (heads == ~0ull) return false;
    const unsigned long long above = (heads >> lane) >> 1;   // heads in the lanes above this one
    len = above ? (uint32_t)__ffsll((long long)above) : 64u - lane;
    head_lane = 63u - (uint32_t)__clzll((long long)(heads & ((2ull << lane) - 1ull)));   // (lane 0 is always a head)
    return true;
}

// LDS of k_big_hist / k_big_scatter: 256 B hdr | cnt u32[NB] | s_re u32[4], s_a0 u64[2] | walk scratch
constexpr size_t BX_WALK_LDS = 256 + (size_t)BX_NB * 4 + 32 + flat_walk_bytes<TK_BLOCK, TK_EPT, true>() + 16;

__global__ __launch_bounds__(TK_BLOCK, 8) void k_big_hist(const uint32_t *__restrict__ bidx, const uint64_t *__restrict__ eb0,
                                                       const uint32_t *__restrict__ elen, const uint32_t *__restrict__ big_rows,
                                                       const uint32_t *__restrict__ row_kmin, const uint32_t *__restrict__ row_kmax,
                                                       const BigPart *__restrict__ parts, uint32_t *__restrict__ part_hist,
                                                       const TaskCounters *__restrict__ ctr)
{
    constexpr int NB = BX_NB, U = FLAT_U;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    uint32_t *cnt = (uint32_t *)(smem + 256);
    uint32_t *s_re = cnt + NB;
    uint64_t *s_a0 = (uint64_t *)(s_re + 4);
    unsigned char *scratch = (unsigned char *)(s_a0 + 2);
    const int tid = threadIdx.x;
    if (ctr->abort_flag) return;
    const uint32_t nparts = ctr->n_parts;
    for (uint32_t pi = blockIdx.x; pi < nparts; pi += gridDim.x) {
        const BigPart pt = parts[pi];
        if (pt.slot == BX_NOPART) continue;   // (uniform)
        const uint32_t e_count = (uint32_t)(parts[pi + 1].e_begin - pt.e_begin);
        const uint32_t row = big_rows[pt.slot];
        const uint32_t kmin = row_kmin[row], wshift = big_wshift(kmin, row_kmax[row]);
        for (int b = tid; b < NB; b += TK_BLOCK) cnt[b] = 0;
        if (tid == 0) {
            s_re[0] = 0;
            s_re[1] = e_count;
            s_a0[0] = pt.e_begin;
        }
        // A part of FEW LONG entries (a hub row of an R-MAT graph: 92 % of its products come from B rows of 1000 columns and more) is
        // walked entry by entry, the workgroup striding along each B row: no owner lookup per product (the flat walk's bitmaps and
        // entry records: 50 VALU + 24 SALU instructions per 64 products against ~25 here).  Other parts: the flat walk.
        const uint32_t p_count = parts[pi + 1].p_begin - pt.p_begin;   // (the record behind a row's last part holds all its products)
        const bool by_entry = HIST_BY_ENTRY && e_count <= HIST_ENTRY_MAX && (uint64_t)e_count * HIST_ENTRY_LEN <= p_count;
        uint64_t *s_eb = (uint64_t *)scratch;
        uint32_t *s_el = (uint32_t *)(s_eb + HIST_ENTRY_MAX);
        if (by_entry && (uint32_t)tid < e_count) {
            s_eb[tid] = eb0[pt.e_begin + tid];
            s_el[tid] = elen[pt.e_begin + tid];
        }
        __syncthreads();
        auto count = [&](uint32_t col, bool on) {
            const uint32_t bk = on ? (col - kmin) >> wshift : 0xFFFFFFFFu;
            bool head;
            uint32_t len, hl;
            if (!wave_runs(bk, head, len, hl)) len = 1u;   // (every lane its own run)
            if (head && bk != 0xFFFFFFFFu) atomicAdd(&cnt[bk], len);
        };
        if (by_entry) {
            for (uint32_t i = 0; i < e_count; ++i) {
                const uint64_t b0 = s_eb[i];
                const uint32_t ln = s_el[i];   // (uniform)
                uint32_t j = 0;
                for (; j + 4u * TK_BLOCK <= ln; j += 4u * TK_BLOCK) {   // four loads in flight
                    uint32_t c[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) c[u] = bidx[b0 + j + (uint32_t)u * TK_BLOCK + tid];
#pragma unroll
                    for (int u = 0; u < 4; ++u) count(c[u], true);
                }
                for (; j < ln; j += TK_BLOCK) {
                    const bool on = j + tid < ln;
                    count(on ? bidx[b0 + j + tid] : 0u, on);
                }
            }
        } else {
        flat_walk<TK_BLOCK, TK_EPT, 1, false, U>(s_re, s_a0, 1u, e_count, eb0, elen, nullptr, bidx, nullptr, scratch, hdr,
                                                 [&](uint32_t(&col)[U], uint32_t(&plr)[U], double(&)[U], uint32_t(&)[U]) {
#pragma unroll
                                                     for (int u = 0; u < U; ++u) count(col[u], plr[u] != LR_NONE);
                                                 });
        }
        __syncthreads();
        // stored as EXCLUSIVE PREFIXES over the buckets (the part's products before every bucket): the sums over the parts that
        // k_big_plan forms are then the row's prefixes, and the cursors of a part need no scan in its serial loop over the parts
        block_exclusive_scan4_dpp(cnt, hdr + 4);
        ((uint4 *)(part_hist + (size_t)pi * NB))[tid] = ((const uint4 *)cnt)[tid];
        __syncthreads();
    }
}

// LDS: 256 B hdr | cnt u32[NB] | pre u32[NB + 1] | aux u32[NB + 1] | rfirst u32[NB + 1]
constexpr size_t BX_PLAN_LDS = 256 + (size_t)BX_NB * 4 + (size_t)(BX_NB + 1) * 4 * 3 + 16;
static_assert(BX_NB == 4 * TK_BLOCK, "a thread owns four consecutive buckets (one uint4 of a part's counts)");

// A row is spilled only if that is cheaper than letting each of its m range tasks find its products in B: a DIRECT range task
// loads the row's E entries and narrows every selected B row to its column range with two binary searches (B rows are sorted),
// so the row costs m * E searches of 1 + log2(P / E) steps instead of a scatter to and a read from HBM.  Direct if
//   m * E * steps <= BX_DIRECT_FACTOR * P   (the searches of the whole row against its products; measured on MI355X: factors
//                                            2 .. 32 within 1 % on the web and mesh surrogates, where nearly every BIG row
//                                            qualifies; 8 best on R-MAT 16) and
//   E * steps <= BX_DIRECT_MAX_SEARCH       (the searches of ONE task: a task that takes long to count its outputs holds up the
//                                            offsets of every task behind it; 4096 = the knee on the web surrogate)
// -- rows with few ranges (web graphs, meshes) go direct, rows with thousands of entries and hundreds of ranges (R-MAT hubs) are
// spilled.  `allow_direct` = 0 spills every row (the sort-merge accumulator numbers the products of a slice in scratch order).
constexpr uint32_t BX_DIRECT_FACTOR = 8, BX_DIRECT_MAX_SEARCH = 4096;
__global__ __launch_bounds__(TK_BLOCK, 8) void k_big_plan(const uint64_t *__restrict__ aptr, uint64_t r0, uint32_t nrows_call, uint32_t allow_direct,
                                                       const uint32_t *__restrict__ big_rows, const uint32_t *__restrict__ row_kmin,
                                                       const uint32_t *__restrict__ row_kmax, const BigPart *__restrict__ parts,
                                                       uint32_t *__restrict__ part_hist, uint32_t *__restrict__ row_m,
                                                       const uint32_t *__restrict__ row_tmp, TaskDesc *__restrict__ tmp, uint32_t tmp_cap,
                                                       BigSlot *__restrict__ slots, uint64_t scr_cap, uint64_t cut_cap, uint32_t cut_factor16,
                                                       uint2 *__restrict__ cut_items, uint64_t cut_item_cap, uint32_t range_cursors,
                                                       uint32_t *__restrict__ row_t, uint32_t *__restrict__ tile_tasks,
                                                       uint32_t *__restrict__ spill_parts /* the part records of the spilled rows: what k_big_scatter walks */,
                                                       TaskCounters *__restrict__ ctr)
{
    const uint32_t lim = ctr->prod_limit;
    constexpr int NB = BX_NB, BPT = NB / TK_BLOCK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    uint32_t *cnt = (uint32_t *)(smem + 256);
    uint32_t *pre = cnt + NB;          // exclusive prefix of cnt, pre[NB] = P
    uint32_t *aux = pre + NB + 1;      // start flags -> range numbers; later: nonempty flags -> compact numbers
    uint32_t *rfirst = aux + NB + 1;   // first bucket of range r, rfirst[NR] = NB
    const int tid = threadIdx.x;
    const uint32_t nbig = ctr->n_big;
    // (an overflow of the scratch or descriptor arrays found HERE must not stop the other rows: the retry sizes the arrays from
    // the cursors, which have to be complete)
    if (ctr->abort_flag & ~1u) return;
    for (uint32_t slot = blockIdx.x; slot < nbig; slot += gridDim.x) {

#if SPADA_PRE_DBG == 2
        pre_tick(0);
#endif
        const uint32_t row = big_rows[slot];
        const uint32_t kmin = row_kmin[row], kmax = row_kmax[row], wshift = big_wshift(kmin, kmax);
        const uint32_t pb = slots[slot].part_begin, pc = slots[slot].part_count;
        // (what the allocating thread needs of the row, asked for here: behind the LDS stages these loads were a round trip of their own)
        unsigned long long al_a0 = 0, al_a1 = 0;
        uint32_t al_tb = 0;
        if (tid == TK_BLOCK - 1) {
            al_a0 = aptr[r0 + row];
            al_a1 = aptr[r0 + row + 1];
            al_tb = row_tmp[row];
        }
        uint32_t nreal = 0;   // parts of the row (uniform over the workgroup)
        {   // products of the row before every bucket = the sum of the parts' prefixes (k_big_hist); the bucket counts are its differences
            // (a hub row of R-MAT 22 has 500 parts and ONE workgroup: the records are taken PLAN_UNROLL at a time, their loads in
            // flight together -- a round trip per part made this loop, and the one over the cursors below, as long as the histogram
            // kernel of the whole chunk)
            uint4 acc = make_uint4(0u, 0u, 0u, 0u);
            bool done = false;
            for (uint32_t k = 0; k < pc && !done; k += PLAN_UNROLL) {
                uint32_t sl[PLAN_UNROLL];
                uint4 h[PLAN_UNROLL];
#pragma unroll
                for (uint32_t i = 0; i < PLAN_UNROLL; ++i) sl[i] = k + i < pc ? parts[pb + k + i].slot : BX_NOPART;
#pragma unroll
                for (uint32_t i = 0; i < PLAN_UNROLL; ++i)   // (the records of the row exist up to pc; those behind its last part are read and dropped)
                    h[i] = k + i < pc ? ((const uint4 *)(part_hist + (size_t)(pb + k + i) * NB))[tid] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (uint32_t i = 0; i < PLAN_UNROLL; ++i) {
                    done = done || sl[i] == BX_NOPART;   // (uniform; the records of a row are its parts, then sentinels)
                    if (!done) {
                        acc.x += h[i].x;
                        acc.y += h[i].y;
                        acc.z += h[i].z;
                        acc.w += h[i].w;
                        ++nreal;
                    }
                }
            }
            pre[tid * 4 + 0] = acc.x;
            pre[tid * 4 + 1] = acc.y;
            pre[tid * 4 + 2] = acc.z;
            pre[tid * 4 + 3] = acc.w;
            if (tid == 0) pre[NB] = parts[pb + nreal].p_begin;   // (the record behind the last part: products before it = all of the row)
        }
        __syncthreads();
#if SPADA_PRE_DBG == 2
        pre_tick(1);
#endif

        // range starts: the buckets are packed greedily, as many as fit one task's table (<= lim products: a range is closed when the
        // next bucket does not fit, so two consecutive ranges hold more than lim products together; a heavy bucket -- more than lim
        // products -- fits nowhere and is a range of its own).  Fuller ranges = fewer range tasks, fewer searches of the direct tasks,
        // fewer hops of the chain.  The range that starts at bucket b ends in front of the first bucket whose products no longer
        // fit -- the first e >= b with pre[e + 1] - pre[b] > lim -- or behind b itself if b is heavy.  (Through round 5 the heavy
        // buckets and their successors were also found as "forced" starts, by a scan of their own: redundant -- a heavy bucket
        // closes the range in front of it by the capacity rule already.)
        //   * a row of FEW ranges (at most PLAN_WALK_MAX * lim products: the BIG rows of web graphs and meshes) walks them one
        //     after the other, every step a search by the whole workgroup -- each thread compares its four buckets' prefixes,
        //     which it keeps in registers, one ballot per wave, one barrier;
        //   * a row of many ranges (the hubs of an R-MAT graph) finds the end of the range that WOULD start at every bucket,
        //     all buckets in parallel by binary search over the prefix sums, then one thread follows those pointers.
        // (Phase clocks of round 6, web input, ticks per row: the forced starts 5.2 k, the 1024 binary searches -- four per
        // thread, one after the other -- 10.0 k, the walk 0.8 k of the kernel's 32.6 k; the rows have five ranges.)
        {
#pragma unroll
            for (int k = 0; k < BPT; ++k) aux[tid * BPT + k] = 0u;
            if (pre[NB] <= PLAN_WALK_MAX * lim) {
                uint32_t pv[BPT + 1];
#pragma unroll
                for (int k = 0; k <= BPT; ++k) pv[k] = pre[tid * BPT + k];
                __syncthreads();   // (aux is cleared)
                const uint32_t lane = (uint32_t)tid & 63u, wave = (uint32_t)tid >> 6;
                uint32_t bk = 0, turn = 0;
                while (bk < (uint32_t)NB) {   // (uniform)
                    const uint32_t plim = pre[bk] + lim;
                    uint32_t mine = 0xFFFFFFFFu;
#pragma unroll
                    for (int k = BPT - 1; k >= 0; --k)
                        if ((uint32_t)(tid * BPT + k) >= bk && pv[k + 1] > plim) mine = (uint32_t)(tid * BPT + k);
                    // (the threads hold the buckets in order: the lowest lane with a hit has the wave's first)
                    const unsigned long long hit = __ballot(mine != 0xFFFFFFFFu);
                    const uint32_t cand = hit ? (uint32_t)__shfl((int)mine, __ffsll((long long)hit) - 1) : (uint32_t)NB;
                    uint32_t *wc = hdr + 32 + (turn & 1u) * 4u;   // (two sets in turn: one barrier per range)
                    if (lane == 0) wc[wave] = cand;
                    if (tid == 0) aux[bk] = 1u;
                    __syncthreads();
                    const uint32_t e = min(min(wc[0], wc[1]), min(wc[2], wc[3]));
                    bk = max(e, bk + 1u);
                    ++turn;
                }
            } else {
                // (cnt: the pointers)
#pragma unroll
                for (int k = 0; k < BPT; ++k) {
                    const uint32_t bk = tid * BPT + k;
                    // largest e with pre[e] - pre[bk] <= lim (pre[NB] = P)
                    const uint32_t plim = pre[bk] + lim;
                    uint32_t lo = bk + 1, n = NB - bk;   // e in [bk + 1, NB]: first e with pre[e] > lim, minus one ... searched as upper bound
                    while (n) {
                        const uint32_t h = n >> 1;
                        if (pre[lo + h] <= plim) {
                            lo += h + 1;
                            n -= h + 1;
                        } else {
                            n = h;
                        }
                    }
                    // lo = first index in [bk + 1, NB + 1] whose prefix exceeds lim; the range [bk, lo - 1) fits
                    cnt[bk] = max(lo - 1, bk + 1);
                }
                __syncthreads();
                if (tid == 0)
                    for (uint32_t bk = 0; bk < (uint32_t)NB; bk = cnt[bk]) aux[bk] = 1u;
            }
            __syncthreads();
        }
#if SPADA_PRE_DBG == 2
        pre_tick(2);
        pre_tick(3);
        pre_tick(4);
#endif
        uint32_t stf[BPT];
#pragma unroll
        for (int k = 0; k < BPT; ++k) stf[k] = aux[tid * BPT + k];
        __syncthreads();
        block_exclusive_scan4_dpp(aux, hdr + 4);
#pragma unroll
        for (int k = 0; k < BPT; ++k)
            if (stf[k]) rfirst[aux[tid * BPT + k]] = tid * BPT + k;
        if (tid == TK_BLOCK - 1) {
            const uint32_t NR = aux[NB - 1] + stf[BPT - 1];
            rfirst[NR] = NB;
            hdr[40] = NR;
        }
        __syncthreads();

#if SPADA_PRE_DBG == 2
        pre_tick(5);
#endif
        const uint32_t NR = hdr[40];
        // non-empty ranges, compacted
        uint32_t nef[BPT];
#pragma unroll
        for (int k = 0; k < BPT; ++k) {
            const uint32_t r = tid * BPT + k;
            nef[k] = (r < NR && pre[rfirst[r + 1]] > pre[rfirst[r]]) ? 1u : 0u;
        }
        __syncthreads();
        if (tid == 0) hdr[47] = 0;
        // descriptors per range: one -- or, for a range that would need several passes over its slice (more distinct columns
        // than the table may take: more than TK_NOUT products AND columns; only a heavy bucket of a row that spans more than
        // BX_NB * TK_NOUT columns can be one), one per TK_NOUT columns: every such task reads the whole slice ONCE and keeps the
        // products of its own columns (bit 1 of `first`), instead of one task halving the range depth first with a counting and
        // an accumulating pass over the slice per node -- R-MAT 22's hubs: 24 instead of 36 bytes per product, in independent tasks
        uint32_t wgt[BPT];
#pragma unroll
        for (int k = 0; k < BPT; ++k) {
            wgt[k] = nef[k];
            if (nef[k]) {
                const uint32_t r = tid * BPT + k, f0 = rfirst[r], f1 = rfirst[r + 1];
                const uint64_t lo = (uint64_t)kmin + ((uint64_t)f0 << wshift);
                const uint64_t hi = min((uint64_t)kmin + ((uint64_t)f1 << wshift) - 1ull, (uint64_t)kmax);
                const uint64_t nsub = (hi - lo + (uint64_t)TK_NOUT) / (uint64_t)TK_NOUT;
                if (allow_direct && pre[f1] - pre[f0] > (uint32_t)TK_NOUT && hi - lo >= (uint64_t)TK_NOUT && nsub <= BX_SUB_MAX)
                    wgt[k] = (uint32_t)nsub;
            }
            aux[tid * BPT + k] = wgt[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BPT; ++k)
            if (nef[k]) {   // a range that needs several passes (more than TK_SOLO_MAX products AND columns) exists only on scratch
                const uint32_t r = tid * BPT + k, f0 = rfirst[r], f1 = rfirst[r + 1];
                if (pre[f1] - pre[f0] > lim && ((uint64_t)(f1 - f0) << wshift) > lim) hdr[47] = 1;
            }
        block_exclusive_scan4_dpp(aux, hdr + 4);
#if SPADA_PRE_DBG == 2
        pre_tick(6);
#endif
        if (tid == TK_BLOCK - 1) {
            const uint32_t m = aux[NB - 1] + wgt[BPT - 1];
            const unsigned long long P = pre[NB];
            const unsigned long long a0 = al_a0, E = al_a1 - a0;
            hdr[48] = (uint32_t)min(E, 0x7FFFFFFFull);
            hdr[49] = (uint32_t)a0;
            hdr[50] = (uint32_t)(a0 >> 32);
            const uint32_t avg_len = (uint32_t)min(P / max(E, 1ull), 0xFFFFFFFFull);
            const unsigned long long steps = 1ull + (avg_len ? 31u - (uint32_t)__clz((int)avg_len) : 0u);   // of one binary search
            const bool direct = allow_direct && hdr[47] == 0 && (unsigned long long)m * E * steps <= (unsigned long long)BX_DIRECT_FACTOR * P &&
                                E * steps <= BX_DIRECT_MAX_SEARCH && (E <= BX_DIRECT_EMAX || nrows_call < BX_DIRECT_ROWS);
            const uint32_t tb = al_tb;   // big_max_ranges(P) >= m records, allocated by k_big_parts
            const unsigned long long sb = direct ? 0ull : atomicAdd(&ctr->scratch_cursor, P);
            // the cut table of a direct row whose range tasks run through the batch stages: (ranges + 1) rows of one word per entry
            // -- if the row's searches are few enough for its products (cut_factor16 / 16 searches steps per product: the host's
            // choice per mode).  One search per (range, entry) in a kernel of its own replaces two per pair inside the tasks, where
            // they hide behind other workgroups' work: the table pays when BOTH phases of the two-phase contract read it, and in the
            // one-pass mode for rows whose tasks would otherwise be late for the chain (few searches per product: the web input);
            // the rows of an R-MAT graph -- hundreds of entries, dozens of ranges -- search for themselves there
            const bool few = (unsigned long long)m * E * steps * 16ull <= (unsigned long long)cut_factor16 * P;
            const unsigned long long cw = direct && few && cut_cap != 0ull && E <= (unsigned long long)BT_EMAX ? ((unsigned long long)m + 1ull) * E : 0ull;
            // (by ROW -- the list of BIG rows is in the order of its atomics, and a retry must find the arenas it has sized -- and
            // hashed: the heavy rows of an R-MAT matrix are the ones with few bits set, row % 16 put most of them into arena 0)
            const uint32_t arena = (row * 0x9E3779B1u) >> 28;
            static_assert(BX_ARENAS == 16, "the arena of a row is the top four bits of its hash");
            const unsigned long long acap = cut_cap / BX_ARENAS, icap = cut_item_cap / BX_ARENAS;
            const unsigned long long co = cw ? atomicAdd(&ctr->cut_arena[arena][0], cw) : 0ull, cb = arena * acap + co;
            // ... and its searches as work items of BX_CUT_ITEM (range, entry) pairs each (k_big_cuts)
            const unsigned long long ni = cw ? ((unsigned long long)m * E + BX_CUT_ITEM - 1) / BX_CUT_ITEM : 0ull;
            const unsigned long long io = ni ? atomicAdd(&ctr->cut_arena[arena][1], ni) : 0ull, ib = arena * icap + io;
            hdr[54] = (uint32_t)ib;
            hdr[55] = (uint32_t)(ib >> 32);
            hdr[56] = (uint32_t)ni;
            hdr[46] = direct ? 1u : 0u;
            hdr[42] = tb;
            hdr[43] = (uint32_t)sb;
            hdr[44] = (uint32_t)(sb >> 32);
            hdr[51] = (uint32_t)cb;
            hdr[52] = (uint32_t)(cb >> 32);
            hdr[53] = cw ? 1u : 0u;
            const bool ok = (unsigned long long)tb + m <= tmp_cap && (direct || sb + P <= scr_cap) && co + cw <= acap && io + ni <= icap;
            hdr[45] = ok ? 1u : 0u;
            if (!ok) atomicOr(&ctr->abort_flag, 1u);
            row_m[row] = m;
            hdr[41] = m;
            // (the tiles were cut before this kernel ran -- k_row_class_cut, where a BIG row starts no task yet: its range tasks join
            // the row's and the tile's counts here, one atomic per BIG row spread over the tiles)
            row_t[row] = m;
            atomicAdd(&tile_tasks[row / (uint32_t)CUT_TILE], m);
            slots[slot].scr_base = sb;
            slots[slot].cut_base = cb;
            // (bit 1: ONE cursor per (part, range) -- see the cursors below; their top bit is the mark, so not for a row of 2^31 products)
            slots[slot].ok = ok ? (range_cursors && P < 0x80000000ull ? 3u : 1u) : 0u;
            slots[slot].direct = direct ? 1u : 0u;
            if (!direct) {
                atomicAdd(&ctr->n_spilled, 1u);
                hdr[57] = atomicAdd(&ctr->n_spill_parts, nreal);   // (the row's parts join the scatter's list: its workgroups take nothing else)
            }
        }
        __syncthreads();
#if SPADA_PRE_DBG == 2
        pre_tick(7);
#endif
        if (hdr[46] == 0u)   // (spilled; the list has room for every part record: both are sized by the parts' capacity)
            for (uint32_t i = tid; i < nreal; i += TK_BLOCK) spill_parts[hdr[57] + i] = pb + i;
        const uint32_t tb = hdr[42];
        const uint64_t sb = ((uint64_t)hdr[44] << 32) | hdr[43], cb = ((uint64_t)hdr[52] << 32) | hdr[51];
        const bool ok = hdr[45] != 0, direct = hdr[46] != 0, has_cuts = hdr[53] != 0;
        const uint32_t m_row = hdr[41];   // (not read back from row_m: a store and a load of the same word, a round trip through the L2)
        if (ok) {
#pragma unroll
            for (int k = 0; k < BPT; ++k)
                if (nef[k]) {
                    const uint32_t r = tid * BPT + k, f0 = rfirst[r], f1 = rfirst[r + 1];
                    TaskDesc d;
                    d.kind = direct ? TASK_RANGE_DIRECT : TASK_RANGE;
                    d.row = row;
                    d.cut = ~0ull;
                    d.ri = 0;
                    d.m = m_row;
                    d.np = pre[f1] - pre[f0];
                    d.src = direct ? ((uint64_t)hdr[50] << 32 | hdr[49]) : sb + pre[f0];
                    const uint32_t lo = kmin + (f0 << wshift);
                    const uint64_t hi64 = (uint64_t)kmin + ((uint64_t)f1 << wshift) - 1ull;
                    const uint32_t hi = hi64 > kmax ? kmax : (uint32_t)hi64;
                    for (uint32_t j = 0; j < wgt[k]; ++j) {
                        // bit 0: first range of its row | bit 1: column sub-range, the slice holds other columns too | direct
                        // tasks: entries of the row above bit 0, first entry in `src`
                        d.first = (aux[r] + j == 0 ? 1u : 0u) | (wgt[k] > 1 ? 2u : 0u) | (direct ? (uint32_t)hdr[48] << 1 : 0u);
                        d.col_lo = lo + j * (uint32_t)TK_NOUT;
                        d.col_hi = j + 1 == wgt[k] ? hi : d.col_lo + (uint32_t)TK_NOUT - 1u;
                        d.ri = aux[r] + j;
                        d.cut = has_cuts ? cb + (uint64_t)d.ri * hdr[48] : ~0ull;   // (hdr[48]: the row's entries; none: the task searches)
                        tmp[tb + aux[r] + j] = d;
                    }
                }
        }
        if (ok && has_cuts) {
            const uint64_t ib = ((uint64_t)hdr[55] << 32) | hdr[54];
            for (uint32_t q = tid; q < hdr[56]; q += TK_BLOCK) cut_items[ib + q] = make_uint2(slot, q);
        }

#if SPADA_PRE_DBG == 2
        pre_tick(8);
#endif
        if (ok && !direct) {
            // spilled: the counts of every part become its cursors.  Layout of the row's slice: RANGE major (a range task reads one
            // contiguous slice), inside a range PART major, inside (range, part) in the order the scatter's waves arrive (by bucket
            // when the row keeps a cursor per bucket: `marks` below) -- the products a part sends to a range
            // form ONE run, and the runs of consecutive parts (which one workgroup of k_big_scatter writes one after the other) are
            // neighbours: a hub row with 10^6 products has ~500 ranges but 1024 buckets, so the runs are twice as long as
            // with one run per (bucket, part)
            __syncthreads();   // (aux: the descriptors above are written)
#pragma unroll
            for (int k = 0; k < BPT; ++k) aux[tid * BPT + k] = stf[k];
            __syncthreads();
            block_exclusive_scan4_dpp(aux, hdr + 4);
            uint32_t rng[BPT], rf[BPT];   // range of the bucket, first bucket of that range
#pragma unroll
            for (int k = 0; k < BPT; ++k) {
                rng[k] = aux[tid * BPT + k] + stf[k] - 1u;   // (bucket 0 starts a range)
                rf[k] = rfirst[rng[k]];
            }
            __syncthreads();
            uint32_t *base = cnt;   // position of the next part's run in range r, relative to the row's slice
#pragma unroll
            for (int k = 0; k < BPT; ++k) {
                const uint32_t r = tid * BPT + k;
                if (r < NR) base[r] = pre[rfirst[r]];
            }
            __syncthreads();   // (pre is read above; from here on it is the second buffer of the loop)
            // One cursor per (part, RANGE), kept at the range's first bucket; the other buckets of the range hold a mark and the number
            // of that bucket.  Nothing reads a run bucket by bucket, and a workgroup of the scatter that appends to ~P / lim runs
            // instead of up to 1024 keeps that many fewer half-written lines open in its L2.
            const bool marks = range_cursors && pre[NB] < 0x80000000u;
            // (the next part's counts are loaded while this one's cursors are formed; the barriers order LDS only: lds_barrier)
            uint4 hn = make_uint4(0u, 0u, 0u, 0u);
            uint32_t dn = 0;
            if (nreal) {
                hn = ((const uint4 *)(part_hist + (size_t)pb * NB))[tid];
                if (tid == 0) dn = parts[pb + 1].p_begin - parts[pb].p_begin;
            }
            for (uint32_t k = 0; k < nreal; ++k) {
                // e[b] = products of the part before bucket b (k_big_hist), e[NB] = all of them: in LDS for the reads at the range
                // starts; two buffers in turn, so that the next part may be written while the range bases take this one in
                uint32_t *e = (k & 1u) ? pre : aux;
                uint4 *hp = (uint4 *)(part_hist + (size_t)(pb + k) * NB) + tid;
                const uint4 h = hn;
                const uint32_t dk = dn;
                if (k + 1 < nreal) {
                    hn = hp[NB / 4];
                    if (tid == 0) dn = parts[pb + k + 2].p_begin - parts[pb + k + 1].p_begin;
                }
                e[tid * 4 + 0] = h.x;
                e[tid * 4 + 1] = h.y;
                e[tid * 4 + 2] = h.z;
                e[tid * 4 + 3] = h.w;
                if (tid == 0) e[NB] = dk;
                lds_barrier();
                uint4 c;
                c.x = base[rng[0]] + h.x - e[rf[0]];
                c.y = base[rng[1]] + h.y - e[rf[1]];
                c.z = base[rng[2]] + h.z - e[rf[2]];
                c.w = base[rng[3]] + h.w - e[rf[3]];
                if (marks) {
                    if (rf[0] != (uint32_t)tid * 4u + 0u) c.x = BX_MARK | rf[0];
                    if (rf[1] != (uint32_t)tid * 4u + 1u) c.y = BX_MARK | rf[1];
                    if (rf[2] != (uint32_t)tid * 4u + 2u) c.z = BX_MARK | rf[2];
                    if (rf[3] != (uint32_t)tid * 4u + 3u) c.w = BX_MARK | rf[3];
                }
                *hp = c;
                lds_barrier();
#pragma unroll
                for (int j = 0; j < BPT; ++j) {
                    const uint32_t r = tid * BPT + j;
                    if (r < NR) base[r] += e[rfirst[r + 1]] - e[rfirst[r]];
                }
            }
        }
        __syncthreads();
#if SPADA_PRE_DBG == 2
        pre_tick(9);
        if (tid == 0 && (slot & 31u) == 0u) {   // (a sample of the rows: the report's own atomics share a line, and a channel, with the counters the kernel allocates from)
            const unsigned long long *tk = pre_ticks();
            for (int k = 0; k < 9; ++k) atomicAdd(&ctr->dbg[k], tk[k + 1] - tk[k]);
            atomicAdd(&ctr->dbg[15], 1ull);
        }
        __syncthreads();
#endif
    }
}

// k_big_cuts: the cut table of the direct rows.  For every range ri and every entry e of such a row: the first position of the selected
// B row with a column >= the range's first column (B rows are ascending: a binary search) -- word ri * E + e of the row's table; the
// last range also writes row m, the B rows' lengths.  The (range, entry) pairs of all rows are cut into work items of BX_CUT_ITEM by
// k_big_plan, so that a hub row with hundreds of ranges is searched by hundreds of workgroups.  A direct range task then reads its entries' narrowed B rows (rows ri and
// ri + 1) in the same round trip as the entries themselves: the dozen dependent search steps that made the range tasks the slowest
// to publish their counts -- and every task behind them in the chain wait -- are done here, in parallel and before the task kernel.
// (bx of gx: the workgroup's number among those that build the table -- a launch of its own, or a share of k_after_plan's grid)
__device__ inline void big_cuts_body(const uint32_t *__restrict__ bidx, const uint64_t *__restrict__ eb0,
                                     const uint32_t *__restrict__ elen, const uint32_t *__restrict__ big_rows,
                                     const uint32_t *__restrict__ row_m, const uint32_t *__restrict__ row_tmp,
                                     const BigSlot *__restrict__ slots, const TaskDesc *__restrict__ tmp,
                                     const uint2 *__restrict__ items, uint64_t item_cap, uint32_t *__restrict__ cuts,
                                     const TaskCounters *__restrict__ ctr, uint32_t bx, uint32_t gx)
{
    if (ctr->abort_flag) return;
    unsigned long long most = 0;
    for (uint32_t a = 0; a < BX_ARENAS; ++a) most = max(most, ctr->cut_arena[a][1]);
    const unsigned long long icap = item_cap / BX_ARENAS;
    // (TWO work items per turn of a workgroup, their searches in lock step: a search is a chain of dependent loads -- a dozen round trips --
    // and the kernel is bound by their latency, not by their number: two chains in flight per thread.  Round 6, next to the single launch behind
    // the plan, where this job is the longest of the three)
    for (unsigned long long x0 = bx; x0 < most * BX_ARENAS; x0 += 2ull * gx) {
        bool on[2][BX_CUT_ITEM / 256];
        uint32_t lo[2][BX_CUT_ITEM / 256], l[2][BX_CUT_ITEM / 256], nn[2][BX_CUT_ITEM / 256], len_[2][BX_CUT_ITEM / 256];
        uint64_t b0[2][BX_CUT_ITEM / 256], dst[2][BX_CUT_ITEM / 256], dst_end[2][BX_CUT_ITEM / 256];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const unsigned long long x = x0 + (unsigned long long)it * gx;
            const uint32_t arena = (uint32_t)(x % BX_ARENAS);
            const unsigned long long k = x / BX_ARENAS;
            const bool have = x < most * BX_ARENAS && k < ctr->cut_arena[arena][1];
            uint2 item = make_uint2(0u, 0u);
            if (have) item = items[arena * icap + k];   // (row of the BIG-row list, number of the item in the row)
            const BigSlot sl = slots[item.x];
            const uint32_t row = big_rows[item.x], m = have ? row_m[row] : 0u, tb = row_tmp[row];
            const TaskDesc d0 = tmp[tb];
            const uint32_t E = d0.first >> 1;
            const uint64_t pairs = (uint64_t)m * E;
#pragma unroll
            for (int kk = 0; kk < BX_CUT_ITEM / 256; ++kk) {
                const uint64_t pr = (uint64_t)item.y * BX_CUT_ITEM + (uint32_t)kk * 256u + threadIdx.x;
                on[it][kk] = have && pr < pairs;
                lo[it][kk] = 0u;
                l[it][kk] = 0u;
                nn[it][kk] = 0u;
                len_[it][kk] = 0u;
                b0[it][kk] = 0ull;
                dst[it][kk] = dst_end[it][kk] = ~0ull;
                if (on[it][kk]) {
                    // pairs numbered ENTRY major: the lanes of a wave search ONE B row (or a few) for neighbouring ranges -- the same
                    // probes at the first steps, the same few lines at the last, where range-major numbering sent every lane to a row
                    // of its own (the table itself stays range major: a task reads rows ri and ri + 1 of it along its entries)
                    const uint32_t e = (uint32_t)(pr / m), ri = (uint32_t)(pr - (uint64_t)e * m);
                    lo[it][kk] = tmp[tb + ri].col_lo;
                    b0[it][kk] = eb0[d0.src + e];
                    len_[it][kk] = nn[it][kk] = elen[d0.src + e];
                    dst[it][kk] = sl.cut_base + (uint64_t)ri * E + e;
                    if (ri + 1 == m) dst_end[it][kk] = sl.cut_base + pairs + e;
                }
            }
        }
        for (;;) {   // (all searches of the thread in lock step: the loads of a step are independent)
            bool any = false;
            uint32_t c[2][BX_CUT_ITEM / 256];
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int kk = 0; kk < BX_CUT_ITEM / 256; ++kk) {
                    c[it][kk] = 0u;
                    if (nn[it][kk]) {
                        any = true;
                        c[it][kk] = bidx[b0[it][kk] + l[it][kk] + (nn[it][kk] >> 1)];
                    }
                }
            if (!any) break;
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int kk = 0; kk < BX_CUT_ITEM / 256; ++kk)
                    if (nn[it][kk]) {
                        const uint32_t h = nn[it][kk] >> 1;
                        if (c[it][kk] < lo[it][kk]) {
                            l[it][kk] += h + 1;
                            nn[it][kk] -= h + 1;
                        } else {
                            nn[it][kk] = h;
                        }
                    }
        }
        // (one 4-byte store per pair at a stride of E words.  Measured, kernel alone on R-MAT 16 / 18: 434 / 3308 us; with the table
        // written in the order of the searches -- coalesced, wrong -- 320 / 1990 us; with the B rows staged in LDS, the searches there
        // and the results through an LDS tile in the table's order, items of 1024 pairs: R-MAT 18 -10 % on the phase, R-MAT 16 and
        // the web input +10 ... 20 % -- the per-item staging costs rows of a few hundred pairs more than it saves: not kept)
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int kk = 0; kk < BX_CUT_ITEM / 256; ++kk)
                if (on[it][kk]) {
                    cuts[dst[it][kk]] = l[it][kk];
                    if (dst_end[it][kk] != ~0ull) cuts[dst_end[it][kk]] = len_[it][kk];
                }
    }
}
__global__ __launch_bounds__(256) void k_big_cuts(const uint32_t *__restrict__ bidx, const uint64_t *__restrict__ eb0,
                                                  const uint32_t *__restrict__ elen, const uint32_t *__restrict__ big_rows,
                                                  const uint32_t *__restrict__ row_m, const uint32_t *__restrict__ row_tmp,
                                                  const BigSlot *__restrict__ slots, const TaskDesc *__restrict__ tmp,
                                                  const uint2 *__restrict__ items, uint64_t item_cap, uint32_t *__restrict__ cuts,
                                                  const TaskCounters *__restrict__ ctr)
{
    big_cuts_body(bidx, eb0, elen, big_rows, row_m, row_tmp, slots, tmp, items, item_cap, cuts, ctr, blockIdx.x, gridDim.x);
}

template <int U = FLAT_U>
__device__ inline void big_scatter_body(const double *__restrict__ aval, const uint32_t *__restrict__ bidx,
                                        const double *__restrict__ bval, const uint64_t *__restrict__ eb0,
                                        const uint32_t *__restrict__ elen, const uint32_t *__restrict__ big_rows,
                                        const uint32_t *__restrict__ row_kmin, const uint32_t *__restrict__ row_kmax,
                                        const BigPart *__restrict__ parts, const uint32_t *__restrict__ part_hist,
                                        const BigSlot *__restrict__ slots, uint32_t *__restrict__ scr_col,
                                        double *__restrict__ scr_val, uint32_t *__restrict__ scr_seq /* may be null */,
                                        uint32_t psh, const uint32_t *__restrict__ spill_parts, TaskCounters *__restrict__ ctr, uint32_t bx, uint32_t gx)
{
    constexpr int NB = BX_NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    uint32_t *cur = (uint32_t *)(smem + 256);
    uint32_t *s_re = cur + NB;
    uint64_t *s_a0 = (uint64_t *)(s_re + 4);
    unsigned char *scratch = (unsigned char *)(s_a0 + 2);
    const int tid = threadIdx.x;
    if (ctr->abort_flag || ctr->n_spilled == 0) return;
    // (round 6: the tickets run over the list of the SPILLED rows' parts k_big_plan has left -- on the web input 120 of 4 000 part records:
    // a workgroup no longer spends its first tickets on the records of direct rows.  Neighbouring entries of the list are neighbouring
    // parts of one row, as before)
    const uint32_t nparts = ctr->n_spill_parts;
    // a workgroup takes BX_RUN consecutive records: the parts of one row (or of neighbouring rows), whose scattered stores fall into
    // the same lines of the row's scratch slice, go through one CU and one L2 one after the other
    // (measured: giving every XCD a contiguous eighth of the records, so that neighbouring runs meet in one L2, is 3 - 8 % SLOWER)
    // (runs only when many rows are spilled: the parts of a few dozen rows -- the web input's rows with more than 512 entries --
    // are better spread over as many workgroups than done eight in a row by one)
    const uint32_t run = ctr->n_spilled > gx / 8u ? max(BX_RUN >> (psh - BX_PART_SHIFT), 1u) : 1u;   // (~64 K products per run)
    // (the runs are handed out by ticket: the parts of direct rows are skipped at once, those of spilled rows are a walk of tens of
    // thousands of products -- a fixed stride left the workgroups whose runs held mostly direct rows idle at the end)
    __shared__ uint32_t s_run0;
    for (;;) {
    __syncthreads();
    if (tid == 0) s_run0 = atomicAdd(&ctr->scatter_next[(bx % SCATTER_NQ) * 32u], 1u) * SCATTER_NQ + bx % SCATTER_NQ;
    __syncthreads();
    if ((unsigned long long)s_run0 * run >= nparts) break;
    const uint32_t pi0 = s_run0 * run;
    for (uint32_t li = pi0; li < min(pi0 + run, nparts); ++li) {
        const uint32_t pi = spill_parts[li];
        const BigPart pt = parts[pi];
        if (pt.slot == BX_NOPART) continue;   // (uniform over the workgroup, like the next one)
        const BigSlot sl = slots[pt.slot];
        if (!sl.ok || sl.direct) continue;
        const uint32_t e_count = (uint32_t)(parts[pi + 1].e_begin - pt.e_begin);
        const uint32_t row = big_rows[pt.slot];
        const uint32_t kmin = row_kmin[row], wshift = big_wshift(kmin, row_kmax[row]);
        ((uint4 *)cur)[tid] = ((const uint4 *)(part_hist + (size_t)pi * NB))[tid];
        if (tid == 0) {
            s_re[0] = 0;
            s_re[1] = e_count;
            s_a0[0] = pt.e_begin;
        }
        __syncthreads();
        const uint64_t sb = sl.scr_base;
        const bool marks = (sl.ok & 2u) != 0;
        flat_walk<TK_BLOCK, TK_EPT, 1, true, U>(s_re, s_a0, 1u, e_count, eb0, elen, aval, bidx, bval, scratch, hdr,
                                                [&](uint32_t(&col)[U], uint32_t(&plr)[U], double(&v)[U], uint32_t(&pp)[U]) {
#pragma unroll
                                                    for (int u = 0; u < U; ++u) {
                                                        uint32_t bk = plr[u] != LR_NONE ? (col[u] - kmin) >> wshift : 0xFFFFFFFFu;
                                                        if (marks && bk != 0xFFFFFFFFu) {   // (the marks never change; a cursor stays below 2^31)
                                                            const uint32_t x = cur[bk];
                                                            if (x & BX_MARK) bk = x & (uint32_t)(NB - 1);
                                                        }
                                                        bool head;
                                                        uint32_t len, hl;
                                                        const bool runs = wave_runs(bk, head, len, hl);
                                                        if (!runs) {
                                                            len = 1u;
                                                            hl = threadIdx.x & 63;
                                                        }
                                                        uint32_t pbase = 0;
                                                        if (head && bk != 0xFFFFFFFFu) pbase = atomicAdd(&cur[bk], len);
                                                        if (runs) pbase = (uint32_t)__shfl((int)pbase, (int)hl);
                                                        if (plr[u] != LR_NONE) {
#ifdef SPADA_SCATTER_SEQ   // measurement build only (scripts/dev/scatter_seq.sh): the stores in walk order -- coalesced, and wrong
                                                            const uint32_t p = pt.p_begin + pp[u] + 0u * pbase;
#else
                                                            const uint32_t p = pbase + ((threadIdx.x & 63) - hl);
#endif
                                                            // (plain stores: the runs of a range are completed in the caches;
                                                            // non-temporal ones made the stage 1.4 - 2 x slower)
                                                            scr_col[sb + p] = col[u];
                                                            scr_val[sb + p] = v[u];
                                                            if (scr_seq) scr_seq[sb + p] = pt.p_begin + pp[u];
                                                        }
                                                    }
                                                });
        __syncthreads();
    }
    }
#ifdef SPADA_SCATTER_SEQ
    if (threadIdx.x == 0 && bx == 0) atomicOr(&ctr->abort_flag, 256u);   // nothing may read this scratch
#endif
}
// (`ctr` is written -- the runs' tickets: not const)
__global__ __launch_bounds__(TK_BLOCK) void k_big_scatter(const double *__restrict__ aval, const uint32_t *__restrict__ bidx,
                                                          const double *__restrict__ bval, const uint64_t *__restrict__ eb0,
                                                          const uint32_t *__restrict__ elen, const uint32_t *__restrict__ big_rows,
                                                          const uint32_t *__restrict__ row_kmin, const uint32_t *__restrict__ row_kmax,
                                                          const BigPart *__restrict__ parts, const uint32_t *__restrict__ part_hist,
                                                          const BigSlot *__restrict__ slots, uint32_t *__restrict__ scr_col,
                                                          double *__restrict__ scr_val, uint32_t *__restrict__ scr_seq /* may be null */,
                                                          uint32_t psh, const uint32_t *__restrict__ spill_parts, TaskCounters *__restrict__ ctr)
{
    big_scatter_body(aval, bidx, bval, eb0, elen, big_rows, row_kmin, row_kmax, parts, part_hist, slots, scr_col, scr_val, scr_seq, psh, spill_parts, ctr,
                     blockIdx.x, gridDim.x);
}


}  // namespace spada
