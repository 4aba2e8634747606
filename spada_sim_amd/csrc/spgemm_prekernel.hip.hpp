// gfx950 kernels of the task pipeline, part 3 of 5: THE STAGE IN FRONT OF THE TASK KERNEL -- entry descriptors and row statistics (k_entry_stats),
// row classes and the cut of the tiles into batches (k_row_class_cut), the ONE launch behind the BIG-row plan (k_after_plan: scatter + cut table +
// task list), and the positions behind a counting run (k_pos1-4).  What it replaces: Scheduler::next_block / next_window (scheduler.rs:296-379,
// :482-606) as a device-built task list.
#pragma once
#include "spgemm_bigrow.hip.hpp"

namespace spada {

// ---- 1. entry descriptors + row statistics ---------------------------------------------------------------------------------
// k_entry_stats: one lane per A entry, 64 consecutive entries per wave and round, whatever the row lengths are (the row of an
// entry comes from A.rowid).  Per entry: the irregular gathers of the path, done exactly once -- the 16-byte B.indptr pair -> eb0 /
// elen (begin, length of the selected B row) and the 8-byte extent of that row (first / last column: spada_dev_csr::rext, kept with
// the matrix).  Entries of one row are adjacent lanes: a segmented wave scan adds them up, and the last lane of every run adds the
// run to the row's totals (row_P, row_kmin, row_kmax, preset to 0 / max / 0) with one device atomic each -- ~1 atomic triple per
// row, none of them contended.
// k_row_class_cut (section 3): one lane per row: class, statistics, the list of BIG rows, and the cut of the row's tile.  (The row's
// accumulators are put back to their presets for the next run by k_preset_rows, behind the end of the run where nobody waits.)
// (Measured and not kept, round 5: both kernels as ONE, a workgroup per tile of 1024 rows walking the tile's entries with the row
// totals in LDS -- no device atomics, no accumulators in HBM: correct, and 4.5 x SLOWER on the web input (0.297 against 0.066 ms): the
// entries of a tile range from 600 to 67 000, and a matrix with few rows (R-MAT 16: 64 tiles) does not fill the GPU at all.  The
// walk has to be balanced over ENTRIES.)
__global__ __launch_bounds__(256) void k_clear_counters(TaskCounters *__restrict__ ctr)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < sizeof(TaskCounters) / 8; i += gridDim.x * 256) ((unsigned long long *)ctr)[i] = 0ull;
}
// The counters of a finished run written straight into pinned host memory, then a sequence number the host polls: what the host waits for at
// the end of a call is this store becoming visible -- no copy command, no event, no wake-up through the runtime
__global__ __launch_bounds__(256) void k_export_counters(const TaskCounters *__restrict__ src, TaskCounters *__restrict__ host_dst,
                                                         unsigned long long *__restrict__ host_seq, unsigned long long seq)
{
    for (uint32_t i = threadIdx.x; i < sizeof(TaskCounters) / 8; i += 256) ((unsigned long long *)host_dst)[i] = ((const unsigned long long *)src)[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (the first run of a context, and a run over more rows than any before it)
// (behind a run the same launch clears the counter set the run after the next one will use -- `clr`: one launch fewer between two calls)
__global__ __launch_bounds__(256) void k_preset_rows(unsigned long long *__restrict__ row_P, uint32_t *__restrict__ row_kmin,
                                                     uint32_t *__restrict__ row_kmax, uint64_t n, TaskCounters *__restrict__ clr)
{
    if (clr)
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < sizeof(TaskCounters) / 8; i += gridDim.x * 256) ((unsigned long long *)clr)[i] = 0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        row_P[i] = 0ull;
        row_kmin[i] = 0xFFFFFFFFu;
        row_kmax[i] = 0u;
    }
}

// k_place_probe: the scatter's store pattern on the scratch arrays -- runs of 12 products (a 4-byte column and an 8-byte value at the same
// product number) at scattered places, five runs per wave and turn.  What it takes depends on WHERE the two arrays lie in physical memory: 2.5 or 3.3 ms
// on R-MAT 18's arrays, and the scatter phase of every run on those arrays takes 7.9 or 8.8 ms accordingly (spada_engine.hip, place_scratch).
__global__ __launch_bounds__(256) void k_place_probe(uint32_t *__restrict__ col, double *__restrict__ val, unsigned long long nprod, uint32_t per_wave)
{
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long x = ((unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 0x9E3779B97F4A7C15ull + 777ull;
    for (uint32_t i = 0; i < per_wave; ++i) {
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        const unsigned long long r = (x + (lane / 12u) * 0x51ED27ull * (x | 1ull)) % (nprod - 16ull) + lane % 12u;
        if (lane < 60u) {
            col[r] = (uint32_t)x;
            val[r] = (double)i;
        }
    }
}

constexpr int ENTRY_STATS_U = 1;   // segments of 64 entries per wave and turn (the engine sizes the grid by it)
template <class ARGS>
__global__ __launch_bounds__(256) void k_entry_stats(const uint64_t *__restrict__ aptr, const uint32_t *__restrict__ aidx,
                                                     const uint32_t *__restrict__ arow, const uint64_t *__restrict__ bptr,
                                                     const uint2 *__restrict__ bext, uint64_t r0, uint32_t nrows,
                                                     uint64_t *__restrict__ eb0, uint32_t *__restrict__ elen,
                                                     unsigned long long *__restrict__ row_P, uint32_t *__restrict__ row_kmin,
                                                     uint32_t *__restrict__ row_kmax, uint32_t limit, TaskCounters *__restrict__ ctr,
                                                     const ARGS g, ARGS *__restrict__ g_dst)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctr->prod_limit = limit;   // (products a task hashes at most: read by the kernels behind this one)
        if (g_dst) *g_dst = g;     // (the arguments of the task kernel travel with the first kernel of the run: see k_task_args)
    }
    const uint64_t e0 = aptr[r0], e1 = aptr[r0 + nrows];
    const int lane = threadIdx.x & 63;
    // (round 6 measured TWO segments of 64 entries per wave and turn with the extent loaded beside the indptr pair -- one dependent round trip
    // less, twice the loads in flight: level on the web input, 2 us SLOWER on the small ones (half the workgroups): not kept)
    for (uint64_t q0 = e0 + (uint64_t)blockIdx.x * 256 + (threadIdx.x & ~63); q0 < e1; q0 += (uint64_t)gridDim.x * 256) {
        const uint64_t q = q0 + lane;
        uint32_t row = 0xFFFFFFFFu, mn = 0xFFFFFFFFu, mx = 0;
        unsigned long long len = 0;
        if (q < e1) {
            const uint32_t k = aidx[q];
            row = arow[q] - (uint32_t)r0;
            const uint64_t b0 = bptr[k], b1 = bptr[k + 1];
            eb0[q] = b0;
            len = b1 - b0;
            elen[q] = (uint32_t)len;
            if (b1 > b0) {   // (first / last column of the selected B row: one 8-byte gather, spada_dev_csr::rext)
                const uint2 ex = bext[k];
                mn = ex.x;
                mx = ex.y;
            }
        }
        // segmented inclusive scan over runs of equal row
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t r2 = __shfl_up(row, o);
            const unsigned long long l2 = __shfl_up(len, o);
            const uint32_t n2 = __shfl_up(mn, o), x2 = __shfl_up(mx, o);
            if (lane >= o && r2 == row) {
                len += l2;
                mn = min(mn, n2);
                mx = max(mx, x2);
            }
        }
        const uint32_t rnext = __shfl_down(row, 1);
        const bool tail = row != 0xFFFFFFFFu && (lane == 63 || rnext != row);
        if (tail && len) {
            atomicAdd(&row_P[row], len);
            atomicMin(&row_kmin[row], mn);
            atomicMax(&row_kmax[row], mx);
        }
    }
}

// (the row classes: k_row_class_cut, with the cut of the tiles -- section 3)


// ---- 3. the cut: rows -> tasks in output order ---------------------------------------------------------------------------
// Tiles of CUT_TILE consecutive rows.  A BIG row is row_m[i] range tasks of its own.  The other rows are packed greedily, in
// row order, into batches that are as full as the table allows: a batch is a maximal run of rows with at most `limit`
// products to hash, at most BT_PMAX products in all (the products of COPY rows never touch the table, but like the hashed ones
// they wait in the registers of the task for their position), at most BT_EMAX A entries and at most `rmax` rows.
// Full batches mean fewer tasks and -- what matters to the chain -- tasks of equal length.  nxt[i] (first row
// after a batch that starts at row i) is found for all rows in parallel by binary search over the tile's prefix sums; the
// starts are what the walks along nxt reach (pointer doubling; batches do not cross tiles).
constexpr uint32_t CUT_FOLD_TILES = 2048;   // (k_cut3 adds up the tile counts itself up to here: O(tiles^2) words read in all)

__device__ inline uint32_t block_scan_excl_u32(uint32_t v, uint32_t *s_w /*[4]*/, uint32_t *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    uint32_t add = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < w) add += s_w[k];
        tot += s_w[k];
    }
    *total = tot;
    return inc - v + add;
}

struct CutRow {
    uint32_t t[CUT_ITEMS];      // tasks started by the row
    uint32_t kind[CUT_ITEMS];   // 0 none, 1 batch start, 2 BIG
};
constexpr uint32_t CUT_END = 0xFFFFFFFFu;
// (k_cut3's arrays over the tile's BIG rows; through round 5 the struct also held the prefix sums of the cut, which k_row_class_cut now keeps
// in a struct of its own: 12 KB instead of 21.5 -- what every workgroup of k_after_plan's launch is given, eight to a CU instead of seven)
struct CutLds {
    uint32_t pc[CUT_TILE + 1], pw[CUT_TILE + 1];
    uint32_t nxt[CUT_TILE];
    uint32_t s_w[4];
};

// batch descriptor word (TaskDesc::np of a TASK_BATCH): rows | A entries << 8 | products (hashed + copied) << 18 | DENSE << 31: the
// column spans of its hashed rows, in blocks, fit the table one slot per block
constexpr uint32_t BINFO_DENSE = 1u << 31;
__host__ __device__ inline uint32_t batch_info(uint32_t R, uint32_t E, uint32_t P) { return R | (E << 8) | (P << 18); }
static_assert(TK_RMAX <= 255 && BT_EMAX <= 1023 && BT_PMAX <= 4095, "batch_info fields");

// k_row_class_cut: the class of every row (by its products P_i and its length), the list of the BIG rows, the statistics -- and the cut of
// its tile of CUT_TILE rows: tasks started by every row -> row_t (0: none, else 1) and the tile's total.  A BIG row starts no task HERE:
// k_big_plan, which knows its ranges, adds them to row_t and to the tile's total (through round 4 the classes and the cut were two
// kernels with the BIG-row stage between them: a launch, its drain and the re-read of the row words on the critical path of every call,
// for a cut that needs nothing the BIG-row kernels write).  (statistics spread over CLS_SLOTS lines: the host sums them)
// (estimates for the first run's workspaces)
__device__ inline unsigned long long est_ranges(unsigned long long P, uint32_t lim, uint32_t kmin, uint32_t kmax)
{
    // (exactly the descriptors k_big_parts sets aside for the row: big_max_ranges)
    const uint32_t P32 = P > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)P;
    return big_max_ranges(P32, lim, (1ull << big_wshift(kmin, kmax)) > (unsigned long long)TK_NOUT);
}
constexpr int RCC_WAVES = CUT_TILE / 64;
struct ClassCutLds {
    uint4 pre[CUT_TILE + 1];    // exclusive prefix sums over the tile's rows: products to hash, products to copy, A entries, blocks of columns
                                // between the first and last column of the hashed rows
    uint32_t nxt[CUT_TILE];
    uint8_t mark[CUT_TILE];
    uint32_t slot[5][RCC_WAVES];   // the scans' wave totals (scan_part / scan_done)
    uint32_t first[RCC_WAVES];     // first batch start of every wave
    unsigned long long rows[N_CLS], prod[N_CLS], entries, est[3];
    uint32_t nbig, big_base, live[2];
};
static_assert(RCC_WAVES == 16, "k_row_class_cut: one row per thread, sixteen waves");
// ONE ROW PER THREAD, workgroups of CUT_TILE = 1024 threads.  Through round 5 (and most of round 6) a workgroup of 256 threads took four
// rows per thread: the kernel is one workgroup's chain of dependent steps -- every tile is resident at once, the kernel takes as long as
// its slowest workgroup -- and with one wave per SIMD every instruction of that chain waited for the one before it (phase clocks of a
// build with SPADA_PRE_DBG: 56 k ticks per workgroup on mc2depi -- 23 of the kernel's 28 us -- of which 13 k in the four binary searches
// of a thread and 10 k in the pointer doubling: neither got faster when the searches were interleaved or the row words passed through
// LDS).  Sixteen waves do the same work four to a SIMD, and a row's values stay in its thread's registers from the classes to the cut.
// The statistics are reduced on DPP inside the waves (eleven 64-bit shuffled reductions before); the BIG rows get their places in
// the list from one device atomic per WORKGROUP whose answer is not needed before the kernel's end; the pointer doubling stops with the
// round in which no pointer is left.
__global__ __launch_bounds__(CUT_TILE) void k_row_class_cut(const uint64_t *__restrict__ aptr, uint64_t r0, uint32_t nrows, uint32_t rmax,
                                                            const unsigned long long *__restrict__ row_P, const uint32_t *__restrict__ row_kmin,
                                                            const uint32_t *__restrict__ row_kmax, uint32_t *__restrict__ row_nprod,
                                                            uint8_t *__restrict__ row_cls, uint32_t *__restrict__ row_cl,
                                                            RowRec *__restrict__ row_rec, uint32_t *__restrict__ row_m,
                                                            uint32_t *__restrict__ big_rows, TaskCounters *__restrict__ ctr,
                                                            uint32_t *__restrict__ tile_tasks, uint32_t *__restrict__ row_t,
                                                            uint32_t *__restrict__ row_binfo, uint32_t want_est /* the run reads its statistics back mid-run: slots 11 - 14 */)
{
    pre_tick(0);
    const uint32_t lim = ctr->prod_limit;
    __shared__ ClassCutLds L;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t tile_base = blockIdx.x * CUT_TILE, cnt = min((uint32_t)CUT_TILE, nrows - tile_base);
    const uint32_t li = tid, i = tile_base + li;
    const bool in = li < cnt;
    if (tid < N_CLS) L.rows[tid] = L.prod[tid] = 0;
    if (tid < 3) L.est[tid] = 0;
    if (tid == 0) {
        L.entries = 0;
        L.nbig = 0;
        L.live[0] = L.live[1] = 0;
    }
    const unsigned long long P = in ? row_P[i] : 0ull;
    const uint64_t a0 = in ? aptr[r0 + i] : 0ull, a1 = in ? aptr[r0 + i + 1] : 0ull;
    const uint32_t kmin = in ? row_kmin[i] : 0u, kmax = in ? row_kmax[i] : 0u;
    __syncthreads();   // (the LDS sums are cleared)
    // ---- class, row words, statistics, the BIG rows counted ----
    const uint32_t L_ = (uint32_t)(a1 - a0);
    const uint8_t cls = in ? row_class(P, L_, rmax, lim) : (uint8_t)CLS_EMPTY;
    const uint32_t P32 = P > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)P;
    uint32_t big_loc = 0xFFFFFFFFu;
    if (in) {
        row_nprod[i] = P32;
        row_cls[i] = cls;
        row_cl[i] = (uint32_t)cls | (min(L_, 0x1FFFFFFFu) << 3);
        row_rec[i] = RowRec{kmin, kmax, P32, (uint32_t)cls};
        row_m[i] = 0;
        if (cls == CLS_BIG) {   // (few rows: LDS atomics of their own)
            const unsigned long long m_est = est_ranges(P, lim, kmin, kmax);
            atomicAdd(&L.est[0], m_est);
            if (L_ <= BT_EMAX) atomicAdd(&L.est[1], (m_est + 1ull) * L_);
            else atomicAdd(&L.est[2], P);
            atomicAdd(&L.prod[CLS_BIG], P);
        }
        if (L_ > BT_EMAX) atomicAdd(&L.entries, (unsigned long long)L_);   // (BIG, or EMPTY with a long row: few)
    }
    {
        // rows per class (COPY .. BIG: a byte each; EMPTY = the rest of the wave's rows), products of the classes a batch takes (at most
        // BT_PMAX a row), entries of the rows a chunk holds: five 32-bit sums per wave
        const unsigned long long inm = __ballot(in);
        const uint32_t r4 = wave_scan_incl_u32(in && cls != CLS_EMPTY ? 1u << (8 * (cls - 1)) : 0u);
        const uint32_t p1 = wave_scan_incl_u32(cls == CLS_COPY ? P32 : 0u), p2 = wave_scan_incl_u32(cls == CLS_SMALL ? P32 : 0u),
                       p3 = wave_scan_incl_u32(cls == CLS_SOLO ? P32 : 0u), en = wave_scan_incl_u32(in && L_ <= BT_EMAX ? L_ : 0u);
        if (lane == 63u && inm) {
            const uint32_t nz = (r4 & 0xFFu) + ((r4 >> 8) & 0xFFu) + ((r4 >> 16) & 0xFFu) + (r4 >> 24);
            const uint32_t ne = (uint32_t)__popcll(inm) - nz;
            if (ne) atomicAdd(&L.rows[CLS_EMPTY], (unsigned long long)ne);
#pragma unroll
            for (int k = 1; k < N_CLS; ++k)
                if ((r4 >> (8 * (k - 1))) & 0xFFu) atomicAdd(&L.rows[k], (unsigned long long)((r4 >> (8 * (k - 1))) & 0xFFu));
            if (p1) atomicAdd(&L.prod[CLS_COPY], (unsigned long long)p1);
            if (p2) atomicAdd(&L.prod[CLS_SMALL], (unsigned long long)p2);
            if (p3) atomicAdd(&L.prod[CLS_SOLO], (unsigned long long)p3);
            if (en) atomicAdd(&L.entries, (unsigned long long)en);
        }
        const unsigned long long bm = __ballot(in && cls == CLS_BIG);
        if (bm) {   // BIG rows: places inside the workgroup's share of the list (one LDS atomic per wave)
            uint32_t base = 0;
            if (lane == (uint32_t)__ffsll((long long)bm) - 1u) base = atomicAdd(&L.nbig, (uint32_t)__popcll(bm));
            base = __shfl(base, __ffsll((long long)bm) - 1);
            if (in && cls == CLS_BIG) big_loc = base + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull));
        }
    }
    pre_tick(1);
    // what the cut needs of the row: products to hash (a BIG row, or an EMPTY row with more entries than a chunk holds -- `fat`: a batch
    // of its own that has nothing to do -- closes every batch: lim + 1), products to copy, entries, blocks of columns a table addressed
    // by column would need
    const bool fat = cls == CLS_EMPTY && L_ > BT_EMAX, hashed = cls == CLS_SMALL || cls == CLS_SOLO;
    uint4 v;
    v.x = (cls == CLS_BIG || fat) ? lim + 1 : (hashed ? P32 : 0u);
    v.y = cls == CLS_COPY ? P32 : 0u;
    v.z = (cls == CLS_BIG || fat) ? 0u : min(L_, 0x1FFFFFFFu);
    v.w = hashed ? min((kmax >> BT_DSHIFT) - (kmin >> BT_DSHIFT) + 1u, 2u * BT_T) : 0u;
    const uint32_t ix = scan_part(v.x, L.slot[0], tid), iy = scan_part(v.y, L.slot[1], tid), iz = scan_part(v.z, L.slot[2], tid),
                   iw = scan_part(v.w, L.slot[3], tid);
    __syncthreads();
    pre_tick(2);
    // (the share of the BIG-row list: asked for here, needed at the kernel's end)
    uint32_t big_base = 0;
    if (tid == 0 && L.nbig) big_base = atomicAdd(&ctr->n_big, L.nbig);
    unsigned long long *part = ctr->cls_part[blockIdx.x % CLS_SLOTS];
    if (tid < N_CLS && L.rows[tid]) {
        atomicAdd(&part[tid], L.rows[tid]);
        if (L.prod[tid]) atomicAdd(&part[N_CLS + tid], L.prod[tid]);
    }
    if (tid == 64 && L.entries) atomicAdd(&part[2 * N_CLS], L.entries);
    if (want_est && tid >= 128 && tid < 131 && L.est[tid - 128]) atomicAdd(&part[12 + tid - 128], L.est[tid - 128]);
    uint4 me;   // exclusive prefix sums of the row
    {
        uint32_t tot;
        me.x = scan_done<RCC_WAVES>(ix, v.x, L.slot[0], &tot, tid);
        me.y = scan_done<RCC_WAVES>(iy, v.y, L.slot[1], &tot, tid);
        me.z = scan_done<RCC_WAVES>(iz, v.z, L.slot[2], &tot, tid);
        me.w = scan_done<RCC_WAVES>(iw, v.w, L.slot[3], &tot, tid);
        L.pre[li] = me;
        if (tid == CUT_TILE - 1) L.pre[CUT_TILE] = make_uint4(me.x + v.x, me.y + v.y, me.z + v.z, me.w + v.w);
    }
    pre_tick(3);
    __syncthreads();
    pre_tick(4);
    // nxt[i]: largest j <= cnt with pc[j] - pc[i] <= lim, (pc + pw)[j] - (pc + pw)[i] <= BT_PMAX, pe[j] - pe[i] <= BT_EMAX, j - i <= rmax
    // (j = i + 1 is always feasible: a row that is not BIG fits a batch by its class)
    {
        uint32_t nx = li + 1;
        if (in && cls != CLS_BIG) {
            const uint32_t limc = me.x + lim, limp = me.x + me.y + BT_PMAX, lime = me.z + BT_EMAX;
            uint32_t lo = li + 1, hi = min(cnt, li + rmax);   // invariant: lo is feasible
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                const uint4 pm = L.pre[mid];
                if (pm.x <= limc && pm.x + pm.y <= limp && pm.z <= lime) lo = mid;
                else hi = mid - 1;
            }
            nx = lo;
        }
        // (BIG rows and the end of the tile stop a walk: they point nowhere)
        L.nxt[li] = (in && cls != CLS_BIG && nx < cnt) ? nx : CUT_END;
        // batch starts, to begin with: BIG rows (tasks of their own), and the first row of every run of non-BIG rows -- the tile's
        // first row, or the row after a BIG row
        const uint32_t c_prev = __shfl_up(v.x, 1);   // (lane 0: the row in front belongs to the wave before)
        const uint32_t before = lane ? c_prev : (li ? me.x - L.pre[li - 1].x : 0u);
        L.mark[li] = in && (cls == CLS_BIG || li == 0 || before > lim) ? 1 : 0;
    }
    __syncthreads();
    pre_tick(5);
    // ... then every row that a walk along nxt reaches from such a start.  Walked by pointer doubling (round k marks what lies
    // 2^k hops behind a marked row, then squares the pointers): log2(batches of the tile) rounds for all runs at once, where one thread
    // per run following the pointers took up to a tile's worth of dependent LDS reads (the cut of cop20k_A: 71 -> 30 us)
    for (uint32_t span = 1, round = 0; span < cnt; span <<= 1, ++round) {
        const uint32_t j1 = in ? L.nxt[li] : CUT_END;
        const uint32_t j2 = j1 != CUT_END ? L.nxt[j1] : CUT_END;
        if (j1 != CUT_END && L.mark[li]) L.mark[j1] = 1;
        if (__ballot(j2 != CUT_END) && lane == 0) L.live[round & 1u] = 1u;
        __syncthreads();
        if (in) L.nxt[li] = j2;
        if (tid == 0) L.live[(round & 1u) ^ 1u] = 0u;   // (the next round's flag: last read before the barrier above)
        __syncthreads();
        if (!L.live[round & 1u]) break;   // (no pointer is left: nothing more can be marked)
    }
    pre_tick(6);
    // where the batch that starts at a row ends: the next start behind it (or the end of the tile)
    const bool start = in && L.mark[li] != 0;
    const unsigned long long sm = __ballot(start);
    if (lane == 0) L.first[wave] = sm ? wave * 64u + (uint32_t)__ffsll((long long)sm) - 1u : CUT_END;
    // (the tile's tasks: a BIG row starts none HERE -- k_big_plan adds its ranges)
    const bool batch = start && cls != CLS_BIG;
    const unsigned long long bq = __ballot(batch);
    if (lane == 0) L.slot[4][wave] = (uint32_t)__popcll(bq);
    if (tid == 0) L.big_base = big_base;   // (the device atomic's answer, asked for long ago)
    __syncthreads();
    uint32_t binfo = 0;
    if (batch) {
        const unsigned long long later = lane == 63u ? 0ull : sm & ~((2ull << lane) - 1ull);
        uint32_t end = later ? wave * 64u + (uint32_t)__ffsll((long long)later) - 1u : CUT_END;
        for (uint32_t k = wave + 1; k < (uint32_t)RCC_WAVES && end == CUT_END; ++k) end = L.first[k];
        if (end == CUT_END) end = cnt;
        const uint4 pe = L.pre[end];
        binfo = fat ? batch_info(1u, 0u, 0u) : batch_info(end - li, pe.z - me.z, (pe.x - me.x) + (pe.y - me.y));
        if (!fat && pe.x > me.x && pe.w - me.w <= BT_T) binfo |= BINFO_DENSE;
    }
    pre_tick(7);
    if (tid == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int k = 0; k < RCC_WAVES; ++k) tot += L.slot[4][k];
        tile_tasks[blockIdx.x] = tot;
        if (want_est && tot) atomicAdd(&part[11], (unsigned long long)tot);
    }
    if (in) {
        row_t[i] = batch ? 1u : 0u;
        row_binfo[i] = binfo;
    }
    if (big_loc != 0xFFFFFFFFu) big_rows[L.big_base + big_loc] = i;
#if SPADA_PRE_DBG == 1
    pre_tick(8);
    if (threadIdx.x == 0) {
        const unsigned long long *tk = pre_ticks();
        for (int k = 0; k < 8; ++k) atomicAdd(&ctr->dbg[k], tk[k + 1] - tk[k]);
        atomicAdd(&ctr->dbg[15], 1ull);
    }
#endif
}

// single workgroup: exclusive scan of the tile totals in place; total -> ctr->ntasks
__global__ __launch_bounds__(256) void k_cut2(uint32_t *__restrict__ tile_tasks, uint32_t ntiles, uint32_t task_cap,
                                              TaskCounters *__restrict__ ctr)
{
    __shared__ uint32_t s_w[4];
    uint32_t carry = 0;
    for (uint32_t b = 0; b < ntiles; b += 256) {
        const uint32_t i = b + threadIdx.x;
        const uint32_t v = i < ntiles ? tile_tasks[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_scan_excl_u32(v, s_w, &tot);
        if (i < ntiles) tile_tasks[i] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ctr->ntasks = carry;
        ctr->need_tasks = carry;
        if (carry > task_cap) atomicOr(&ctr->abort_flag, 2u);
    }
}

// k_cut3: task descriptors at their final place: tile offset + prefix of row_t inside the tile.  gridDim.y workgroups share a tile:
// each works out the tile's layout, workgroup y = 0 writes the batch tasks, and the range descriptors of the tile's BIG rows are
// copied by all of them (a chunk of R-MAT 22: 95 tiles with 11 000 descriptors each -- 4.5 ms on 95 workgroups)
// (tile bx of gx, share by of gy of the tile: blockIdx / gridDim of a launch of its own, or a share of k_after_plan's grid)
__device__ inline void cut3_body(const uint8_t *__restrict__ row_cls, const uint32_t *__restrict__ row_t,
                                 const uint32_t *__restrict__ row_binfo, const uint64_t *__restrict__ aptr, uint64_t r0,
                                 const uint32_t *__restrict__ row_tmp, uint32_t n,
                                 uint32_t *__restrict__ tile_tasks, const TaskDesc *__restrict__ tmp,
                                 TaskDesc *__restrict__ tasks, uint32_t task_cap, uint32_t fold /* no k_cut2 has run */,
                                 uint32_t *__restrict__ tile_first, uint32_t *__restrict__ legacy,
                                 unsigned long long *__restrict__ status /* the chain's status words: cleared with the task they belong to */,
                                 uint32_t scatter_launched, TaskCounters *__restrict__ ctr, uint32_t bx, uint32_t gx, uint32_t by, uint32_t gy)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    CutLds &L = *reinterpret_cast<CutLds *>(smem);
    const bool lead_wg = by == 0;
    // (the engine leaves k_big_scatter out when the context's previous run spilled no row; should the plan of THIS run have spilled some,
    // the run is stopped here -- flag 64: the task kernel returns at once -- before any task can walk a scratch slice nobody filled)
    if (!scatter_launched && bx == 0 && by == 0 && threadIdx.x == 0 && ctr->n_spilled != 0) atomicOr(&ctr->abort_flag, 64u);
    // first task of the tile.  Up to CUT_FOLD_TILES tiles every workgroup adds up the counts of the tiles before its own itself (a
    // few KB of L2-resident words) and the one-workgroup scan kernel in front of k_cut3 is not launched: one launch and its
    // gap less on the critical path of every call (~7 us; what matters once a GPU holds an eighth of the rows).  tile_first keeps the
    // result for k_pos4.
    uint32_t first;
    if (fold) {
        uint32_t mine = 0;
        for (uint32_t i = threadIdx.x; i < bx; i += 256) mine += tile_tasks[i];
        uint32_t before;
        (void)block_scan_excl_u32(mine, L.s_w, &before);
        first = before;
        __syncthreads();
        if (threadIdx.x == 0 && lead_wg) {
            tile_first[bx] = first;
            if (bx == gx - 1) {
                const uint32_t all = first + tile_tasks[bx];
                ctr->ntasks = all;
                ctr->need_tasks = all;
                if (all > task_cap) atomicOr(&ctr->abort_flag, 2u);
            }
        }
    } else {
        first = tile_tasks[bx];
        if (threadIdx.x == 0 && lead_wg) tile_first[bx] = first;
    }
    CutRow cr;
    uint32_t tot, local = 0;
    {
        const uint32_t b = bx * CUT_TILE + threadIdx.x * CUT_ITEMS;
#pragma unroll
        for (int j = 0; j < CUT_ITEMS; ++j) {
            cr.t[j] = b + j < n ? row_t[b + j] : 0u;
            cr.kind[j] = cr.t[j] == 0 ? 0u : (row_cls[b + j] == CLS_BIG ? 2u : 1u);
            local += cr.t[j];
        }
    }
    uint32_t idx = block_scan_excl_u32(local, L.s_w, &tot) + first;
    // (the tile's BIG rows numbered in row order: the same numbers in every workgroup of the tile)
    uint32_t mybig = 0;
#pragma unroll
    for (int j = 0; j < CUT_ITEMS; ++j) mybig += (cr.kind[j] == 2 && cr.t[j]) ? 1u : 0u;
    uint32_t nb;
    __syncthreads();
    uint32_t kbn = block_scan_excl_u32(mybig, L.s_w, &nb);
    if (ctr->abort_flag & ~2u) return;   // a workspace overflowed upstream: nothing below may be trusted (every write is bounded by task_cap)
    const uint32_t base = bx * CUT_TILE + threadIdx.x * CUT_ITEMS;
    uint32_t kb[CUT_ITEMS], idxb[CUT_ITEMS];
#pragma unroll
    for (int j = 0; j < CUT_ITEMS; ++j) {
        kb[j] = 0xFFFFFFFFu;
        idxb[j] = 0;
        if (cr.kind[j] == 1 && idx < task_cap) {
            if (lead_wg) {
            TaskDesc d;      // everything the task needs to start its loads: rows, entries, products, first A entry
            d.kind = TASK_BATCH;
            d.row = base + j;
            d.np = row_binfo[base + j];
            d.first = 0;
            d.src = aptr[r0 + base + j];
            d.col_lo = d.col_hi = 0;
            d.cut = 0;
            d.ri = d.m = 0;
            tasks[idx] = d;
            status[(size_t)idx * ST_STRIDE] = 0ull;
            }
        } else if (cr.kind[j] == 2 && cr.t[j]) {
            kb[j] = kbn++;
            idxb[j] = idx;
        }
        idx += cr.t[j];
    }
    __syncthreads();   // every thread is done with the cut arrays: they now hold the tile's BIG rows
    uint32_t *b_first = L.pc, *b_tb = L.pw, *b_pre = L.nxt;   // first task | first descriptor in tmp | range tasks -> their prefix
#pragma unroll
    for (int j = 0; j < CUT_ITEMS; ++j)
        if (kb[j] != 0xFFFFFFFFu) {
            b_first[kb[j]] = idxb[j];
            b_tb[kb[j]] = row_tmp[base + j];
            b_pre[kb[j]] = cr.t[j];
        }
    __syncthreads();
    // the range descriptors of the tile's BIG rows, copied by the whole workgroup: descriptor q of the concatenation belongs
    // to the BIG row k with pre[k] <= q < pre[k + 1]
    uint32_t M = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += 256) {
        const uint32_t k = b0 + threadIdx.x;
        uint32_t tot2;
        const uint32_t ex = block_scan_excl_u32(k < nb ? b_pre[k] : 0u, L.s_w, &tot2);
        __syncthreads();
        if (k < nb) b_pre[k] = M + ex;
        M += tot2;
        __syncthreads();
    }
    for (uint32_t q = by * 256u + threadIdx.x; q < M; q += 256u * gy) {
        uint32_t lo = 0, hi = nb - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (b_pre[mid] <= q) lo = mid;
            else hi = mid - 1;
        }
        const uint32_t off = q - b_pre[lo];
        bool leg = false;
        if (b_first[lo] + off < task_cap) {
            const TaskDesc d = tmp[b_tb[lo] + off];
            tasks[b_first[lo] + off] = d;
            status[(size_t)(b_first[lo] + off) * ST_STRIDE] = 0ull;
            leg = !task_is_batch(d);
        }
        // the numbers of the tasks that take the older range path (spilled multi-pass / heavy ranges, rows with many entries): one
        // device atomic per wave
        const unsigned long long lm = __ballot(leg);
        if (lm) {
            const int lead = __ffsll((long long)lm) - 1, ln = (int)(threadIdx.x & 63);
            uint32_t lb = 0;
            if (ln == lead) lb = atomicAdd(&ctr->n_legacy, (uint32_t)__popcll(lm));
            lb = (uint32_t)__shfl((int)lb, lead);
            if (leg) legacy[lb + (uint32_t)__popcll(lm & ((1ull << ln) - 1ull))] = b_first[lo] + off;
        }
    }
}

__global__ __launch_bounds__(256) void k_cut3(const uint8_t *__restrict__ row_cls, const uint32_t *__restrict__ row_t,
                                              const uint32_t *__restrict__ row_binfo, const uint64_t *__restrict__ aptr, uint64_t r0,
                                              const uint32_t *__restrict__ row_tmp, uint32_t n,
                                              uint32_t *__restrict__ tile_tasks, const TaskDesc *__restrict__ tmp,
                                              TaskDesc *__restrict__ tasks, uint32_t task_cap, uint32_t fold,
                                              uint32_t *__restrict__ tile_first, uint32_t *__restrict__ legacy,
                                              unsigned long long *__restrict__ status, uint32_t scatter_launched, TaskCounters *__restrict__ ctr)
{
    cut3_body(row_cls, row_t, row_binfo, aptr, r0, row_tmp, n, tile_tasks, tmp, tasks, task_cap, fold, tile_first, legacy, status, scatter_launched, ctr,
              blockIdx.x, gridDim.x, blockIdx.y, gridDim.y);
}

// ONE launch behind the plan (round 6).  The three jobs between k_big_plan and the task kernel need nothing of each other -- the scatter
// of the spilled rows fills the scratch slices, the cut table narrows the direct rows' entries, k_cut3 writes the task list (it reads the
// plan's range descriptors only) -- and the task kernel needs all three.  Until round 5 they were three kernels on three streams: two
// fork / join event pairs (~30 us of the 82 us between the plan's end and the task kernel's start on the web input, whose longest branch
// alone takes 40) and 5 000 mostly idle workgroups of three grids turning over on the same CUs.  Now: one grid whose workgroups take
// scatter runs (by ticket), cut-table items or tiles of the cut by their number -- scatter first: its workgroups walk the longest.
struct AfterPlanArgs {
    // scatter
    const double *aval, *bval;
    const uint32_t *bidx;
    const uint64_t *eb0;
    const uint32_t *elen, *big_rows, *row_kmin, *row_kmax;
    const BigPart *parts;
    const uint32_t *part_hist;
    const BigSlot *slots;
    uint32_t *scr_col;
    double *scr_val;
    uint32_t *scr_seq;
    const uint32_t *spill_parts;
    uint32_t psh, n_scatter;     // workgroups that scatter (0: the scatter is left out of this run)
    // cut table
    const uint32_t *row_m, *row_tmp;
    const TaskDesc *tmp;
    const uint2 *items;
    uint64_t item_cap;
    uint32_t *cuts;
    uint32_t n_cuts, pad0;       // workgroups that build the cut table (0: none)
    // task list
    const uint8_t *row_cls;
    const uint32_t *row_t, *row_binfo;
    const uint64_t *aptr;
    uint64_t r0;
    uint32_t n, task_cap, fold, scatter_launched, ntiles, cut_sub;
    uint32_t *tile_tasks, *tile_first, *legacy;
    TaskDesc *tasks;
    unsigned long long *status;
    TaskCounters *ctr;
};
constexpr size_t AFTER_PLAN_LDS = BX_WALK_LDS > sizeof(CutLds) ? BX_WALK_LDS : sizeof(CutLds);
// LIGHT: compiled for 64 registers (eight workgroups of 256 threads per CU instead of five; the scatter's walk with two product segments in
// flight per thread instead of four) -- for runs whose scatter has little to do (the engine's guess from the context's previous run: the web
// input's 29 spilled rows; a wrong guess costs time only).  The launch is bound by its workgroups' latencies times the slots the CUs have:
// 5 073 workgroups of the web input in 1 280 slots (85 registers: the walk) against 2 048.
template <bool LIGHT>
__global__ __launch_bounds__(256, LIGHT ? 8 : 4) void k_after_plan(const AfterPlanArgs a)
{
    uint32_t b = blockIdx.x;
    if (b < a.n_scatter) {
        big_scatter_body<LIGHT ? 2 : FLAT_U>(a.aval, a.bidx, a.bval, a.eb0, a.elen, a.big_rows, a.row_kmin, a.row_kmax, a.parts, a.part_hist, a.slots, a.scr_col, a.scr_val,
                         a.scr_seq, a.psh, a.spill_parts, a.ctr, b, a.n_scatter);
        return;
    }
    b -= a.n_scatter;
    if (b < a.n_cuts) {
        big_cuts_body(a.bidx, a.eb0, a.elen, a.big_rows, a.row_m, a.row_tmp, a.slots, a.tmp, a.items, a.item_cap, a.cuts, a.ctr, b, a.n_cuts);
        return;
    }
    b -= a.n_cuts;
    cut3_body(a.row_cls, a.row_t, a.row_binfo, a.aptr, a.r0, a.row_tmp, a.n, a.tile_tasks, a.tmp, a.tasks, a.task_cap, a.fold, a.tile_first, a.legacy,
              a.status, a.scatter_launched, a.ctr, b % a.ntiles, a.ntiles, b / a.ntiles, a.cut_sub);
}


// ---- positions after a COUNT run: exclusive scan of the tasks' counts (left in range_out by the task kernel) ---------------------
// k_pos1: sums per tile of POS_TILE tasks; k_pos2 (one workgroup): exclusive scan of the tile sums, nnz(C); k_pos3: positions of the
// tile's tasks -> range_out (every task) and C.indptr of the first range of a BIG row; k_pos4 (one thread per row, the cut's
// tiles): C.indptr of the rows of the batches -- their offsets inside the batch are there already, the position of the batch is added.
constexpr int POS_TILE = 2048, POS_PER = POS_TILE / 256;

__global__ __launch_bounds__(256) void k_pos1(const uint64_t *__restrict__ range_out, const TaskCounters *__restrict__ ctr,
                                              unsigned long long *__restrict__ tile_sum)
{
    __shared__ unsigned long long wtot[4];
    if (ctr->abort_flag) return;
    const uint32_t nt = ctr->ntasks, tiles = (nt + POS_TILE - 1) / POS_TILE;
    for (uint32_t b = blockIdx.x; b < tiles; b += gridDim.x) {
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < POS_PER; ++i) {
            const uint32_t t = b * POS_TILE + i * 256 + threadIdx.x;
            if (t < nt) v += range_out[t];
        }
        unsigned long long tot;
        (void)group_scan_excl_u64<256>(v, threadIdx.x, wtot, &tot);
        if (threadIdx.x == 0) tile_sum[b] = tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_pos2(unsigned long long *__restrict__ tile_sum, uint64_t *__restrict__ cptr, uint32_t nrows,
                                              TaskCounters *__restrict__ ctr)
{
    __shared__ unsigned long long wtot[4];
    if (ctr->abort_flag) return;
    const uint32_t nt = ctr->ntasks, tiles = (nt + POS_TILE - 1) / POS_TILE;
    unsigned long long carry = 0;
    for (uint32_t b0 = 0; b0 < tiles; b0 += 256) {
        const uint32_t b = b0 + threadIdx.x;
        const unsigned long long v = b < tiles ? tile_sum[b] : 0ull;
        unsigned long long tot;
        const unsigned long long ex = group_scan_excl_u64<256>(v, threadIdx.x, wtot, &tot);
        if (b < tiles) tile_sum[b] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cptr[nrows] = carry;
        ctr->nnz_c = carry;
    }
}

__global__ __launch_bounds__(256) void k_pos3(const TaskDesc *__restrict__ tasks, const unsigned long long *__restrict__ tile_sum,
                                              const TaskCounters *__restrict__ ctr, uint64_t *__restrict__ range_out,
                                              uint64_t *__restrict__ cptr)
{
    __shared__ unsigned long long wtot[4];
    if (ctr->abort_flag) return;
    const uint32_t nt = ctr->ntasks, tiles = (nt + POS_TILE - 1) / POS_TILE;
    const int tid = threadIdx.x;
    for (uint32_t b = blockIdx.x; b < tiles; b += gridDim.x) {
        const uint32_t t0 = b * POS_TILE, cnt = min((uint32_t)POS_TILE, nt - t0);
        unsigned long long v[POS_PER], mine = 0;   // thread `tid` owns tasks t0 + tid * POS_PER + i
#pragma unroll
        for (int i = 0; i < POS_PER; ++i) {
            const uint32_t k = tid * POS_PER + i;
            v[i] = k < cnt ? range_out[t0 + k] : 0ull;
            mine += v[i];
        }
        unsigned long long tot;
        unsigned long long pos = tile_sum[b] + group_scan_excl_u64<256>(mine, tid, wtot, &tot);
#pragma unroll
        for (int i = 0; i < POS_PER; ++i) {
            const uint32_t k = tid * POS_PER + i;
            if (k < cnt) {
                const TaskDesc td = tasks[t0 + k];
                range_out[t0 + k] = pos;
                if (td.kind != TASK_BATCH && (td.first & 1u)) cptr[td.row] = pos;
            }
            pos += v[i];
        }
        __syncthreads();
    }
}

// The batch of row r is task  tile_tasks[tile of r] + (tasks started by the tile's rows up to and including r) - 1  (batches do
// not cross the cut's tiles; row_t and tile_tasks are what k_row_class_cut / k_big_plan / k_cut2 left).
__global__ __launch_bounds__(256) void k_pos4(const uint8_t *__restrict__ row_cls, const uint32_t *__restrict__ row_t,
                                              const uint32_t *__restrict__ tile_tasks, uint32_t n, const uint64_t *__restrict__ range_out,
                                              const TaskCounters *__restrict__ ctr, uint64_t *__restrict__ cptr)
{
    __shared__ uint32_t s_w[4];
    if (ctr->abort_flag) return;
    const uint32_t base = blockIdx.x * CUT_TILE + threadIdx.x * CUT_ITEMS;
    uint32_t t[CUT_ITEMS], local = 0;
#pragma unroll
    for (int j = 0; j < CUT_ITEMS; ++j) {
        t[j] = base + j < n ? row_t[base + j] : 0u;
        local += t[j];
    }
    uint32_t tot;
    uint32_t idx = block_scan_excl_u32(local, s_w, &tot) + tile_tasks[blockIdx.x];
#pragma unroll
    for (int j = 0; j < CUT_ITEMS; ++j) {
        idx += t[j];
        if (base + j < n && idx && row_cls[base + j] != CLS_BIG) cptr[base + j] += range_out[idx - 1];
    }
}

// first output position of tasks t[0 .. n) (t[k] == number of tasks: nnz(C)) after a COUNT run: the chunk boundaries of a
// numeric phase that is run in pieces (spada_dev_spgemm_numeric_plan)
__global__ void k_task_positions(const TaskDesc *__restrict__ tasks, const uint64_t *__restrict__ cptr,
                                 const uint64_t *__restrict__ range_out, const TaskCounters *__restrict__ ctr, uint32_t nrows,
                                 const uint32_t *__restrict__ t, uint32_t n, uint64_t *__restrict__ pos)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t ti = t[k];
    if (ti >= ctr->ntasks) pos[k] = cptr[nrows];
    else pos[k] = tasks[ti].kind == TASK_BATCH ? cptr[tasks[ti].row] : range_out[ti];
}


}  // namespace spada
