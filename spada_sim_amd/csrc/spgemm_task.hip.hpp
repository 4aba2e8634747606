// gfx950 kernels, part 3: the TASK pipeline -- one persistent kernel computes all of C in row order.
//
// What it replaces in the reference (citations into /root/reference/src): Scheduler::next_block / next_window
// (scheduler.rs:296-379, :482-606) become a device-built task list; the PE datapath multiply -> sort -> merge
// (simulator.rs:86-230) and the partial-fiber merging (scheduler.rs:381-480, adder_tree.rs:145-188) become one LDS
// accumulator per task; psum write-back and result assembly (simulator.rs:955-1062) become the chained output offsets.
//
//   table fill           ->  `limit` = products a task hashes at most: 2040 (the registers of a task hold 2048 products; the block table
//                            of the batch tasks -- 3072 slots -- is then two thirds full at worst)
//   rows of C            ->  classes by products P_i: EMPTY | COPY (one A entry: C_i = a * B_k) | SMALL (P <= 512) | SOLO
//                            (P <= limit), both packed into batches with their neighbours | BIG (larger)
//   BIG rows             ->  k_big_parts / k_big_hist / k_big_plan: histogram of the row's products over <= 1024 column buckets
//                            (the row cut into parts of ~8192 products, one workgroup each), buckets packed greedily into column
//                            RANGES of <= limit products; every range becomes a task of its own.  A DIRECT range task finds its
//                            products in B itself (B rows are ascending: the positions of the range's bounds in every selected B row
//                            come from the cut table k_big_cuts has left, or from two binary searches inside the task); rows for
//                            which that costs more than a spill (thousands of entries, hundreds of ranges: R-MAT hubs) have their
//                            products scattered into HBM scratch range by range (k_big_scatter, the "spill" of rows whose
//                            accumulator does not fit LDS)
//   task list            ->  k_row_class_cut, (k_cut2), k_cut3: consecutive non-BIG rows are cut into batches of <= limit products; tasks are numbered in
//                            output order (row, then column range)
//   k_task               ->  persistent workgroups of 512 threads take tasks by ticket.  A task expands its products into registers,
//                            accumulates them in the LDS block table of spgemm_batch.hip.hpp (3072 slots keyed by 32-column blocks,
//                            monotone in (row, column): the table is its own sort), counts the distinct outputs, obtains the
//                            position of its slice of C from the per-task status words of the chain (a scanner workgroup turns the
//                            tasks' counts into prefixes) and stores its outputs in (row, column) order.  The range tasks that do
//                            not fit those stages (multi-pass / heavy spilled ranges, rows with more than 512 entries) keep the
//                            older path: flat walk, 2048-slot table keyed by column, bucket-ranked emission (range_task_body; in
//                            the modes without a chain in a kernel of its own, k_task_range)
//                            MODE COUNT   : symbolic phase of the two-phase ABI -- counts only, no chain: the tasks leave their
//                                           counts and k_pos1-4 scan them into C.indptr and the positions of the range tasks
//                            MODE NUMERIC : numeric phase after COUNT -- C.indptr known, no chain
//                            MODE FUSED   : one pass, C written into an upper-bound buffer, C.indptr produced by the chain
// Every workgroup of k_task uses the same 40 KB of LDS (4 per CU = eight waves per SIMD): nothing needs a CU of its own.
#pragma once
#include "spgemm_common.hip.hpp"

// (C is written once and not read again by the pipeline: its stores are non-temporal, which keeps A, B and the descriptors in the caches:
// web -3 %, cop20k_A -1.3 %, R-MAT 16 -1 % per step)
// wave priority from a task's start until its count is published (one-pass mode): what the tasks behind it wait for wins the
// arbitration against emissions and stores; measured -1.4 % (web), -1.1 % (R-MAT 16), 0 elsewhere
constexpr int TASK_PRIO = 3;
#ifndef SPADA_TASK_DBG
#define SPADA_TASK_DBG 0   /* 1 (scripts/build_dbg.sh): phase cycle counters of k_task, printed to stderr */
#endif
#ifndef SPADA_WA_PROBE
#define SPADA_WA_PROBE 0   /* measurement builds, WRONG RESULTS (scripts/dev/write_amp.sh: where the one-pass kernel's extra HBM writes come from):
                              1 the tasks of the older range path are not run | 2 no chain: no status word is stored or read, task t stores at t * 1500 */
#endif

namespace spada {

constexpr uint8_t CLS_EMPTY = 0, CLS_COPY = 1, CLS_SMALL = 2, CLS_SOLO = 3, CLS_BIG = 4;
constexpr int N_CLS = 5, CLS_SLOTS = 64;
constexpr int TK_BLOCK = 256, TK_EPT = 2, TK_LOG_T = 11, TK_T = 1 << TK_LOG_T, TK_RMAX = 128;
// The task kernel itself runs workgroups of TKW = 512 threads (eight waves), four per CU, compiled for 64 VGPRs: eight waves per
// SIMD (round 3: four, 66 % of the wave cycles were waits).  Measured: the utilisation stayed where it was (VALU busy 50 %, waits
// 70 %) -- the kernel is bound by the instructions of its per-wave, per-task overhead, not by latency.  The
// BIG-row kernels, the older range path and the sort-merge variant keep TK_BLOCK = 256.
constexpr int TKW = 512, TKW_EPT = 1;
// Products a task hashes at most (`limit`): TK_LIMIT_HI = 2040 on every input (rounds 1 - 2 sampled the products / outputs ratio of the
// input to choose between 1920 and 2040 for a table keyed by columns; keyed by blocks of columns the table never fills, and the sweep
// is monotone: the fullest tasks are fastest everywhere -- DESIGN.md).  The sort-merge accumulator's limit is TK_SOLO_MAX.
constexpr int TK_NOUT = TK_T;                          // outputs the emission's LDS arrays are sized for
constexpr uint32_t TK_LIMIT_HI = 2040;
constexpr uint32_t TK_SMALL_MAX = 512;   // class boundary SMALL | SOLO (statistics only: both are packed into batches)
constexpr uint32_t TK_SOLO_MAX = 1536;   // the sort-merge accumulator's limit (its network holds 2048 pairs)
constexpr int TK_NQ = 16;                // ticket queues: task t belongs to queue t % TK_NQ, workgroup b serves queue b % TK_NQ (1 / 4 / 8 queues, queues
                                         // spread over the XCDs: within 1 % on web / R-MAT 16, one queue 5 - 10 % slower on cop20k_A: profiles/r04_experiments.txt)
constexpr long long ST_STRIDE = 2;   // words between the chain's status words of consecutive tasks: 16 bytes per task, measured 3 % faster than adjacent words (fewer writers per line)
constexpr int BX_NB = 1024;              // column buckets of the big-row histogram
static_assert(TK_LIMIT_HI + 8 <= (uint32_t)TK_T && TK_NOUT % TK_BLOCK == 0, "the table must keep empty slots");

struct TaskDesc {
    uint32_t kind;      // TASK_BATCH: rows [row, row of the next task) | TASK_RANGE: columns [col_lo, col_hi] of BIG row `row`
    uint32_t row;
    uint32_t np;        // RANGE: products of the slice | BATCH: batch_info(rows, A entries, products)
    uint32_t first;     // RANGE: bit 0 = first range of its row (writes C.indptr[row]); DIRECT: the row's A entries above it
    uint64_t src;       // RANGE: first product of the slice in the scratch arrays | BATCH, DIRECT: first A entry
    uint32_t col_lo, col_hi;
    uint64_t cut;       // DIRECT (rows with at most BT_EMAX entries): the range's two rows of the cut table (k_big_cuts): for entry e of
                        // the row, cuts[cut + e] / cuts[cut + E + e] = first position of the selected B row with a column >= col_lo /
                        // of the next range (the B row's length behind the last range)
    uint32_t ri, m;     // DIRECT: number of the range in its row, ranges of the row
};
static_assert(sizeof(TaskDesc) == 48, "three 16-byte words (load_task)");
constexpr uint32_t TASK_BATCH = 1, TASK_RANGE = 2, TASK_RANGE_DIRECT = 3;   // (DIRECT: the products are taken from B, not from the scratch)

// device counters of one pipeline run (zeroed at its start)
constexpr uint32_t SCATTER_NQ = 16;
struct TaskCounters {   // (a multiple of 8 bytes: k_init clears it in 8-byte words)
    unsigned long long nprod, a_nnz, nprod_big;       // of the row range
    unsigned long long scratch_cursor;                // products handed out in the scratch arrays
    // the cut table and the work items of k_big_cuts are handed out from BX_ARENAS arenas (a hash of the row number: the same arena in every run), a
    // cursor pair per 128-byte line: one hot word takes ~88 atomics per microsecond, and every direct row allocates (k_big_plan 33 ->
    // 99 us on the web input with one cursor)
    unsigned long long cut_arena[16][16];             // [arena][0]: words handed out, [1]: work items
    unsigned long long nnz_c;                         // written by the last task (COUNT / FUSED)
    unsigned long long cls_rows[N_CLS], cls_prod[N_CLS];
    uint32_t n_big, tmp_cursor, ntasks, n_parts;
    uint32_t n_spilled, n_spill_parts;                // BIG rows whose products go through the scratch arrays; their parts (the list k_big_scatter walks)
    uint32_t prod_limit, pad_limit;                   // products a task hashes at most (set by k_entry_stats from its argument)
    uint32_t abort_flag;                              // a workspace was too small: results invalid, sizes below say what is needed
    uint32_t cap_overflow;                            // FUSED: nnz(C) exceeded the caller's capacity (C.indptr is complete)
    uint32_t need_tmp, need_tasks;                    // (abort_flag bits: 1 tmp / scratch, 2 tasks, 4 row too long, 8 BIG rows, 16 parts)
    uint32_t multi_pass_tasks;
    uint32_t scanner_cu;                              // one-pass mode: where the chain's scanner runs (XCC, SE, SH, CU | valid bit)
    uint32_t ticket[2 * TK_NQ * 32];  // TK_NQ ticket counters, one per 128-byte line (a single hot word sustains ~88 atomics / us); the second half: k_task_range
    uint32_t n_legacy;                // tasks of the older range path (their numbers: TaskArgs::legacy)
    uint32_t scanner_leavers;         // one-pass mode: workgroups that left the scanner's CU to it (at most SCANNER_LEAVERS_MAX)
    uint32_t scatter_next[SCATTER_NQ * 32];   // k_big_scatter: runs of parts by ticket (direct rows' parts cost nothing, spilled ones a walk), SCATTER_NQ
                                              // counters on a 128-byte line each: queue q hands out the runs q, q + NQ, q + 2 NQ, ...
#if SPADA_TASK_DBG
    unsigned long long dbgh[3][24];
    unsigned long long dbgs[2048][2][16];  // (per workgroup: no contended atomics in the measurement) tasks that published late (> 30 000 ticks) | all: tasks, products, entries, rows, displaced, outputs, second attempts, dense, range, ticks ticket -> task start, -> gathers arrived, -> publication, tasks in the kernel's last 1000  // per task kind: [0..19] histogram of the cycles from ticket to publish (4096-cycle bins), [20] sum, [21] tasks, [22] max
#endif
    // statistics of k_row_class_cut, spread over CLS_SLOTS lines (workgroup b adds to slot b % CLS_SLOTS; the host sums them): rows per
    // class [0 .. 4], products per class [5 .. 9], A entries [10]; what the first run of a context sizes its workspaces from
    // (task_pipeline, mid-run read): batch tasks [11], an upper estimate of the BIG rows' range tasks [12] and cut-table words [13],
    // products of the BIG rows that are spilled whatever the plan finds [14].  One hot word takes ~90 atomics per microsecond: with the
    // sums in one place the kernel had to run on one workgroup per CU (29 us for a million rows, a third of its memory rate)
    unsigned long long cls_part[64][16];
    unsigned long long dbg[16];  // SPADA_TASK_DBG builds: [0] cycles in the chain, [1] look-back windows, [2] spin retries, [3] cycles
                                 // of the task loop, [4] cycles before the chain (expand + accumulate), [5] cycles after it (emit)
};

// A batch is sized so that ONE chunk of the walk holds its A entries and the registers of the workgroup hold its products
// (spgemm_batch.hip.hpp): at most BT_EMAX entries (hashed, copied and empty ones alike: the entries of consecutive rows are
// contiguous) and at most BT_PMAX products (hashed + copied: two rounds of four per thread).  Rows that cannot be part of such a
// batch -- more than BT_EMAX entries, or one entry that selects more than BT_PMAX products -- are BIG whatever their products.
constexpr uint32_t BT_EMAX = 512, BT_PMAX = 2048;
constexpr int BT_BSHIFT = 5;   // a block = 32 consecutive columns of one row of C (the batch tasks key their table by blocks)
constexpr uint32_t BT_T = 3072;   // slots of the batch tasks' block table (at most `limit` = 2040 blocks: two thirds full at worst)
// a BIG row goes direct only if its range tasks fit the batch stages (one chunk of entries: BX_DIRECT_EMAX = BT_EMAX; 384 / 256 measured:
// profiles/r03_experiments.txt) ...
// ... in calls over at least this many rows: on a small row block (an eighth of the web input) the scatter of the few spilled rows costs
// 20 us and the chain, with a few thousand tasks in all, gains nothing
constexpr uint32_t BX_DIRECT_ROWS = 400000u;
constexpr unsigned long long BX_DIRECT_EMAX = 512;
constexpr int BT_DSHIFT = BT_BSHIFT;   // columns per slot of a dense batch / direct range: 2^BT_DSHIFT
// (batches / ranges whose blocks fit the table slot for slot skip hashing and sorting: DENSE, spgemm_batch.hip.hpp)
static_assert(BT_EMAX == (uint32_t)TKW * TKW_EPT && BT_PMAX == 4u * TKW && TK_LIMIT_HI <= BT_PMAX, "one entry and four products per thread");

// Which tasks run through the batch stages (spgemm_batch.hip.hpp): consecutive non-BIG rows, a column range of a BIG row with at
// most one chunk of entries and at most as many products as the registers hold (a heavy histogram bucket -- many products on few
// columns -- may have more), or a single-pass spilled range that fits the registers and whose blocks fit the
// table slot for slot
// (a single-pass spilled range whose products fit the registers: its slice holds exactly its products.  Round 6: whatever its column
// span -- a range wider than the table's 3072 blocks goes through the HASHED instantiation, as a direct range of that width does; until
// round 5 those took the older range path: 283 tasks of the web input, a kernel of their own in the modes without a chain)
__device__ inline bool task_spill_batch(const TaskDesc &td) { return td.kind == TASK_RANGE && !(td.first & 2u) && td.np <= BT_PMAX; }
__device__ inline bool task_spill_dense(const TaskDesc &td)
{
    return task_spill_batch(td) && (td.col_hi >> BT_DSHIFT) - (td.col_lo >> BT_DSHIFT) < BT_T;   // (slots of 32 columns)
}
__device__ inline bool task_is_batch(const TaskDesc &td)
{
    return td.kind == TASK_BATCH || (td.kind == TASK_RANGE_DIRECT && (td.first >> 1) <= BT_EMAX && td.np <= BT_PMAX) || task_spill_batch(td);
}

// what a task needs to know about a row, in one 16-byte load (written by k_row_class_cut)
struct __attribute__((aligned(16))) RowRec {
    uint32_t kmin, kmax;   // first / last column that can occur in the row of C
    uint32_t nprod;        // products (saturated at 2^32 - 1)
    uint32_t cls;
};

__device__ inline uint8_t row_class(uint64_t P, uint32_t L, uint32_t rmax, uint32_t lim)
{
    if (P == 0) return CLS_EMPTY;
    if (L > BT_EMAX) return CLS_BIG;
    if (L == 1) return P <= BT_PMAX ? CLS_COPY : CLS_BIG;
    if (P <= TK_SMALL_MAX && rmax > 1) return CLS_SMALL;
    if (P <= lim) return CLS_SOLO;
    return CLS_BIG;
}
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
__global__ __launch_bounds__(256) void k_preset_rows(unsigned long long *__restrict__ row_P, uint32_t *__restrict__ row_kmin,
                                                     uint32_t *__restrict__ row_kmax, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        row_P[i] = 0ull;
        row_kmin[i] = 0xFFFFFFFFu;
        row_kmax[i] = 0u;
    }
}

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

__global__ __launch_bounds__(256) void k_big_parts(const uint64_t *__restrict__ aptr, const uint32_t *__restrict__ elen, uint64_t r0,
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
        for (int rr = wave; rr < BP_ROWS; rr += 4) {
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

__global__ __launch_bounds__(TK_BLOCK) void k_big_hist(const uint32_t *__restrict__ bidx, const uint64_t *__restrict__ eb0,
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
__global__ __launch_bounds__(TK_BLOCK) void k_big_plan(const uint64_t *__restrict__ aptr, uint64_t r0, uint32_t nrows_call, uint32_t allow_direct,
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
        const uint32_t row = big_rows[slot];
        const uint32_t kmin = row_kmin[row], kmax = row_kmax[row], wshift = big_wshift(kmin, kmax);
        const uint32_t pb = slots[slot].part_begin, pc = slots[slot].part_count;
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
#pragma unroll
        for (int k = 0; k < BPT; ++k) cnt[tid * BPT + k] = pre[tid * BPT + k + 1] - pre[tid * BPT + k];
        __syncthreads();
        // range starts: a heavy bucket is a range of its own; the light buckets between heavy ones are packed greedily, as many
        // as fit one task's table (<= TK_SOLO_MAX products: a range is closed when the next bucket does not fit, so two consecutive
        // ranges hold more than TK_SOLO_MAX products together).  Fuller ranges = fewer range tasks,
        // fewer searches of the direct tasks, fewer hops of the chain.  Greedy packing is sequential, so it is done with jump
        // pointers: nxt[b] = where the range that starts at b ends (capacity by binary search over the prefix sums, or the next
        // forced start -- a heavy bucket or the bucket after one), all b in parallel; then one thread follows the pointers.
        {
            uint32_t forced[BPT], fex[BPT];
#pragma unroll
            for (int k = 0; k < BPT; ++k) {
                const int bk = tid * BPT + k;
                forced[k] = (bk == 0 || cnt[bk] > lim || cnt[bk - (bk > 0)] > lim) ? 1u : 0u;
            }
            __syncthreads();   // (cnt is read above and reused for the pointers below)
#pragma unroll
            for (int k = 0; k < BPT; ++k) aux[tid * BPT + k] = forced[k];
            __syncthreads();
            block_exclusive_scan4_dpp(aux, hdr + 4);
#pragma unroll
            for (int k = 0; k < BPT; ++k) {
                fex[k] = aux[tid * BPT + k];
                if (forced[k]) rfirst[fex[k]] = tid * BPT + k;   // positions of the forced starts, ascending
            }
            if (tid == TK_BLOCK - 1) rfirst[fex[BPT - 1] + forced[BPT - 1]] = NB;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < BPT; ++k) {
                const uint32_t bk = tid * BPT + k;
                const uint32_t nf = rfirst[fex[k] + forced[k]];   // next forced start behind bk
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
                const uint32_t cap_end = max(lo - 1, bk + 1);
                cnt[bk] = min(cap_end, nf);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < BPT; ++k) aux[tid * BPT + k] = 0u;
            __syncthreads();
            if (tid == 0)
                for (uint32_t bk = 0; bk < (uint32_t)NB; bk = cnt[bk]) aux[bk] = 1u;
            __syncthreads();
        }
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
        if (tid == TK_BLOCK - 1) {
            const uint32_t m = aux[NB - 1] + wgt[BPT - 1];
            const unsigned long long P = pre[NB];
            const unsigned long long a0 = aptr[r0 + row], E = aptr[r0 + row + 1] - a0;
            hdr[48] = (uint32_t)min(E, 0x7FFFFFFFull);
            hdr[49] = (uint32_t)a0;
            hdr[50] = (uint32_t)(a0 >> 32);
            const uint32_t avg_len = (uint32_t)min(P / max(E, 1ull), 0xFFFFFFFFull);
            const unsigned long long steps = 1ull + (avg_len ? 31u - (uint32_t)__clz((int)avg_len) : 0u);   // of one binary search
            const bool direct = allow_direct && hdr[47] == 0 && (unsigned long long)m * E * steps <= (unsigned long long)BX_DIRECT_FACTOR * P &&
                                E * steps <= BX_DIRECT_MAX_SEARCH && (E <= BX_DIRECT_EMAX || nrows_call < BX_DIRECT_ROWS);
            const uint32_t tb = row_tmp[row];   // big_max_ranges(P) >= m records, allocated by k_big_parts
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
        if (hdr[46] == 0u)   // (spilled; the list has room for every part record: both are sized by the parts' capacity)
            for (uint32_t i = tid; i < nreal; i += TK_BLOCK) spill_parts[hdr[57] + i] = pb + i;
        const uint32_t tb = hdr[42];
        const uint64_t sb = ((uint64_t)hdr[44] << 32) | hdr[43], cb = ((uint64_t)hdr[52] << 32) | hdr[51];
        const bool ok = hdr[45] != 0, direct = hdr[46] != 0, has_cuts = hdr[53] != 0;
        const uint32_t m_row = row_m[row];
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
struct CutLds {
    uint32_t pc[CUT_TILE + 1], pw[CUT_TILE + 1], pe[CUT_TILE + 1];   // prefix sums: products to hash, products to copy, A entries
    uint32_t ps[CUT_TILE + 1];                                       // ... blocks of columns between the first and last column of the hashed rows
    uint32_t nxt[CUT_TILE];
    uint8_t mark[CUT_TILE];
    uint32_t s_w[4];
};

// exclusive suffix minimum across the workgroup (the minimum of v over the threads behind this one; none: 0xFFFFFFFF)
__device__ inline uint32_t block_suffix_min_excl_u32(uint32_t v, uint32_t *s_w /*[4]*/)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_down(inc, o);
        if (lane + o < 64) inc = min(inc, t);
    }
    __syncthreads();
    if (lane == 0) s_w[w] = inc;
    __syncthreads();
    uint32_t ex = __shfl_down(inc, 1);
    if (lane == 63) ex = 0xFFFFFFFFu;
    for (int k = w + 1; k < 4; ++k) ex = min(ex, s_w[k]);
    return ex;
}

// batch descriptor word (TaskDesc::np of a TASK_BATCH): rows | A entries << 8 | products (hashed + copied) << 18 | DENSE << 31: the
// column spans of its hashed rows, in blocks, fit the table one slot per block
constexpr uint32_t BINFO_DENSE = 1u << 31;
__host__ __device__ inline uint32_t batch_info(uint32_t R, uint32_t E, uint32_t P) { return R | (E << 8) | (P << 18); }
static_assert(TK_RMAX <= 255 && BT_EMAX <= 1023 && BT_PMAX <= 4095, "batch_info fields");

// tasks started by every row of the tile; returns the exclusive prefix of this thread's first row and the tile total.
// A batch is a maximal run of non-BIG rows (greedy, in row order) with at most `lim` products to hash, at most BT_PMAX products in
// all (hashed + copied: the task keeps them in registers), at most BT_EMAX A entries (one chunk of the walk) and at most `rmax`
// rows.  binfo[j] (batch starts only) = batch_info(rows, entries, products) of the batch that starts at the thread's row j.
__device__ inline uint32_t cut_tile(const uint32_t *__restrict__ row_cl, const uint32_t *__restrict__ row_nprod,
                                    const RowRec *__restrict__ row_rec,
                                    const uint32_t *__restrict__ row_m, uint32_t n, uint32_t rmax, uint32_t lim, CutLds &L,
                                    CutRow &cr, uint32_t *tile_total, uint32_t (&binfo)[CUT_ITEMS])
{
    const uint32_t tile_base = blockIdx.x * CUT_TILE, base = tile_base + threadIdx.x * CUT_ITEMS;
    const uint32_t cnt = min((uint32_t)CUT_TILE, n - tile_base);
    uint8_t cls[CUT_ITEMS];
    bool fat[CUT_ITEMS];   // EMPTY row with more entries than a chunk holds: a batch of its own that has nothing to do
    uint32_t c[CUT_ITEMS], w[CUT_ITEMS], e[CUT_ITEMS], sp[CUT_ITEMS], sc = 0, sw = 0, se = 0, ss = 0;
#pragma unroll
    for (int j = 0; j < CUT_ITEMS; ++j) {
        const uint32_t i = base + j;
        const uint32_t cl = i < n ? row_cl[i] : (uint32_t)CLS_EMPTY;
        RowRec rr{0u, 0u, 0u, (uint32_t)CLS_EMPTY};
        if (i < n) rr = row_rec[i];   // (issued with the other loads of the row, not behind its class)
        cls[j] = (uint8_t)(cl & 7u);
        const uint32_t len = cl >> 3;
        const uint32_t P = i < n ? row_nprod[i] : 0u;
        fat[j] = cls[j] == CLS_EMPTY && len > BT_EMAX;
        c[j] = (cls[j] == CLS_BIG || fat[j]) ? lim + 1 : ((cls[j] == CLS_SMALL || cls[j] == CLS_SOLO) ? P : 0u);
        w[j] = cls[j] == CLS_COPY ? P : 0u;
        e[j] = (cls[j] == CLS_BIG || fat[j]) ? 0u : len;
        sp[j] = 0;
        if ((cls[j] == CLS_SMALL || cls[j] == CLS_SOLO))   // blocks a table addressed by column would need for the row
            sp[j] = min((rr.kmax >> BT_DSHIFT) - (rr.kmin >> BT_DSHIFT) + 1u, 2u * BT_T);
        L.mark[threadIdx.x * CUT_ITEMS + j] = 0;
        sc += c[j];
        sw += w[j];
        se += e[j];
        ss += sp[j];
    }
    uint32_t tot;
    uint32_t ec = block_scan_excl_u32(sc, L.s_w, &tot);
    __syncthreads();
    uint32_t ew = block_scan_excl_u32(sw, L.s_w, &tot);
    __syncthreads();
    uint32_t ee = block_scan_excl_u32(se, L.s_w, &tot);
    __syncthreads();
    uint32_t es = block_scan_excl_u32(ss, L.s_w, &tot);
#pragma unroll
    for (int j = 0; j < CUT_ITEMS; ++j) {
        L.pc[threadIdx.x * CUT_ITEMS + j] = ec;
        L.pw[threadIdx.x * CUT_ITEMS + j] = ew;
        L.pe[threadIdx.x * CUT_ITEMS + j] = ee;
        L.ps[threadIdx.x * CUT_ITEMS + j] = es;
        ec += c[j];
        ew += w[j];
        ee += e[j];
        es += sp[j];
    }
    if (threadIdx.x == 255) {
        L.pc[CUT_TILE] = ec;
        L.pw[CUT_TILE] = ew;
        L.pe[CUT_TILE] = ee;
        L.ps[CUT_TILE] = es;
    }
    __syncthreads();
    // nxt[i]: largest j <= cnt with pc[j] - pc[i] <= lim, (pc + pw)[j] - (pc + pw)[i] <= BT_PMAX, pe[j] - pe[i] <= BT_EMAX, j - i <= rmax
    // (j = i + 1 is always feasible: a row that is not BIG fits a batch by its class)
#pragma unroll
    for (int j = 0; j < CUT_ITEMS; ++j) {
        const uint32_t li = threadIdx.x * CUT_ITEMS + j;
        uint32_t nx = li + 1;
        if (li < cnt && cls[j] != CLS_BIG) {
            const uint32_t limc = L.pc[li] + lim, limp = L.pc[li] + L.pw[li] + BT_PMAX, lime = L.pe[li] + BT_EMAX;
            uint32_t lo = li + 1, hi = min(cnt, li + rmax);   // invariant: lo is feasible
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if (L.pc[mid] <= limc && L.pc[mid] + L.pw[mid] <= limp && L.pe[mid] <= lime) lo = mid;
                else hi = mid - 1;
            }
            nx = lo;
        }
        // (BIG rows and the end of the tile stop a walk: they point nowhere)
        L.nxt[li] = (li < cnt && cls[j] != CLS_BIG && nx < cnt) ? nx : CUT_END;
        // batch starts, to begin with: BIG rows (tasks of their own), and the first row of every run of non-BIG rows -- the tile's
        // first row, or the row after a BIG row
        if (li < cnt) L.mark[li] = (cls[j] == CLS_BIG || li == 0 || L.pc[li] - L.pc[li - 1] > lim) ? 1 : 0;
    }
    __syncthreads();
    // ... then every row that a walk along nxt reaches from such a start.  Walked by pointer doubling (round k marks what lies
    // 2^k hops behind a marked row, then squares the pointers): log2(tile) rounds for all runs at once, where one thread per run
    // following the pointers took up to a tile's worth of dependent LDS reads (the cut of cop20k_A: 71 -> 30 us)
    for (uint32_t span = 1; span < cnt; span <<= 1) {
        uint32_t j1[CUT_ITEMS], j2[CUT_ITEMS];
#pragma unroll
        for (int j = 0; j < CUT_ITEMS; ++j) {
            const uint32_t li = threadIdx.x * CUT_ITEMS + j;
            j1[j] = li < cnt ? L.nxt[li] : CUT_END;
            j2[j] = j1[j] != CUT_END ? L.nxt[j1[j]] : CUT_END;
            if (j1[j] != CUT_END && L.mark[li]) L.mark[j1[j]] = 1;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CUT_ITEMS; ++j) {
            const uint32_t li = threadIdx.x * CUT_ITEMS + j;
            if (li < cnt) L.nxt[li] = j2[j];
        }
        __syncthreads();
    }
    // where the batch that starts at a row ends: the next start behind it (or the end of the tile)
    uint32_t nm[CUT_ITEMS];
    {
        uint32_t first = CUT_END;
#pragma unroll
        for (int j = CUT_ITEMS - 1; j >= 0; --j) {
            const uint32_t li = threadIdx.x * CUT_ITEMS + j;
            if (li < cnt && L.mark[li]) first = li;
        }
        uint32_t run = block_suffix_min_excl_u32(first, L.s_w);
#pragma unroll
        for (int j = CUT_ITEMS - 1; j >= 0; --j) {
            const uint32_t li = threadIdx.x * CUT_ITEMS + j;
            nm[j] = run == CUT_END ? cnt : run;
            if (li < cnt && L.mark[li]) run = li;
        }
    }
    uint32_t local = 0;
#pragma unroll
    for (int j = 0; j < CUT_ITEMS; ++j) {
        const uint32_t i = base + j, li = threadIdx.x * CUT_ITEMS + j;
        cr.t[j] = 0;
        cr.kind[j] = 0;
        binfo[j] = 0;
        if (i < n && L.mark[li]) {
            cr.kind[j] = cls[j] == CLS_BIG ? 2u : 1u;
            cr.t[j] = cls[j] == CLS_BIG ? row_m[i] : 1u;
            if (cls[j] != CLS_BIG) {
                const uint32_t end = nm[j];
                binfo[j] = fat[j] ? batch_info(1u, 0u, 0u)
                                  : batch_info(end - li, L.pe[end] - L.pe[li], (L.pc[end] - L.pc[li]) + (L.pw[end] - L.pw[li]));
                if (!fat[j] && L.pc[end] > L.pc[li] && L.ps[end] - L.ps[li] <= BT_T) binfo[j] |= BINFO_DENSE;
            }
        }
        local += cr.t[j];
    }
    __syncthreads();
    return block_scan_excl_u32(local, L.s_w, tile_total);
}

// k_row_class_cut: the class of every row (by its products P_i and its length), the list of the BIG rows, the statistics -- and the cut of
// its tile of CUT_TILE rows: tasks started by every row -> row_t (0: none, else 1) and the tile's total.  A BIG row starts no task HERE:
// k_big_plan, which knows its ranges, adds them to row_t and to the tile's total (through round 4 the classes and the cut were two
// kernels with the BIG-row stage between them: a launch, its drain and the re-read of the row words on the critical path of every call,
// for a cut that needs nothing the BIG-row kernels write).  (statistics spread over CLS_SLOTS lines: the host sums them)
// (estimates for the first run's workspaces: a BIG row of P products becomes at most 2 P / limit + 2 ranges -- the plan packs buckets
// greedily, two neighbouring ranges together exceed the limit -- plus the column sub-ranges of heavy buckets on very wide matrices)
__device__ inline unsigned long long est_ranges(unsigned long long P, uint32_t lim) { return 2ull * P / lim + P / BT_PMAX + 2ull; }
__global__ __launch_bounds__(256) void k_row_class_cut(const uint64_t *__restrict__ aptr, uint64_t r0, uint32_t nrows, uint32_t rmax,
                                                       const unsigned long long *__restrict__ row_P, const uint32_t *__restrict__ row_kmin,
                                                       const uint32_t *__restrict__ row_kmax, uint32_t *__restrict__ row_nprod,
                                                       uint8_t *__restrict__ row_cls, uint32_t *__restrict__ row_cl,
                                                       RowRec *__restrict__ row_rec, uint32_t *__restrict__ row_m,
                                                       uint32_t *__restrict__ big_rows, TaskCounters *__restrict__ ctr,
                                                       uint32_t *__restrict__ tile_tasks, uint32_t *__restrict__ row_t,
                                                       uint32_t *__restrict__ row_binfo)
{
    const uint32_t lim = ctr->prod_limit;
    __shared__ unsigned long long s_rows[N_CLS], s_prod[N_CLS], s_tot, s_est[3];
    __shared__ CutLds L;
    if (threadIdx.x < N_CLS) s_rows[threadIdx.x] = s_prod[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_tot = 0;
    if (threadIdx.x < 3) s_est[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned long long c_rows[N_CLS] = {0, 0, 0, 0, 0}, c_prod[N_CLS] = {0, 0, 0, 0, 0}, tot_l = 0;
#pragma unroll
    for (int q = 0; q < CUT_ITEMS; ++q) {
        const uint32_t i = blockIdx.x * CUT_TILE + (uint32_t)q * 256u + threadIdx.x;
        uint8_t cls = CLS_EMPTY;
        if (i < nrows) {
            const unsigned long long P = row_P[i];
            const uint32_t L_ = (uint32_t)(aptr[r0 + i + 1] - aptr[r0 + i]);
            cls = row_class(P, L_, rmax, lim);
            const uint32_t P32 = P > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)P;
            row_nprod[i] = P32;
            row_cls[i] = cls;
            row_cl[i] = (uint32_t)cls | (min(L_, 0x1FFFFFFFu) << 3);
            row_rec[i] = RowRec{row_kmin[i], row_kmax[i], P32, (uint32_t)cls};
            row_m[i] = 0;
#pragma unroll
            for (int k = 0; k < N_CLS; ++k) {
                c_rows[k] += cls == k ? 1ull : 0ull;
                c_prod[k] += cls == k ? P : 0ull;
            }
            tot_l += L_;
            if (cls == CLS_BIG) {   // (few rows: LDS atomics of their own)
                const unsigned long long m_est = est_ranges(P, lim);
                atomicAdd(&s_est[0], m_est);
                if (L_ <= BT_EMAX) atomicAdd(&s_est[1], (m_est + 1ull) * L_);
                else atomicAdd(&s_est[2], P);
            }
        }
        const unsigned long long bm = __ballot(i < nrows && cls == CLS_BIG);
        if (bm) {   // BIG rows: one global atomic per wave
            uint32_t base = 0;
            if (lane == __ffsll((long long)bm) - 1) base = atomicAdd(&ctr->n_big, (uint32_t)__popcll(bm));
            base = __shfl(base, __ffsll((long long)bm) - 1);
            if (i < nrows && cls == CLS_BIG) big_rows[base + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull))] = i;
        }
    }
#pragma unroll
    for (int k = 0; k < N_CLS; ++k) {
        const unsigned long long r = wave_sum_u64(c_rows[k]), p = wave_sum_u64(c_prod[k]);
        if (lane == 0 && r) {
            atomicAdd(&s_rows[k], r);
            atomicAdd(&s_prod[k], p);
        }
    }
    const unsigned long long wl = wave_sum_u64(tot_l);
    if (lane == 0 && wl) atomicAdd(&s_tot, wl);
    __syncthreads();   // (the tile's row words are written: the cut below reads them back)
    unsigned long long *part = ctr->cls_part[blockIdx.x % CLS_SLOTS];
    if (threadIdx.x < N_CLS && s_rows[threadIdx.x]) {
        atomicAdd(&part[threadIdx.x], s_rows[threadIdx.x]);
        atomicAdd(&part[N_CLS + threadIdx.x], s_prod[threadIdx.x]);
    }
    if (threadIdx.x == 0 && s_tot) atomicAdd(&part[2 * N_CLS], s_tot);
    CutRow cr;
    uint32_t tot, binfo[CUT_ITEMS];
    (void)cut_tile(row_cl, row_nprod, row_rec, row_m, nrows, rmax, lim, L, cr, &tot, binfo);
    if (threadIdx.x == 0) {
        tile_tasks[blockIdx.x] = tot;
        if (tot) atomicAdd(&part[11], (unsigned long long)tot);
    }
    if (threadIdx.x < 3 && s_est[threadIdx.x]) atomicAdd(&part[12 + threadIdx.x], s_est[threadIdx.x]);
    const uint32_t base = blockIdx.x * CUT_TILE + threadIdx.x * CUT_ITEMS;
#pragma unroll
    for (int j = 0; j < CUT_ITEMS; ++j)
        if (base + j < nrows) {
            row_t[base + j] = cr.t[j];
            row_binfo[base + j] = binfo[j];
        }
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

// ---- 4. the task kernel ------------------------------------------------------------------------------------------------
constexpr int MODE_COUNT = 0, MODE_NUMERIC = 1, MODE_FUSED = 2;
constexpr unsigned long long ST_AGG = 1ull << 62, ST_INC = 2ull << 62, ST_MASK = 3ull << 62;

struct TaskArgs {
    const uint64_t *aptr;
    const double *aval;
    const uint32_t *bidx;
    const double *bval;
    const uint64_t *eb0;
    const uint32_t *elen;
    uint64_t r0;
    uint32_t nrows;
    uint32_t colbits;
    const uint8_t *row_cls;
    const uint32_t *row_kmin, *row_kmax, *row_nprod;
    const uint32_t *arow;           // row of every A entry (spada_dev_csr::rowid)
    const RowRec *row_rec;          // per row: column bounds, products, class (k_row_class_cut)
    const TaskDesc *tasks;
    const uint32_t *scr_col;
    const double *scr_val;
    const uint32_t *scr_seq;        // sort-merge accumulator only: number of the product inside its row (ascending k)
    const uint32_t *legacy;         // numbers of the tasks that take the older range path (k_cut3; the modes without a chain: k_task_range)
    uint32_t b_off32;               // nnz(B) < 2^29: byte offsets into B's index and value arrays fit 32 bits
    uint32_t scanner;               // one-pass mode: enough workgroups are resident to spare one for the chain's scanner (launch_task)
    const uint32_t *cuts;           // cut table of the direct range tasks (k_big_cuts)
    uint64_t *cptr;                 // nrows + 1: COUNT / FUSED write it, NUMERIC reads it
    uint64_t *range_out;            // per task: first output of a RANGE task (COUNT writes, NUMERIC reads)
    unsigned long long *status;     // per task: chain words, zeroed before the launch
    TaskCounters *ctr;
    uint32_t *c_idx;
    double *c_val;
    uint64_t capacity;              // FUSED: entries the caller's C buffers hold
    uint32_t task_lo, task_hi;      // tasks [task_lo, min(task_hi, all)) are run (NUMERIC in chunks; otherwise 0, 0xFFFFFFFF)
    uint32_t stall_task;            // tests only (SPADA_TEST_STALL_TASK): this task never publishes its count -- the chain stops there (0xFFFFFFFF: none)
    uint32_t pad_stall;
    unsigned long long chain_limit; // one-pass mode: wall-clock ticks a wait on the chain may last before the run gives itself up (flag 128)
};

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

// LDS: 256 B hdr | table: keys u32[T], vals f64[T] (re-used after accumulation as lk u32[NOUT], lv f64[NOUT])
//      | region 2: bcnt u32[NOUT], aliased by the walk scratch (disjoint phases)
//      | rows: s_row RowEmit[RMAX + 1], s_a0 u64[RMAX], s_out u64[RMAX], s_re u32[RMAX + 1], s_cnt u32[RMAX]
__host__ __device__ constexpr size_t task_region2()
{
    return (flat_walk_bytes<TK_BLOCK, TK_EPT, true>() > (size_t)TK_NOUT * 4 ? flat_walk_bytes<TK_BLOCK, TK_EPT, true>() : (size_t)TK_NOUT * 4) + 16;
}
__host__ __device__ constexpr size_t task_lds()
{
    return 256 + ((size_t)12 << TK_LOG_T) + ((task_region2() + 15) & ~(size_t)15) + (size_t)(TK_RMAX + 1) * 16 + (size_t)TK_RMAX * 16 +
           (size_t)(TK_RMAX + 1) * 4 + (size_t)TK_RMAX * 4 + 32;
}

// Position of this task's slice of C: sum of the counts of all tasks before it.  Decoupled look-back (one wave): the task
// publishes its own count (AGG), then walks back over the status words of its predecessors, 64 at a time, adding AGG counts
// until it meets an inclusive prefix (INC); finally it publishes its own inclusive prefix.  Status words are single 8-byte
// agent-scope atomics (flag | value): no ordering between separate words is needed.  Tasks are taken by ticket, so every
// predecessor has been started by a resident workgroup and never waits for a later task: the wait is bounded.
// chain_publish: the task's own count, as soon as it is known (thread 0).  chain_lookback: the exclusive prefix, as late as it is
// needed (all threads) -- the LDS half of the emission sits between the two, so the wait for predecessors that are still
// accumulating is mostly over by the time the look-back starts.
//
// The scanner: the walking is taken away from the tasks.  Workgroup 0 of the kernel takes no tasks; its first
// wave reads the status words in task order, SCAN_WIN windows of 64 per step with all loads in flight together, and turns every
// count it finds (AGG) into the inclusive prefix (INC) in place.  A task then waits for ITS OWN word with one lane.  Why: a status
// word is an agent-scope access that no L2 serves (about a microsecond), and with tasks walking back 64 predecessors per round
// trip the front of known prefixes cannot advance faster than 64 tasks per round trip, while a thousand workgroups re-read the
// windows in front of it -- measured (development build with a look-back that costs nothing): 85 of 818 us on the web input, 150 of
// 654 us on the mesh input were spent waiting there.  The scanner reads every word about once and keeps 256 of them in flight.
// 64-word windows of status words the scanner has in flight per step.  Round 4 (task kernel, ms: web / cop20k_A / cage12 / R-MAT 16):
// 4: 0.772 - 0.787 / 0.580 / 0.226 / 4.63 - 4.68; 8: 0.766 - 0.770 / 0.564 / 0.217 / 4.59 - 4.61; 6: 0.799 / 0.596 / 0.230 / 4.67; 12: 0.791 /
// 0.576 / 0.221 / 4.71; 16: 0.87 / 0.73 / 0.274 / 5.1
constexpr int SCAN_WIN = 8;
constexpr uint32_t SCANNER_LEAVERS_MAX = 3;   // (four workgroups per CU: the scanner and three others)
// (a grid too small to spare a workgroup -- every ticket queue must keep one that takes tasks -- walks back as before)
__device__ inline uint32_t task_queue() { return blockIdx.x % (uint32_t)TK_NQ; }
// RESIDENCY: workgroup 0 takes no tasks, so every ticket queue needs ANOTHER workgroup that is resident while the others wait for
// their positions -- queue 0's is workgroup TK_NQ -- i.e. more than TK_NQ workgroups of the grid must run at the same time.  The
// host decides (launch_task: occupancy x CUs must be at least 4 TK_NQ, TaskArgs::scanner); a device whose CUs are masked or held by
// other kernels below that walks back per task as before.
__device__ inline bool chain_has_scanner(uint32_t host_says) { return host_says != 0u && gridDim.x >= 2u * (uint32_t)TK_NQ; }
// A wait on the chain is BOUNDED (MI355X_MICROARCH.md, correctness boundaries: "bound every spin").  The chain is live as long as every
// ticket queue has a resident workgroup (k_task); should that ever fail -- CUs taken away under the kernel, a fault in a predecessor -- a
// waiter that has spun for `limit` wall-clock ticks (seconds: nothing legitimate waits that long, a task waits for tasks that started
// before it) raises flag 128, every waiter looks at the flag now and then and gives up, the kernel drains and the call returns an error
// instead of holding the GPU for ever.
constexpr uint32_t ABORT_CHAIN = 128u;
__device__ inline bool chain_gave_up(TaskCounters *ctr, unsigned long long t0, unsigned long long limit)
{
    if (__hip_atomic_load(&ctr->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ABORT_CHAIN) return true;
    if (limit && wall_clock64() - t0 > limit) {
        atomicOr(&ctr->abort_flag, ABORT_CHAIN);
        return true;
    }
    return false;
}
__device__ inline void chain_publish(unsigned long long *status, uint32_t t, unsigned long long count, uint32_t scanner)
{
    if (threadIdx.x == 0)
        __hip_atomic_store(&status[(size_t)t * ST_STRIDE], ((t == 0 && !chain_has_scanner(scanner)) ? ST_INC : ST_AGG) | count, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// the scanner (first wave of workgroup 0, one-pass mode): tasks [t_lo, t_end) in order
__device__ inline void chain_scanner(unsigned long long *status, uint32_t t_lo, uint32_t t_end, TaskCounters *ctr, unsigned long long limit)
{
    if (threadIdx.x >= 64) return;
    const uint32_t lane = threadIdx.x;
    unsigned long long run = 0;   // counts of the tasks before `next`
    uint32_t next = t_lo;
    uint32_t idle = 0;
    unsigned long long t_idle = 0;
    while (next < t_end) {
        unsigned long long sv[SCAN_WIN];
#pragma unroll
        for (int j = 0; j < SCAN_WIN; ++j) {
            const uint32_t idx = next + (uint32_t)j * 64u + lane;
            sv[j] = 0ull;
            if (idx < t_end && idx >= next) sv[j] = __hip_atomic_load(&status[(size_t)idx * ST_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t adv = 0;
        bool open = true;
#pragma unroll
        for (int j = 0; j < SCAN_WIN; ++j) {
            if (!open) continue;   // (uniform)
            const uint32_t idx = next + (uint32_t)j * 64u + lane;
            const unsigned long long ready = __ballot((sv[j] & ST_MASK) == ST_AGG);
            const uint32_t lead = ready == ~0ull ? 64u : (uint32_t)__ffsll((long long)~ready) - 1u;   // leading lanes with a count
            // (a task's count fits 32 bits -- it has at most 2^32 - 1 products -- and so does the sum of 64 of them as long as they
            // are the counts of table-sized tasks; a window whose counts are larger takes the 64-bit scan)
            const unsigned long long v = lane < lead ? (sv[j] & ~ST_MASK) : 0ull;
            unsigned long long inc;
            if (__ballot(v >> 26) == 0ull) {   // (uniform)
                inc = (unsigned long long)wave_scan_incl_u32((uint32_t)v);
            } else {
                inc = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned long long u = __shfl_up(inc, o);
                    if ((int)lane >= o) inc += u;
                }
            }
            if (lane < lead) __hip_atomic_store(&status[(size_t)idx * ST_STRIDE], ST_INC | (run + inc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            run += __shfl(inc, 63);
            adv += lead;
            open = lead == 64u;
        }
        next += adv;
        if (!adv) {
            __builtin_amdgcn_s_sleep(2);   // (a pause that grows to 2 / 8 / 32 steps, or none at all: within the run-to-run noise)
            if (idle == 0) t_idle = wall_clock64();
            if ((++idle & 255u) == 0u && chain_gave_up(ctr, t_idle, limit)) return;   // (uniform: one wave, the same loads in every lane)
        } else {
            idle = 0;
        }
    }
}

__device__ inline unsigned long long chain_lookback(unsigned long long *status, uint32_t t, unsigned long long count, uint32_t *hdr,
                                                    TaskCounters *ctr, uint32_t scanner, unsigned long long limit)
{
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned long long dbg_win = 0, dbg_spin = 0;
    if (chain_has_scanner(scanner)) {
    // wait for the scanner to turn this task's own count into the inclusive prefix (one lane, one word)
    if (tid == 0) {
        unsigned long long s;
        uint32_t pause = 0, spins = 0;
        unsigned long long t_wait = 0;
        for (;;) {
            s = __hip_atomic_load(&status[(size_t)t * ST_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((s & ST_MASK) == ST_INC) break;
            __builtin_amdgcn_s_sleep(2);
            if (pause < LB_PAUSE_MAX) ++pause;
            for (uint32_t z = 0; z < pause; ++z) __builtin_amdgcn_s_sleep(4);
            if (spins == 0) t_wait = wall_clock64();
            if ((++spins & 255u) == 0u && chain_gave_up(ctr, t_wait, limit)) {   // (the run is given up: any position inside the buffers will do)
                s = ST_INC | count;
                hdr[51] = 1u;   // (the workgroup leaves the task loop: k_task)
                break;
            }
        }
        const unsigned long long excl = (s & ~ST_MASK) - count;
        hdr[48] = (uint32_t)excl;
        hdr[49] = (uint32_t)(excl >> 32);
    }
    __syncthreads();
    {
        const unsigned long long base_ = ((unsigned long long)hdr[49] << 32) | hdr[48];
        __syncthreads();
        return base_;
    }
    }
    if (tid < 64) {
        if (t == 0) {
            if (lane == 0) {
                hdr[48] = 0;
                hdr[49] = 0;
            }
        } else {
            unsigned long long excl = 0;
            long long pos = (long long)t - 1;   // nearest predecessor not yet accounted for
            for (;;) {
                const long long idx = pos - lane;
                unsigned long long s = ST_INC;   // before task 0: inclusive prefix 0
                if (idx >= 0) s = __hip_atomic_load(&status[idx * ST_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long m_inc = __ballot((s & ST_MASK) == ST_INC), m_empty = __ballot((s & ST_MASK) == 0);
                const int first_inc = m_inc ? __ffsll((long long)m_inc) - 1 : 64;
                const unsigned long long relevant = first_inc >= 63 ? ~0ull : ((2ull << first_inc) - 1ull);
                ++dbg_win;
                if (m_empty & relevant) {
                    // wait for the nearest missing predecessor (it started last, it tends to publish last) with ONE lane and a
                    // growing pause: a thousand workgroups re-reading whole windows would slow down the very tasks they wait for
                    const int who = __ffsll((long long)(m_empty & relevant)) - 1;
                    unsigned long long spins = 0;
                    bool gave_up = false;
                    if (lane == who) {
                        uint32_t pause = 0;
                        unsigned long long t_wait = 0;
                        while ((__hip_atomic_load(&status[idx * ST_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ST_MASK) == 0) {
                            if (spins == 0) t_wait = wall_clock64();
                            ++spins;
                            __builtin_amdgcn_s_sleep(8);
                            if (pause < LB_PAUSE_MAX) ++pause;
                            for (uint32_t z = 0; z < pause; ++z) __builtin_amdgcn_s_sleep(16);
                            if ((spins & 255ull) == 0ull && chain_gave_up(ctr, t_wait, limit)) {
                                gave_up = true;
                                break;
                            }
                        }
                    }
                    dbg_spin += __shfl(spins, who);
                    if (__ballot(gave_up)) {   // (the run is given up: any position will do)
                        if (lane == 0) hdr[51] = 1u;
                        break;
                    }
                    continue;
                }
                unsigned long long v = lane <= first_inc ? (s & ~ST_MASK) : 0ull;
                v = wave_sum_u64(v);
                excl += v;
                if (first_inc < 64) break;
                pos -= 64;
            }
            if (lane == 0) {
                __hip_atomic_store(&status[(size_t)t * ST_STRIDE], ST_INC | (excl + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hdr[48] = (uint32_t)excl;
                hdr[49] = (uint32_t)(excl >> 32);
                if (SPADA_TASK_DBG) {
                    atomicAdd(&ctr->dbg[1], dbg_win);
                    atomicAdd(&ctr->dbg[2], dbg_spin);
                }
            }
        }
    }
    __syncthreads();
    const unsigned long long base = ((unsigned long long)hdr[49] << 32) | hdr[48];
    __syncthreads();
    return base;
}

// The counting mode needs no chain: nothing is stored at the positions, so a task leaves its count in range_out[t] (and the
// offsets of its rows inside the batch in C.indptr) and k_pos1/2/3 turn the counts into positions afterwards.  With the chain,
// one task that takes long to count -- a multi-pass range of an R-MAT hub -- holds up every task behind it while they occupy
// the workgroup slots: measured 7x on the hub chunks of R-MAT 22 (277 ms against 38 ms for the numeric phase over the same tasks).
template <int MODE, class G>
__device__ inline void task_publish(const G &g, uint32_t t, unsigned long long count)
{
    if constexpr (MODE == MODE_COUNT) {
        if (threadIdx.x == 0) g.range_out[t] = count;
    } else {
        if (t == g.stall_task) return;   // (tests: a predecessor that never publishes)
        if (SPADA_WA_PROBE & 2) return;
        chain_publish(g.status, t, count, g.scanner);
    }
}
template <int MODE, class G>
__device__ inline unsigned long long task_position(const G &g, uint32_t t, unsigned long long count, uint32_t *hdr)
{
    if constexpr (MODE == MODE_COUNT) return 0ull;
    else if (SPADA_WA_PROBE & 2) {
        __syncthreads();
        return (unsigned long long)t * 1500ull;
    } else return chain_lookback(g.status, t, count, hdr, g.ctr, g.scanner, g.chain_limit);
}

// Ordered emission of the table (all waves): every occupied slot -> bucket = boff[lr] + floor((col - kmin) * n / span), monotone
// inside a row, rows laid out in order, so bucket order IS the order of the task's outputs up to permutations inside a bucket;
// count, scan, scatter into bucket order (the lists re-use the table's LDS), rank inside the bucket -- all in LDS; then
// `resolve()` supplies the position of the task's slice of C (the chain look-back, or 0 when s_out is absolute; NO_STORE = the
// caller's buffers are too small) and the outputs are stored.  Returns what resolve() returned.
// s_row[lr] = {boff, n, kmin, scale}, s_out[lr] = first output of the row relative to that position.  NO = outputs in the table.
constexpr unsigned long long NO_STORE = ~0ull;
template <int BLOCK, bool SINGLE_ROW, int NOUT, class Resolve>
__device__ inline unsigned long long emit_table(unsigned char *smem, uint32_t NO, uint32_t colbits, uint32_t *__restrict__ c_idx,
                                                double *__restrict__ c_val, Resolve &&resolve)
{
    constexpr int T = TK_T, SPT = T / BLOCK, OPT = NOUT / BLOCK;
    uint32_t *hdr = (uint32_t *)smem;
    uint32_t *keys = (uint32_t *)(smem + 256);
    double *vals = (double *)(keys + T);
    uint32_t *lk = keys;
    double *lv = (double *)(smem + 256 + (((size_t)NOUT * 4 + 7) & ~(size_t)7));
    unsigned char *region2 = smem + 256 + ((size_t)12 << TK_LOG_T);
    uint32_t *bcnt = (uint32_t *)region2;
    const RowEmit *s_row = (const RowEmit *)(region2 + ((task_region2() + 15) & ~(size_t)15));
    const uint64_t *s_out = (const uint64_t *)(s_row + TK_RMAX + 1) + TK_RMAX;
    const int tid = threadIdx.x;
    const uint32_t colmask = colbits >= 32 ? 0xFFFFFFFFu : ((1u << colbits) - 1u);
    for (int s = tid; s < NOUT; s += BLOCK) bcnt[s] = 0;
    __syncthreads();
    uint32_t myk[SPT];
    uint16_t myb[SPT];
#pragma unroll
    for (int i = 0; i < SPT; ++i) myk[i] = keys[tid + i * BLOCK];
    RowEmit r0e;
    if constexpr (SINGLE_ROW) r0e = s_row[0];
#pragma unroll
    for (int i = 0; i < SPT; ++i) {
        myb[i] = 0;
        if (myk[i] != EMPTY_KEY) {
            const uint32_t lr = (SINGLE_ROW || colbits >= 32) ? 0u : (myk[i] >> colbits), col = myk[i] & colmask;
            const RowEmit rw = SINGLE_ROW ? r0e : s_row[lr];
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
    group_exclusive_scan<BLOCK, NOUT>(bcnt, tid, hdr + 2);   // ends with a barrier: table fully read by now
#pragma unroll
    for (int i = 0; i < SPT; ++i)
        if (myk[i] != EMPTY_KEY) {
            const uint32_t p = atomicAdd(&bcnt[myb[i]], 1u);   // afterwards bcnt[b] = end of bucket b
            lk[p] = myk[i];
            lv[p] = myv[i];
        }
    __syncthreads();
    uint32_t ecol[OPT];
    uint64_t epos[OPT];
    double evl[OPT];
#pragma unroll
    for (int w = 0; w < OPT; ++w) {
        const uint32_t p = tid + w * BLOCK;
        ecol[w] = 0;
        epos[w] = 0;
        evl[w] = 0.0;
        if (p < NO) {
            const uint32_t k = lk[p];
            const uint32_t lr = (SINGLE_ROW || colbits >= 32) ? 0u : (k >> colbits), col = k & colmask;
            const RowEmit rw = SINGLE_ROW ? r0e : s_row[lr];
            uint32_t bk = (uint32_t)((float)(col - rw.kmin) * rw.scale);
            bk = rw.boff + (bk < rw.n ? bk : rw.n - 1);
            const uint32_t lo = bk ? bcnt[bk - 1] : 0u, hi = bcnt[bk];
            uint32_t r = lo;
            for (uint32_t j = lo; j < hi; ++j) r += (lk[j] < k) ? 1u : 0u;
            epos[w] = s_out[lr] + (r - rw.boff);
            ecol[w] = col;
            evl[w] = lv[p];
        }
    }
    const unsigned long long base = resolve();   // all threads; contains barriers
    if (base != NO_STORE) {
#pragma unroll
        for (int w = 0; w < OPT; ++w)
            if (tid + w * BLOCK < NO) {
                __builtin_nontemporal_store(ecol[w], &c_idx[base + epos[w]]);
                __builtin_nontemporal_store(evl[w], &c_val[base + epos[w]]);
            }
    }
    __syncthreads();
    return base;
}

// Double hashing: the probe sequence of a key advances by an odd step of its own (odd: it visits every slot of the power-of-two
// table).  A wave waits for the longest probe sequence among its 64 lanes, and linear probing's clusters make that tail long.
__device__ inline uint32_t probe_step(uint32_t key) { return ((key * 0x85EBCA6Bu) >> (32 - TK_LOG_T)) | 1u; }

template <int BLOCK>
__device__ inline void table_clear(unsigned char *smem)
{
    uint4 *k4 = (uint4 *)(smem + 256);
    for (int s = threadIdx.x; s < TK_T / 4; s += BLOCK) k4[s] = make_uint4(EMPTY_KEY, EMPTY_KEY, EMPTY_KEY, EMPTY_KEY);
    double2 *v2 = (double2 *)(smem + 256 + (size_t)TK_T * 4);
    for (int s = threadIdx.x; s < TK_T / 2; s += BLOCK) v2[s] = make_double2(0.0, 0.0);
}

// insert `key`, add `v`; returns true when the key was new
template <bool VALUES>
__device__ inline bool table_insert(uint32_t *keys, double *vals, uint32_t key, double v)
{
    uint32_t h = hash_slot<TK_LOG_T>(key);
    const uint32_t step = probe_step(key);
    bool isnew = false;
    for (;;) {
        const uint32_t o = atomicCAS(&keys[h], EMPTY_KEY, key);
        if (o == EMPTY_KEY) { isnew = true; break; }
        if (o == key) break;
        h = (h + step) & (TK_T - 1);
    }
    if constexpr (VALUES) atomicAdd(&vals[h], v);   // simulator.rs:213-218 (order differs)
    return isnew;
}

// RANGE task: accumulate the products of the scratch slice whose column lies in [lo, hi]; returns the number of distinct columns
template <int BLOCK, bool VALUES>
__device__ inline uint32_t range_accumulate(unsigned char *smem, const uint32_t *__restrict__ scr_col,
                                            const double *__restrict__ scr_val, uint64_t src, uint32_t np, uint32_t lo, uint32_t hi,
                                            bool filter)
{
    uint32_t *hdr = (uint32_t *)smem;
    uint32_t *keys = (uint32_t *)(smem + 256);
    double *vals = (double *)(keys + TK_T);
    table_clear<BLOCK>(smem);
    __syncthreads();
    uint32_t mine = 0;
    constexpr int U = 4;
    for (uint32_t p0 = threadIdx.x; p0 < np; p0 += U * BLOCK) {
        uint32_t c[U];
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t p = p0 + u * BLOCK;
            c[u] = p < np ? scr_col[src + p] : EMPTY_KEY;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t p = p0 + u * BLOCK;
            v[u] = 0.0;
            if constexpr (VALUES) v[u] = p < np ? scr_val[src + p] : 0.0;
            if (filter && (c[u] < lo || c[u] > hi)) c[u] = EMPTY_KEY;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (c[u] != EMPTY_KEY) mine += table_insert<VALUES>(keys, vals, c[u], v[u]) ? 1u : 0u;
    }
    const uint32_t n = group_sum<BLOCK>(mine, hdr);
    __syncthreads();
    return n;
}

template <int BLOCK>
__device__ inline uint32_t range_count_products(unsigned char *smem, const uint32_t *__restrict__ scr_col, uint64_t src, uint32_t np,
                                                uint32_t lo, uint32_t hi)
{
    uint32_t mine = 0;
    for (uint32_t p = threadIdx.x; p < np; p += BLOCK) {
        const uint32_t c = scr_col[src + p];
        mine += (c >= lo && c <= hi) ? 1u : 0u;
    }
    const uint32_t n = group_sum<BLOCK>(mine, (uint32_t *)smem);
    __syncthreads();
    return n;
}

// Multi-pass RANGE task: the slice may hold more distinct columns than the table takes (a heavy bucket wider than TK_SOLO_MAX
// columns: only matrices with more than BX_NB * TK_SOLO_MAX = 1.5 M columns can produce one).  Depth-first halving of the column
// range, ascending; a leaf holds <= TK_SOLO_MAX products or <= TK_SOLO_MAX columns, so it fits.  The walk is deterministic: it is
// run once to count (the chain needs the task's total before anything is stored) and once more to emit.
// `stack` = 2 * 40 words of LDS that nothing else uses during a RANGE task.
template <int BLOCK, bool EMIT, int NOUT>
__device__ inline uint32_t range_dfs(unsigned char *smem, uint32_t *stack, RowEmit *s_row, uint64_t *s_out, const TaskDesc &td,
                                     const uint32_t *__restrict__ scr_col, const double *__restrict__ scr_val,
                                     unsigned long long base, uint32_t *__restrict__ c_idx, double *__restrict__ c_val)
{
    const int tid = threadIdx.x;
    uint32_t total = 0, sp = 1;
    if (tid == 0) {
        stack[0] = td.col_lo;
        stack[1] = td.col_hi;
    }
    __syncthreads();
    while (sp) {
        --sp;
        const uint32_t lo = stack[2 * sp], hi = stack[2 * sp + 1];
        __syncthreads();
        const uint32_t cntp = range_count_products<BLOCK>(smem, scr_col, td.src, td.np, lo, hi);
        if (cntp == 0) continue;
        if (cntp > (uint32_t)NOUT && hi - lo >= (uint32_t)NOUT) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (tid == 0) {   // upper half below the lower half: the lower half is popped first
                stack[2 * sp] = mid + 1;
                stack[2 * sp + 1] = hi;
                stack[2 * sp + 2] = lo;
                stack[2 * sp + 3] = mid;
            }
            sp += 2;
            __syncthreads();
            continue;
        }
        const uint32_t nl = range_accumulate<BLOCK, EMIT>(smem, scr_col, scr_val, td.src, td.np, lo, hi, true);
        if constexpr (EMIT) {
            if (nl) {
                if (tid == 0) {
                    s_row[0] = RowEmit{0u, nl, lo, (float)nl / ((float)(hi - lo) + 1.0f)};
                    s_out[0] = base + total;
                }
                __syncthreads();
                (void)emit_table<BLOCK, true, NOUT>(smem, nl, 32u, c_idx, c_val, []() -> unsigned long long { return 0ull; });
            }
        }
        total += nl;
    }
    return total;
}

// DIRECT range task: the products of BIG row `td.row` whose column lies in [col_lo, col_hi], taken from B itself: every entry's
// B row is narrowed to the range by two binary searches (B rows are ascending), then the walk is the usual flat one.
template <int BLOCK, int EPT, bool VALUES, class G>
__device__ inline uint32_t direct_accumulate(unsigned char *smem, unsigned char *region2, uint32_t *s_re, uint64_t *s_a0,
                                             const G &g, const TaskDesc &td)
{
    constexpr int U = FLAT_U;
    uint32_t *hdr = (uint32_t *)smem;
    uint32_t *keys = (uint32_t *)(smem + 256);
    double *vals = (double *)(keys + TK_T);
    const uint64_t a0 = td.src;          // (first entry and entry count travel in the descriptor: no look-up of A's row pointers)
    const uint32_t E = td.first >> 1;
    table_clear<BLOCK>(smem);
    if (threadIdx.x == 0) {
        s_re[0] = 0;
        s_re[1] = E;
        s_a0[0] = a0;
    }
    __syncthreads();
    const uint32_t lo = td.col_lo, hi = td.col_hi;
    const uint32_t *__restrict__ bidx = g.bidx;
    uint32_t mine = 0;
    flat_walk<BLOCK, EPT, 1, VALUES, U>(
        s_re, s_a0, 1u, E, g.eb0, g.elen, g.aval, g.bidx, g.bval, region2, hdr,
        [&](uint32_t(&col)[U], uint32_t(&plr)[U], double(&v)[U], uint32_t(&)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (plr[u] != LR_NONE) mine += table_insert<VALUES>(keys, vals, col[u], v[u]) ? 1u : 0u;
        },
        [&](uint64_t(&b0)[EPT], uint32_t(&len)[EPT]) {
            // l1 = first position with column >= lo, l2 = first position with column > hi; all searches in lock step
            uint32_t l1[EPT], n1[EPT], l2[EPT], n2[EPT];
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                l1[i] = l2[i] = 0;
                n1[i] = n2[i] = len[i];
            }
            for (;;) {
                uint32_t any = 0;
#pragma unroll
                for (int i = 0; i < EPT; ++i) any |= n1[i] | n2[i];
                if (!any) break;
                uint32_t c1[EPT], c2[EPT];
#pragma unroll
                for (int i = 0; i < EPT; ++i) {
                    c1[i] = n1[i] ? bidx[b0[i] + l1[i] + (n1[i] >> 1)] : 0u;
                    c2[i] = n2[i] ? bidx[b0[i] + l2[i] + (n2[i] >> 1)] : 0u;
                }
#pragma unroll
                for (int i = 0; i < EPT; ++i) {
                    if (n1[i]) {
                        const uint32_t h = n1[i] >> 1;
                        if (c1[i] < lo) {
                            l1[i] += h + 1;
                            n1[i] -= h + 1;
                        } else {
                            n1[i] = h;
                        }
                    }
                    if (n2[i]) {
                        const uint32_t h = n2[i] >> 1;
                        if (c2[i] <= hi) {
                            l2[i] += h + 1;
                            n2[i] -= h + 1;
                        } else {
                            n2[i] = h;
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                b0[i] += l1[i];
                len[i] = l2[i] - l1[i];
            }
        });
    const uint32_t n = group_sum<BLOCK>(mine, hdr);
    __syncthreads();
    return n;
}

}  // namespace spada
#include "spgemm_batch.hip.hpp"
namespace spada {

// dynamic LDS of k_task: its RANGE tasks use the layout of task_lds(), its BATCH tasks the one of spgemm_batch.hip.hpp
// (SPADA_TASK_DBG builds: 32 more words behind everything for the per-workgroup counters -- phase ticks [0 .. 7], batch tasks [8],
// task shapes [9 .. 13] -- that thread 0 adds to TaskCounters::dbg when the workgroup ends: one hot word takes ~90 atomics per
// microsecond, per-task atomics made the instrumented kernel six times slower)
__host__ __device__ constexpr size_t task_dbg_off() { return ((task_lds() > batch_lds() ? task_lds() : batch_lds()) + 15) & ~(size_t)15; }
__host__ __device__ constexpr size_t task_kernel_lds() { return task_dbg_off() + (SPADA_TASK_DBG ? 128 : 0); }
static_assert(task_kernel_lds() <= 40960, "four workgroups per CU");
static_assert(flat_walk_bytes<TKW, TKW_EPT, true>() == flat_walk_bytes<TK_BLOCK, TK_EPT, true>(), "one walk scratch size for both workgroup shapes");

// ---- RANGE task of the older kind: columns [col_lo, col_hi] of a BIG row -- the products of a spilled row's scratch slice that do not
// fit the batch stages (more products than the registers hold, sub-ranges of a heavy bucket, multi-pass ranges), or a direct range
// of a row with more than BT_EMAX entries: table keyed by column, monotone buckets + in-bucket rank (emit_table).
// NOT inlined into the task kernel: its registers (the flat walk holds four products and their entry records per thread) are
// allocated on their own, and nothing of it is kept live across the batch tasks of the loop.
// Single pass when the slice cannot overflow the table (at most NOUT products or columns; a column sub-range of a heavy
// bucket keeps the products of its own columns), else range_dfs.
// descriptor of task t (uniform).  The task list was written by the kernels before this one: read through the constant address
// space, i.e. with one scalar load into scalar registers
__device__ inline TaskDesc load_task(const TaskDesc *tasks, uint32_t t)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const u32x4 __attribute__((address_space(4))) *cptr4;
    const cptr4 p = (cptr4)(unsigned long long)(tasks + t);
    const u32x4 a = p[0], b = p[1], c = p[2];
    TaskDesc d;
    d.kind = a.x;
    d.row = a.y;
    d.np = a.z;
    d.first = a.w;
    d.src = ((uint64_t)b.y << 32) | b.x;
    d.col_lo = b.z;
    d.col_hi = b.w;
    d.cut = ((uint64_t)c.y << 32) | c.x;
    d.ri = c.z;
    d.m = c.w;
    return d;
}
// (arguments of a function that is not inlined arrive in vector registers: what is uniform is made scalar again)
typedef const TaskArgs __attribute__((address_space(4))) TaskArgsC;
__device__ inline TaskArgsC &uniform_args(const TaskArgs *p)
{
    const unsigned long long v = (unsigned long long)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return *(TaskArgsC *)(((unsigned long long)hi << 32) | lo);
}
template <int BLOCK, int EPT, int MODE, int NOUT, class ARGS>
__device__ __forceinline__ void range_task_body(const ARGS &g, const TaskDesc &td, uint32_t t, uint32_t ntasks, unsigned char *smem)
{
    constexpr int RMAX = TK_RMAX;
    constexpr bool VALUES = MODE != MODE_COUNT;
    uint32_t *hdr = (uint32_t *)smem;
    unsigned char *region2 = smem + 256 + ((size_t)12 << TK_LOG_T);
    unsigned char *rows = region2 + ((task_region2() + 15) & ~(size_t)15);
    RowEmit *s_row = (RowEmit *)rows;
    uint64_t *s_a0 = (uint64_t *)(s_row + RMAX + 1);
    uint64_t *s_out = s_a0 + RMAX;
    uint32_t *s_re = (uint32_t *)(s_out + RMAX);
    const int tid = threadIdx.x;
    const bool direct = td.kind == TASK_RANGE_DIRECT;
    const bool single = direct || td.np <= (uint32_t)NOUT || td.col_hi - td.col_lo < (uint32_t)NOUT;
    uint32_t total;
    if (direct) {
        total = direct_accumulate<BLOCK, EPT, VALUES>(smem, region2, s_re, s_a0, g, td);
    } else if (single) {
        total = range_accumulate<BLOCK, VALUES>(smem, g.scr_col, g.scr_val, td.src, td.np, td.col_lo, td.col_hi, (td.first & 2u) != 0);
    } else {
        if (tid == 0) atomicAdd(&g.ctr->multi_pass_tasks, 1u);
        total = range_dfs<BLOCK, false, NOUT>(smem, s_re, s_row, s_out, td, g.scr_col, g.scr_val, 0ull, nullptr, nullptr);
    }
    if constexpr (MODE != MODE_NUMERIC) task_publish<MODE>(g, t, total);
    if constexpr (MODE == MODE_FUSED) __builtin_amdgcn_s_setprio(0);
    auto resolve = [&]() -> unsigned long long {
        if constexpr (MODE == MODE_NUMERIC) {
            return g.range_out[t];
        } else {
            const unsigned long long b0 = task_position<MODE>(g, t, total, hdr);
            if (MODE != MODE_COUNT && tid == 0) {
                if (td.first & 1u) g.cptr[td.row] = b0;
                g.range_out[t] = b0;
                if (t == ntasks - 1) {
                    g.cptr[g.nrows] = b0 + total;
                    g.ctr->nnz_c = b0 + total;
                }
            }
            if constexpr (MODE == MODE_FUSED) {
                if (b0 + total > g.capacity) {
                    if (tid == 0) atomicOr(&g.ctr->cap_overflow, 1u);
                    return NO_STORE;
                }
            }
            return b0;
        }
    };
    if (MODE != MODE_COUNT && single && total) {
        if (tid == 0) {
            s_row[0] = RowEmit{0u, total, td.col_lo, (float)total / ((float)(td.col_hi - td.col_lo) + 1.0f)};
            s_out[0] = 0;
        }
        __syncthreads();
        (void)emit_table<BLOCK, true, NOUT>(smem, total, 32u, g.c_idx, g.c_val, resolve);
    } else {
        const unsigned long long base = resolve();
        if (MODE != MODE_COUNT && !single && base != NO_STORE)
            (void)range_dfs<BLOCK, true, NOUT>(smem, s_re, s_row, s_out, td, g.scr_col, g.scr_val, base, g.c_idx, g.c_val);
    }
}

// ... as a function of its own inside the 512-thread task kernel (one-pass mode: every task of the chain runs there)
template <int MODE, int NOUT>
__device__ __attribute__((noinline)) void range_task(const TaskArgs *gp_, uint32_t t_, uint32_t ntasks_)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TaskArgsC &g = uniform_args(gp_);
    const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_), ntasks = (uint32_t)__builtin_amdgcn_readfirstlane((int)ntasks_);
    const TaskDesc td = load_task(g.tasks, t);
    range_task_body<TKW, TKW_EPT, MODE, NOUT>(g, td, t, ntasks, smem);
}

// which instantiation of the batch stages a descriptor takes.  DENSE: the blocks between the first and the last column of every
// hashed row fit the table slot for slot
__device__ inline int task_variant(const TaskDesc &d)
{
    if (task_spill_batch(d)) return task_spill_dense(d) ? 2 : 3;
    const bool dense = (d.kind == TASK_BATCH ? (d.np & BINFO_DENSE) != 0
                                                                : (d.col_hi >> BT_DSHIFT) - (d.col_lo >> BT_DSHIFT) < BT_T);
    return dense ? 1 : 0;
}
// The prologue of task t, if it is a batch task (else nothing).  A function of its own: it is called where the task before waits
// for its position -- next to nothing is live there, and its saves and restores lie in the wait -- and ONE copy of the three
// instantiations serves every call site.
template <int MODE>
__device__ __attribute__((noinline)) BatchHead task_prologue(const TaskArgs *gp_, uint32_t t_)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TaskArgsC &g = uniform_args(gp_);
    const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_);
    const TaskDesc td = load_task(g.tasks, t);
    if (!task_is_batch(td)) return BatchHead{0u, 0u, 0u, 0u};
    const int v = task_variant(td);
    if (v == 3) return batch_prologue<MODE, false, true>(g, td, t, smem);
    if (v == 2) return batch_prologue<MODE, true, true>(g, td, t, smem);
    if (v == 1) return batch_prologue<MODE, true>(g, td, t, smem);
    return batch_prologue<MODE, false>(g, td, t, smem);
}

// The arguments live in device memory (TaskArgs written by k_task_args just before) and are read through the constant address
// space where they are used: passed by value, the ~30 words would be loaded in the kernel's first block and stay live -- i.e.
// spilled to VGPR lanes -- through the whole task loop (84 v_writelane, 308 v_readlane in the first build of this kernel).
__global__ void k_task_args(const TaskArgs g, TaskArgs *__restrict__ dst)
{
    if (threadIdx.x == 0) *dst = g;
}
// the head of a numeric call in ONE launch: the task kernel's arguments, the ticket counters put back to zero, and C.indptr copied into the
// caller's buffer (until round 5: a copy command, a fill command and k_task_args -- ~30 us of commands and gaps in front of the task kernel)
__global__ __launch_bounds__(256) void k_numeric_head(const TaskArgs g, TaskArgs *__restrict__ dst, uint32_t *__restrict__ ticket, uint32_t n_ticket,
                                                      const uint64_t *__restrict__ cptr, uint64_t *__restrict__ cptr_out, uint64_t n_ptr)
{
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) *dst = g;
        for (uint32_t i = threadIdx.x; i < n_ticket; i += 256) ticket[i] = 0u;
    }
    if (cptr_out)
        for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_ptr; i += (uint64_t)gridDim.x * 256) cptr_out[i] = cptr[i];
}
constexpr int TASK_WAVES = 8;   // waves per SIMD the task kernel is compiled for (HIP: second argument of __launch_bounds__): 8 = 64 VGPRs
template <int MODE, int NOUT>
__global__ __launch_bounds__(TKW, TASK_WAVES) void k_task(const TaskArgs *__restrict__ gp_)
{
    TaskArgsC &g = *(TaskArgsC *)gp_;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    const int tid = threadIdx.x;
    const uint32_t ntasks = g.ctr->ntasks, task_end = min(ntasks, g.task_hi);
    if (g.ctr->abort_flag) return;
    if (chain_has_scanner(g.scanner) && MODE == MODE_FUSED) {
        // The chain's scanner: workgroup 0 takes no tasks.  It gets its CU for itself: the other workgroups that
        // land there leave at once (three of 1024).  Every link of the chain -- a task's count to the scanner, the prefix back -- is a
        // hand-off whose price sits in the memory queue of the CU that reads: 1.1 us on a CU with nothing else in flight, 3 - 5 us on
        // one that streams (MI355X_MICROARCH.md, handoff-1to1), and every task of the kernel waits on both links
        const uint32_t me = ((uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFF00u)            /* HW_ID: CU, SH, SE */
                            | ((uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 16) | 1u;      /* XCC_ID */
        if (blockIdx.x == 0) {
            if (tid == 0) __hip_atomic_store(&g.ctr->scanner_cu, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_setprio(3);
            if (SPADA_WA_PROBE & 2) return;
            chain_scanner(g.status, g.task_lo, task_end, g.ctr, g.chain_limit);
            return;
        }
        {
            if (tid == 0) {
                uint32_t sc;
                while (!((sc = __hip_atomic_load(&g.ctr->scanner_cu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 1u)) __builtin_amdgcn_s_sleep(8);
                // At most SCANNER_LEAVERS_MAX workgroups leave -- the ones that share the CU with the scanner when the kernel starts -- and
                // never one of the first 2 TK_NQ: the chain is live as long as every ticket queue keeps a resident worker, and a slot that
                // every later arrival left again would let a grid that is held back (CUs masked, or busy with another kernel) drain
                // through the scanner's CU without ever serving its queues
                hdr[51] = sc == me && blockIdx.x >= 2u * (uint32_t)TK_NQ && atomicAdd(&g.ctr->scanner_leavers, 1u) < SCANNER_LEAVERS_MAX ? 1u : 0u;
            }
            __syncthreads();
            if (hdr[51]) return;
        }
    }

    // Tasks are taken by ticket, in (almost) chain order: queue q hands out tasks q, q + NQ, q + 2 NQ, ...  The smallest task that
    // is not finished is either running -- it waits for finished tasks only -- or the next one of its queue, whose workgroups
    // all hold smaller, hence finished, tasks and are free to take it: no cycle of waiting workgroups can form as long as
    // every queue has a resident workgroup, which a grid of at least TK_NQ workgroups dispatched in order guarantees.
    uint32_t *my_ticket = &g.ctr->ticket[(task_queue()) * 32];
    // (STATIC assignment instead of tickets -- worker w of G takes tasks w, w + G, ...: round 4 measured it at -1 % (web) / -6.5 % (R-MAT 16)
    // and did not adopt it, because the chain is then live only while EVERY workgroup of the grid is resident.  Round 6 built it with
    // what it was supposed to make possible -- the next task is known, so its whole prologue runs under the wait for the position / under
    // the stores -- and a bounded wait as the safety net: one-pass web 0.766 against 0.769 ms, cop20k_A 0.554 / 0.562, R-MAT 16 4.16 / 3.50;
    // the modes without a chain LOSE 12 % (web count 0.466 against 0.411, numeric 0.738 against 0.658: tickets balance the tail, a fixed
    // share does not).  Hiding two of a task's three dependent round trips buys nothing: profiles/r06_experiments.txt section 2.)
    if (tid == 0) {
        hdr[50] = g.task_lo + atomicAdd(my_ticket, 1u) * TK_NQ + task_queue();
        hdr[51] = 0u;
    }
    if (SPADA_TASK_DBG && tid < 32) ((uint32_t *)(smem + task_dbg_off()))[tid] = 0u;
    __syncthreads();
    // (t is uniform: the descriptor is a scalar load, what is derived from it lives in scalar registers)
    uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[50]);
    uint32_t *dbg_ph = (uint32_t *)(smem + task_dbg_off());   // SPADA_TASK_DBG builds
    if (SPADA_TASK_DBG && tid == 0) dbg_ph[16] = (uint32_t)__builtin_amdgcn_s_memtime();
    TaskDesc td{};
    BatchHead hd{0u, 0u, 0u, 0u};
    bool bt = false;
    auto uniform_head = [](const BatchHead &h) {   // (what a function returns arrives in vector registers)
        return BatchHead{(uint32_t)__builtin_amdgcn_readfirstlane((int)h.P), (uint32_t)__builtin_amdgcn_readfirstlane((int)h.nent),
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)h.NBK), (uint32_t)__builtin_amdgcn_readfirstlane((int)h.ncopy)};
    };
    if (t < task_end) {
        td = load_task(g.tasks, t);
        bt = task_is_batch(td);
        hd = uniform_head(task_prologue<MODE>(gp_, t));
    }
    // Nothing of a task may be computed once before this loop and held in registers across every task -- the kernel is compiled for
    // 64 of them, such values are spilled, and a scratch reload waits for ALL vector loads in flight (one counter), which cut the
    // one-round-trip prologue of the first build of this kernel into pieces: the batch task makes its thread number and its
    // constants opaque per task, and the prologue and the older range path are functions of their own.
    while (t < task_end) {
        // The NEXT ticket is taken -- and the next task's prologue is run -- where this task has nothing left to do but wait for its
        // position and store (`next`, called by the task).  Tickets taken earlier than that cost more than they hide: a task that
        // sits unstarted in the chain holds up every task behind it (measured in rounds 2 and 3: +6 % on the web surrogate, +30 % on
        // R-MAT 16 for a ticket taken across the stores); taken HERE the workgroup would otherwise idle.
        if constexpr (MODE == MODE_FUSED) __builtin_amdgcn_s_setprio(TASK_PRIO);
        uint32_t t2 = 0xFFFFFFFFu;
        TaskDesc td2{};
        BatchHead hd2{0u, 0u, 0u, 0u};
        bool bt2 = false;
        // The modes without a chain take the next ticket EARLY -- where the task's products have arrived and it works in LDS for
        // thousands of ticks (`early`, called by the batch task): the atomic's round trip, a third of the dependent chain ticket ->
        // descriptor -> row / entry loads in front of every task, lies under the insertion.  (With the chain a ticket taken before
        // the task is done holds up every task behind it: comment above.  Issued any earlier, the atomic would stand in front of the
        // task's own loads in the in-order return queue.)
        uint32_t tk_early = 0;
        bool have_early = false;
        auto early = [&]() {
            if constexpr (MODE != MODE_FUSED) {
                if (threadIdx.x == 0) tk_early = atomicAdd(my_ticket, 1u);
                have_early = true;
            }
        };
        auto next = [&]() {
            __syncthreads();   // (the ticket word of the task before has been read by everyone; this task's outputs are complete in LDS)
            if (threadIdx.x == 0) hdr[50] = g.task_lo + (have_early ? tk_early : atomicAdd(my_ticket, 1u)) * TK_NQ + task_queue();
            __syncthreads();
            t2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[50]);
            if (SPADA_TASK_DBG && threadIdx.x == 0) dbg_ph[16] = (uint32_t)__builtin_amdgcn_s_memtime();
            if (t2 < task_end) {
                td2 = load_task(g.tasks, t2);
                bt2 = task_is_batch(td2);
                hd2 = uniform_head(task_prologue<MODE>(gp_, t2));
            }
        };
        if (bt) {
            const int v = task_variant(td);
            if (v == 3) batch_main<MODE, false, true>(g, td, t, ntasks, smem, dbg_ph, hd, next, early);
            else if (v == 2) batch_main<MODE, true, true>(g, td, t, ntasks, smem, dbg_ph, hd, next, early);
            else if (v == 1) batch_main<MODE, true>(g, td, t, ntasks, smem, dbg_ph, hd, next, early);
            else batch_main<MODE, false>(g, td, t, ntasks, smem, dbg_ph, hd, next, early);
        } else {
            if constexpr (MODE == MODE_FUSED) {
                if (!(SPADA_WA_PROBE & 1)) range_task<MODE, NOUT>(gp_, t, ntasks);
                else task_publish<MODE>(g, t, 0ull);   // (the chain goes on without the task's outputs)
            }   // (the modes without a chain: k_task_range takes these tasks)
            next();
        }
        t = t2;
        td = td2;
        hd = hd2;
        bt = bt2;
        if (MODE == MODE_FUSED && hdr[51]) break;   // (uniform; a wait on the chain ran into its limit -- chain_gave_up: the run is lost, the workgroup leaves)
    }
    if (SPADA_TASK_DBG && tid == 0) {
        const uint32_t *d = (const uint32_t *)(smem + task_dbg_off());
        for (int k = 0; k < 8; ++k) atomicAdd(&g.ctr->dbg[8 + k], (unsigned long long)d[k]);
        atomicAdd(&g.ctr->dbg[6], (unsigned long long)d[8]);
        if (d[8]) atomicAdd(&g.ctr->dbg[7], 1ull);   // workgroups that took a task: the resident ones
        for (int k = 0; k < 6; ++k) atomicAdd(&g.ctr->dbg[k], (unsigned long long)d[9 + k]);   // shapes (spgemm_batch.hip.hpp)
#if SPADA_TASK_DBG
        for (int k = 0; k < 2; ++k) {   // ticket -> publication per kind (batch, range): sum / 16, tasks, maximum
            atomicAdd(&g.ctr->dbgh[k][0], (unsigned long long)d[17 + 3 * k]);
            atomicAdd(&g.ctr->dbgh[k][1], (unsigned long long)d[18 + 3 * k]);
            atomicMax(&g.ctr->dbgh[k][2], (unsigned long long)d[19 + 3 * k]);
        }
        for (int k = 0; k < 9; ++k) atomicAdd(&g.ctr->dbgh[0][3 + k], (unsigned long long)d[23 + k]);   // second attempts, latency histogram
#endif
    }
}

// The modes WITHOUT a chain (COUNT, NUMERIC) run the tasks of the older range path in a kernel of their own, in the shape that path was
// written for: workgroups of 256 threads with 128 registers (inside the 512-thread kernel it is compiled for 64 and spills: the
// chunks of R-MAT 22 whose hub rows are spilled took 2.5 x as long).  k_cut3 leaves the numbers of those tasks in `legacy`; tasks are
// independent in these modes, so the two kernels simply follow each other on the stream.
template <int MODE, int NOUT>
__global__ __launch_bounds__(TK_BLOCK, 4) void k_task_range(const TaskArgs *__restrict__ gp_)
{
    TaskArgsC &g = *(TaskArgsC *)gp_;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    const int tid = threadIdx.x;
    if (g.ctr->abort_flag) return;
    const uint32_t nl = g.ctr->n_legacy, ntasks = g.ctr->ntasks, task_end = min(ntasks, g.task_hi);
    uint32_t *my_ticket = &g.ctr->ticket[(TK_NQ + task_queue()) * 32];
    for (;;) {
        __syncthreads();
        if (tid == 0) hdr[50] = atomicAdd(my_ticket, 1u) * TK_NQ + task_queue();
        __syncthreads();
        const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)hdr[50]);
        if (k >= nl) break;
        const uint32_t t = g.legacy[k];
        if (t < g.task_lo || t >= task_end) continue;   // (a numeric phase in pieces)
        const TaskDesc td = load_task(g.tasks, t);
        range_task_body<TK_BLOCK, TK_EPT, MODE, NOUT>(g, td, t, ntasks, smem);
    }
}

// ---- 5. the task kernel with the SORT-MERGE accumulator (SPADA_ACC_SORT_MERGE) --------------------------------------------
// The closest GPU analogue of what the reference's PE does to one group: collect the products (simulator.rs:86-111), sort them
// by column (SortingNetwork, simulator.rs:143-171), add runs of equal column left to right (MergeTree, simulator.rs:199-230).
// Same task list, same chain (none in the counting mode), same three modes as k_task -- only the accumulator differs, over exactly the same rows, so the
// two variants of BASELINE.json configs[2] are like for like.  A task writes its products to LDS as (key, value) pairs with
// key = (local row, column, product number) in 64 bits (the product number -- ascending k -- breaks ties, so a run is added in
// the order of the CPU restatement and the values are bit-identical to a sequential sort-merge), sorts them with a bitonic
// network sized to the next power of two of its product count, and the first product of every run adds its run.
// LDS: 256 B hdr | sk u64[SM_N] | sv f64[SM_N] | heads u64[SM_N / 64] | hpre u32[SM_N / 64] | walk scratch | rows (as k_task)
constexpr int SM_N = 2048;
static_assert(SM_N >= (int)TK_SOLO_MAX, "a task holds at most TK_SOLO_MAX products");
__host__ __device__ constexpr size_t task_sm_lds()
{
    return 256 + (size_t)SM_N * 16 + (size_t)(SM_N / 64) * 12 + ((flat_walk_bytes<TK_BLOCK, TK_EPT, true>() + 31) & ~(size_t)15) +
           (size_t)(TK_RMAX + 1) * 16 + (size_t)TK_RMAX * 16 + (size_t)(TK_RMAX + 1) * 4 + (size_t)TK_RMAX * 4 + 32;
}

// bitonic network over the first N (a power of two) pairs, ascending keys
template <bool VALUES>
__device__ inline void sm_sort(unsigned long long *sk, double *sv, uint32_t N)
{
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < N / 2; t += TK_BLOCK) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), x = i | j;   // i has bit j clear
                const unsigned long long a = sk[i], c = sk[x];
                if ((a > c) == ((i & k) == 0)) {
                    sk[i] = c;
                    sk[x] = a;
                    if constexpr (VALUES) {
                        const double va = sv[i];
                        sv[i] = sv[x];
                        sv[x] = va;
                    }
                }
            }
            __syncthreads();
        }
}

// run heads of the sorted keys (a run = equal upper 32 bits): bit masks + prefix counts; returns the number of runs
__device__ inline uint32_t sm_heads(const unsigned long long *sk, uint32_t N, unsigned long long *heads, uint32_t *hpre, uint32_t *hdr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t HW = N / 64;   // 1 .. 32
    for (uint32_t w = wave; w < HW; w += TK_BLOCK / 64) {
        const uint32_t p = w * 64 + lane;
        const unsigned long long cur = sk[p], prev = p ? sk[p - 1] : ~0ull;
        const bool head = cur != ~0ull && (p == 0 || (cur >> 32) != (prev >> 32));
        const unsigned long long m = __ballot(head);
        if (lane == 0) heads[w] = m;
    }
    __syncthreads();
    if (wave == 0) {
        const uint32_t c = (uint32_t)lane < HW ? (uint32_t)__popcll(heads[lane]) : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if ((uint32_t)lane < HW) hpre[lane] = inc - c;
        if (lane == 63) hdr[47] = inc;
    }
    __syncthreads();
    return hdr[47];
}

template <int MODE>
__global__ __launch_bounds__(TK_BLOCK, 3) void k_task_sm(const TaskArgs g)
{
    constexpr int BLOCK = TK_BLOCK, EPT = TK_EPT, RMAX = TK_RMAX, U = FLAT_U;
    constexpr bool VALUES = MODE != MODE_COUNT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    unsigned long long *sk = (unsigned long long *)(smem + 256);
    double *sv = (double *)(sk + SM_N);
    unsigned long long *heads = (unsigned long long *)(sv + SM_N);
    uint32_t *hpre = (uint32_t *)(heads + SM_N / 64);
    unsigned char *region2 = (unsigned char *)(hpre + SM_N / 64);
    unsigned char *rows = region2 + ((flat_walk_bytes<BLOCK, EPT, true>() + 31) & ~(size_t)15);
    RowEmit *s_row = (RowEmit *)rows;
    uint64_t *s_a0 = (uint64_t *)(s_row + RMAX + 1);
    uint64_t *s_out = s_a0 + RMAX;
    uint32_t *s_re = (uint32_t *)(s_out + RMAX);
    uint32_t *s_cnt = s_re + RMAX + 1;
    const int tid = threadIdx.x;
    const uint32_t ntasks = g.ctr->ntasks, task_end = min(ntasks, g.task_hi);
    if (g.ctr->abort_flag) return;
    if (chain_has_scanner(g.scanner) && MODE == MODE_FUSED && blockIdx.x == 0) {   // the chain's scanner (see k_task)
        chain_scanner(g.status, g.task_lo, task_end, g.ctr, g.chain_limit);
        return;
    }
    const uint32_t colmask = g.colbits >= 32 ? 0xFFFFFFFFu : ((1u << g.colbits) - 1u);
    uint32_t *my_ticket = &g.ctr->ticket[(task_queue()) * 32];
    if (tid == 0) hdr[50] = g.task_lo + atomicAdd(my_ticket, 1u) * TK_NQ + task_queue();
    __syncthreads();
    uint32_t t = hdr[50];
    __syncthreads();

    // sorted pairs -> C: the first product of every run adds its run left to right (simulator.rs:209-220) and stores it
    auto emit_runs = [&](uint32_t N, unsigned long long base, bool batch) {
        for (uint32_t p = tid; p < N; p += BLOCK) {
            const unsigned long long hw = heads[p >> 6];
            if (!((hw >> (p & 63)) & 1ull)) continue;
            const uint32_t rank = hpre[p >> 6] + (uint32_t)__popcll(hw & ((1ull << (p & 63)) - 1ull));
            const uint32_t key = (uint32_t)(sk[p] >> 32);
            double acc = sv[p];
            for (uint32_t q = p + 1; q < N && (uint32_t)(sk[q] >> 32) == key; ++q) acc += sv[q];
            const uint32_t lr = (!batch || g.colbits >= 32) ? 0u : (key >> g.colbits);
            const uint64_t pos = base + s_out[lr] + (rank - s_row[lr].boff);
            __builtin_nontemporal_store(batch ? (key & colmask) : key, &g.c_idx[pos]);   // (as in k_task: C is not read again)
            __builtin_nontemporal_store(acc, &g.c_val[pos]);
        }
    };

    while (t < task_end) {
        const TaskDesc td = g.tasks[t];
        if (td.kind == TASK_BATCH) {
            const uint32_t rb = td.row;
            const uint32_t re = t + 1 < ntasks ? g.tasks[t + 1].row : g.nrows;
            const uint32_t R = re - rb;
            uint32_t L = 0, n = 0, rid = 0, clen = 0;
            uint64_t cb0 = 0, c0 = 0;
            double cav = 0.0;
            uint8_t cls = CLS_EMPTY;
            if ((uint32_t)tid < R) {
                rid = rb + tid;
                const uint64_t a0 = g.aptr[g.r0 + rid], a1 = g.aptr[g.r0 + rid + 1];
                cls = g.row_cls[rid];
                s_a0[tid] = a0;
                s_cnt[tid] = 0;
                if (cls == CLS_SMALL || cls == CLS_SOLO) {
                    L = (uint32_t)(a1 - a0);
                } else if (cls == CLS_COPY) {
                    cb0 = g.eb0[a0];
                    clen = g.elen[a0];
                    if constexpr (VALUES) cav = g.aval[a0];
                }
                if constexpr (MODE == MODE_NUMERIC) {
                    c0 = g.cptr[rid];
                    n = (uint32_t)(g.cptr[rid + 1] - c0);
                }
            }
            uint32_t E;
            const uint32_t exl = group_scan_excl<BLOCK>(L, tid, hdr + 2, &E);
            if ((uint32_t)tid < R) s_re[tid] = exl;
            if (tid == 0) s_re[R] = E;
            for (int q = tid; q < SM_N; q += BLOCK) sk[q] = ~0ull;   // padding sorts to the end
            __syncthreads();
            // expand + scale: product number pp of the batch -> sk[pp], sv[pp]
            uint32_t mine = 0;
            if (E)
                flat_walk<BLOCK, EPT, RMAX, VALUES, U>(s_re, s_a0, R, E, g.eb0, g.elen, g.aval, g.bidx, g.bval, region2, hdr,
                                                       [&](uint32_t(&col)[U], uint32_t(&plr)[U], double(&v)[U], uint32_t(&pp)[U]) {
#pragma unroll
                                                           for (int u = 0; u < U; ++u)
                                                               if (plr[u] != LR_NONE) {
                                                                   const uint32_t key = compose_key(plr[u], col[u], g.colbits);
                                                                   sk[pp[u]] = ((unsigned long long)key << 32) | pp[u];
                                                                   if constexpr (VALUES) sv[pp[u]] = v[u];
                                                                   mine = max(mine, pp[u] + 1u);
                                                               }
                                                       });
            const uint32_t np = group_max<BLOCK>(mine, hdr);   // products of the batch (<= TK_SOLO_MAX)
            uint32_t N = 64;
            while (N < np) N <<= 1;
            __syncthreads();
            uint32_t NO = 0;
            if (np) {
                sm_sort<VALUES>(sk, sv, N);
                NO = sm_heads(sk, N, heads, hpre, hdr);
                if constexpr (MODE != MODE_NUMERIC) {   // outputs per row
                    for (uint32_t p = tid; p < N; p += BLOCK)
                        if ((heads[p >> 6] >> (p & 63)) & 1ull) {
                            const uint32_t key = (uint32_t)(sk[p] >> 32);
                            atomicAdd(&s_cnt[g.colbits >= 32 ? 0u : (key >> g.colbits)], 1u);
                        }
                    __syncthreads();
                }
            }
            const bool hashed = cls == CLS_SMALL || cls == CLS_SOLO;
            if constexpr (MODE != MODE_NUMERIC) n = hashed ? s_cnt[tid < RMAX ? tid : 0] : (cls == CLS_COPY ? clen : 0u);
            unsigned long long tot64;
            const unsigned long long ex64 = group_scan_excl_u64<BLOCK>(((unsigned long long)n << 32) | (hashed ? n : 0u), tid,
                                                                       (unsigned long long *)(hdr + 4), &tot64);
            const uint32_t boff = (uint32_t)ex64, ooff = (uint32_t)(ex64 >> 32), total = (uint32_t)(tot64 >> 32);
            __syncthreads();
            unsigned long long base = 0;
            if constexpr (MODE != MODE_NUMERIC) {
                task_publish<MODE>(g, t, total);
                base = task_position<MODE>(g, t, total, hdr);
                if ((uint32_t)tid < R) g.cptr[rid] = base + ooff;
                if (MODE != MODE_COUNT && t == ntasks - 1 && tid == 0) {
                    g.cptr[g.nrows] = base + total;
                    g.ctr->nnz_c = base + total;
                }
                c0 = base + ooff;
            }
            bool store = MODE != MODE_COUNT;
            if constexpr (MODE == MODE_FUSED) {
                if (base + total > g.capacity) {
                    store = false;
                    if (tid == 0) atomicOr(&g.ctr->cap_overflow, 1u);
                }
            }
            if (store) {
                if ((uint32_t)tid < R) {
                    s_row[tid] = RowEmit{boff, n, 0u, 0.f};
                    s_out[tid] = MODE == MODE_NUMERIC ? c0 : (uint64_t)ooff;
                }
                __syncthreads();
                if (NO) emit_runs(N, base, true);
                // COPY rows, as in k_task
                uint32_t *s_cpre = (uint32_t *)region2;
                uint64_t *s_cb0 = (uint64_t *)(region2 + (RMAX + 2) * 4);
                double *s_cav = (double *)(s_cb0 + RMAX);
                uint64_t *s_cc0 = (uint64_t *)(s_cav + RMAX);
                const bool copy = (uint32_t)tid < R && cls == CLS_COPY;
                uint32_t Cp;
                const uint32_t cex = group_scan_excl<BLOCK>(copy ? clen : 0u, tid, hdr + 2, &Cp);
                if (Cp) {
                    if ((uint32_t)tid < R) {
                        s_cpre[tid] = cex;
                        s_cb0[tid] = cb0;
                        s_cav[tid] = cav;
                        s_cc0[tid] = c0;
                    }
                    if (tid == 0) s_cpre[R] = Cp;
                    __syncthreads();
                    for (uint32_t p = tid; p < Cp; p += BLOCK) {
                        uint32_t lo = 0;
#pragma unroll
                        for (int step = RMAX / 2; step >= 1; step >>= 1)
                            if (lo + step < R && s_cpre[lo + step] <= p) lo += step;
                        const uint32_t off = p - s_cpre[lo];
                        __builtin_nontemporal_store(g.bidx[s_cb0[lo] + off], &g.c_idx[s_cc0[lo] + off]);
                        __builtin_nontemporal_store(s_cav[lo] * g.bval[s_cb0[lo] + off], &g.c_val[s_cc0[lo] + off]);
                    }
                }
            }
        } else {
            // RANGE task: the products of the scratch slice with a column in [lo, hi], as pairs key = (column, number of the
            // product in its row); `leaf` sorts, counts and (EMIT) stores one column range that fits the network
            const bool single = td.np <= TK_SOLO_MAX;
            auto leaf = [&](uint32_t lo, uint32_t hi, bool filter, bool do_emit, unsigned long long at) -> uint32_t {
                for (int q = tid; q < SM_N; q += BLOCK) sk[q] = ~0ull;
                if (tid == 0) hdr[46] = 0;
                __syncthreads();
                for (uint32_t p = tid; p < td.np; p += BLOCK) {
                    const uint32_t c = g.scr_col[td.src + p];
                    if (filter && (c < lo || c > hi)) continue;
                    const uint32_t d = filter ? atomicAdd(&hdr[46], 1u) : p;   // any order: the sort restores ascending k
                    sk[d] = ((unsigned long long)c << 32) | g.scr_seq[td.src + p];
                    if constexpr (VALUES) sv[d] = g.scr_val[td.src + p];
                }
                __syncthreads();
                const uint32_t cnt = filter ? hdr[46] : td.np;
                uint32_t N = 64;
                while (N < cnt) N <<= 1;
                __syncthreads();
                sm_sort<VALUES>(sk, sv, N);
                const uint32_t nl = sm_heads(sk, N, heads, hpre, hdr);
                if (do_emit && nl) {
                    if (tid == 0) {
                        s_row[0] = RowEmit{0u, nl, 0u, 0.f};
                        s_out[0] = 0;
                    }
                    __syncthreads();
                    emit_runs(N, at, false);
                }
                __syncthreads();
                return nl;
            };
            // One column with more products than the network holds (a row with thousands of entries whose B rows all contain
            // that column): its products are taken in ascending ranges of their product number -- halved, depth first, until a
            // range fits -- sorted, and added one after the other to an accumulator that is carried from range to range, which
            // is still the left-to-right sum of the whole run.
            auto single_column = [&](uint32_t col, bool do_emit, unsigned long long at) -> uint32_t {
                uint32_t *stack2 = s_cnt;
                double *acc = (double *)(hdr + 52);
                uint32_t sp2 = 1;
                bool first = true;
                if (tid == 0) {
                    stack2[0] = 0u;
                    stack2[1] = 0xFFFFFFFFu;
                }
                __syncthreads();
                while (sp2) {
                    --sp2;
                    const uint32_t slo = stack2[2 * sp2], shi = stack2[2 * sp2 + 1];
                    __syncthreads();
                    uint32_t mine = 0;
                    for (uint32_t p = tid; p < td.np; p += BLOCK) {
                        const uint32_t q = g.scr_seq[td.src + p];
                        mine += (g.scr_col[td.src + p] == col && q >= slo && q <= shi) ? 1u : 0u;
                    }
                    const uint32_t cnt = group_sum<BLOCK>(mine, hdr);
                    __syncthreads();
                    if (cnt == 0) continue;
                    if (cnt > TK_SOLO_MAX) {   // product numbers are distinct: shi > slo here
                        const uint32_t mid = slo + (shi - slo) / 2;
                        if (tid == 0) {
                            stack2[2 * sp2] = mid + 1;
                            stack2[2 * sp2 + 1] = shi;
                            stack2[2 * sp2 + 2] = slo;
                            stack2[2 * sp2 + 3] = mid;
                        }
                        sp2 += 2;
                        __syncthreads();
                        continue;
                    }
                    for (int q = tid; q < SM_N; q += BLOCK) sk[q] = ~0ull;
                    if (tid == 0) hdr[46] = 0;
                    __syncthreads();
                    for (uint32_t p = tid; p < td.np; p += BLOCK) {
                        const uint32_t q = g.scr_seq[td.src + p];
                        if (g.scr_col[td.src + p] != col || q < slo || q > shi) continue;
                        const uint32_t d = atomicAdd(&hdr[46], 1u);
                        sk[d] = ((unsigned long long)q << 32) | d;
                        if constexpr (VALUES) sv[d] = g.scr_val[td.src + p];
                    }
                    __syncthreads();
                    uint32_t N = 64;
                    while (N < cnt) N <<= 1;
                    sm_sort<VALUES>(sk, sv, N);
                    if (VALUES && tid == 0) {
                        double a = first ? sv[0] : *acc;
                        for (uint32_t q = first ? 1u : 0u; q < cnt; ++q) a += sv[q];
                        *acc = a;
                    }
                    first = false;
                    __syncthreads();
                }
                if (do_emit && tid == 0) {
                    g.c_idx[at] = col;
                    g.c_val[at] = *acc;
                }
                __syncthreads();
                return 1u;
            };
            // column ranges with more than TK_SOLO_MAX products are halved, depth first and ascending (as range_dfs of k_task)
            auto dfs = [&](bool do_emit, unsigned long long at) -> uint32_t {
                uint32_t *stack = s_re;
                uint32_t total = 0, sp = 1;
                if (tid == 0) {
                    stack[0] = td.col_lo;
                    stack[1] = td.col_hi;
                }
                __syncthreads();
                while (sp) {
                    --sp;
                    const uint32_t lo = stack[2 * sp], hi = stack[2 * sp + 1];
                    __syncthreads();
                    const uint32_t cntp = range_count_products<BLOCK>(smem, g.scr_col, td.src, td.np, lo, hi);
                    if (cntp == 0) continue;
                    if (cntp > TK_SOLO_MAX && hi > lo) {
                        const uint32_t mid = lo + (hi - lo) / 2;
                        if (tid == 0) {
                            stack[2 * sp] = mid + 1;
                            stack[2 * sp + 1] = hi;
                            stack[2 * sp + 2] = lo;
                            stack[2 * sp + 3] = mid;
                        }
                        sp += 2;
                        __syncthreads();
                        continue;
                    }
                    if (cntp > TK_SOLO_MAX) total += single_column(lo, do_emit, at + total);
                    else total += leaf(lo, hi, true, do_emit, at + total);
                }
                return total;
            };
            uint32_t total;
            if (single) total = leaf(td.col_lo, td.col_hi, false, false, 0ull);
            else {
                if (tid == 0) atomicAdd(&g.ctr->multi_pass_tasks, 1u);
                total = dfs(false, 0ull);
            }
            unsigned long long base;
            if constexpr (MODE != MODE_NUMERIC) {
                task_publish<MODE>(g, t, total);
                base = task_position<MODE>(g, t, total, hdr);
                if (MODE != MODE_COUNT && tid == 0) {
                    if (td.first & 1u) g.cptr[td.row] = base;
                    g.range_out[t] = base;
                    if (t == ntasks - 1) {
                        g.cptr[g.nrows] = base + total;
                        g.ctr->nnz_c = base + total;
                    }
                }
            } else {
                base = g.range_out[t];
            }
            bool store = MODE != MODE_COUNT;
            if constexpr (MODE == MODE_FUSED) {
                if (base + total > g.capacity) {
                    store = false;
                    if (tid == 0) atomicOr(&g.ctr->cap_overflow, 1u);
                }
            }
            if (store && total) {
                if (single) {   // the sorted pairs of the single pass are still in LDS
                    uint32_t N = 64;
                    while (N < td.np) N <<= 1;
                    if (tid == 0) {
                        s_row[0] = RowEmit{0u, total, 0u, 0.f};
                        s_out[0] = 0;
                    }
                    __syncthreads();
                    emit_runs(N, base, false);
                } else {
                    (void)dfs(true, base);
                }
            }
        }
        __syncthreads();
        if (tid == 0) hdr[50] = g.task_lo + atomicAdd(my_ticket, 1u) * TK_NQ + task_queue();
        __syncthreads();
        t = hdr[50];
        __syncthreads();
    }
}

}  // namespace spada
