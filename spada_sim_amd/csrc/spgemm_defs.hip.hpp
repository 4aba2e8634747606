// gfx950 kernels of the task pipeline, part 1 of 5: CONSTANTS AND RECORDS shared by the kernels and the engine -- task shapes and limits, the task
// descriptor, the device counters of a run, the per-row record, the row classes, the modes and the arguments of the task kernel.
// (spgemm_task.hip.hpp is the umbrella: it includes the five parts in order and says what the pipeline replaces in the reference.)
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

#ifndef SPADA_PRE_DBG
#define SPADA_PRE_DBG 0   /* 1: phase clocks of k_row_class_cut, 2: of k_big_plan (thread 0 of every workgroup), summed into TaskCounters::dbg, printed to stderr */
#endif
#if SPADA_PRE_DBG
__device__ inline unsigned long long *pre_ticks()
{
    __shared__ unsigned long long s_pre_tick[16];
    return s_pre_tick;
}
__device__ inline void pre_tick(int k)
{
    if (threadIdx.x == 0) pre_ticks()[k] = __builtin_amdgcn_s_memtime();
}
#else
__device__ inline void pre_tick(int) {}
#endif

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


}  // namespace spada
