// gfx950 kernels, part 4: the BATCH task of the task kernel (included by spgemm_task.hip.hpp) -- consecutive non-BIG rows of C
// computed by one workgroup of 512 threads (eight waves; four such workgroups per CU = eight waves per SIMD) with one round trip
// per stage; the columns of a task are ordered by the TABLE ITSELF.
//
// What it replaces in the reference (citations into /root/reference/src): for the rows of one batch, the window fetch of A
// scalars (scheduler.rs:482-606, storage.rs:279-323), the B-fiber streaming (simulator.rs:892-953), the multiplier
// (simulator.rs:86-111), the sort by column (simulator.rs:143-171), the sum of equal columns (simulator.rs:199-230) and the
// result assembly (simulator.rs:1034-1062: columns ascending and unique, explicit zeros kept).
//
// Shape of a batch (k_row_class_cut, batch_info): R <= 128 consecutive rows, E <= 512 A entries (contiguous: the entries of consecutive
// rows are), P <= 2048 products of which at most `limit` are hashed; the rest belong to COPY rows (one A entry: C_i = a * B_k is
// already ascending and needs no accumulator).  Stages:
//   prologue   the descriptor carries R, E, P and the first A entry: the entry loads (B-row begin / length, A value, row of the
//              entry: ONE entry per thread) and the row records (one 16-byte RowRec per row) are issued together, the table is
//              cleared under them (one-pass mode, where the prologue runs behind the stores of the task before; in the other modes it
//              runs under those stores and the table is cleared at the start of the task); the scans over the rows and over the entries
//              share one barrier
//   expand     four products per thread, 64 consecutive products per wave and step (coalesced gathers from B); the entry of a
//              product comes from the tail bits of the entries (2048 bits) with a v_mbcnt pair.  The products STAY IN REGISTERS
//   accumulate the table is keyed by BLOCK -- (local row, column / 32) -- and holds a 32-bit mask of the columns seen in the block.
//              It is NOT hashed: the table's slots are laid out over the rows in proportion to their products, and inside a row's
//              region the home slot of a block is a MONOTONE function of its column (linear interpolation between the row's first
//              and last possible block; one slot per block when the span fits the region).  Collisions probe upwards (ds_cmpst).
//              So the slots hold the blocks in (row, column) order except inside a cluster of occupied slots, and no separate
//              sorting pass (bucket count, scan, scatter) exists
//   count      new mask bits = outputs of the task: published to the chain (one-pass mode) as soon as the expansion is done
//   order      ONE prefix sum of the masks' popcounts in slot order (six consecutive slots per thread) = first output of every
//              block, then a fix for the blocks that were displaced from their home slot: a displaced block looks at the slots
//              between its home and its place and trades places (first outputs) with the larger keys it finds there
//   scale-add  output of a product = first(block) + popcount(mask below its bit): no search, no comparison.  The product that set
//              a mask bit first stores its value at that rank, the others add theirs (ds_add_f64; simulator.rs:213-218 adds left
//              to right, here the order is arbitrary: 1e-9)
//   emit       once the task's position is known (the chain): the dense arrays are stored as they are (neighbouring lanes,
//              neighbouring addresses); the retained products of COPY rows were put into the same arrays at their place in the slice
//   second     a task whose blocks CLUSTER (R-MAT rows: a third of a row's blocks on a twentieth of its span) would probe
//   attempt    quadratically with a linear home-slot mapping: a lane that is displaced by BT_PROBE_MAX slots gives up, the task
//              counts its products in 256 bins of the rows' spans and starts over with home slots in proportion to the bins'
//              products (still monotone in (row, column))
// DENSE (k_row_class_cut / the range's bounds decide): the blocks between the first and the last column of every hashed row, added up over
// the rows, fit the table one slot per block: no keys at all (meshes, banded matrices, narrow column ranges).
// The DIRECT RANGE tasks of BIG rows (columns [col_lo, col_hi] of one row; the row's entries are narrowed to the range by the cut
// table k_big_cuts has left: B rows are ascending) and the single-pass SPILLED ranges whose blocks fit (products read from the
// scratch slice) run through the same stages: one row, no COPY entries, the range as the row's column bounds.
// LDS (40 832 bytes: four workgroups per CU; regions are reused by the stages):
//   hdr 256 | keys u32[3072] | masks u32[3072] (together later: values f64[2048] + composite keys u32[2048] in output order)
//   | entry records 8 KB (later: first outputs u16[3072]) | tail bits 256 | rows 5 KB (BtRow + six words per row) | bins 1 KB
//   | displaced blocks u32[352]
#pragma once

namespace spada {

#ifndef SPADA_ABLATE
#define SPADA_ABLATE 0   /* development (scripts/dev/ablate.sh): WRONG RESULTS -- stages of the batch task left out, to see what the kernel's time is sensitive
                            to: 1 displaced-block fix | 2 popcount sweep + ranks | 4 scale-add into LDS | 8 stores to C | 16 table atomics | 32 B gathers;
                            bits 8 .. 11: the NUMERIC mode cut short behind stage k (instruction counts per stage: scripts/dev/stage_counts.sh) */
#endif
constexpr int BW = TKW, BT_NWAVE = BW / 64;
constexpr uint32_t BT_LAYOUT = BT_T - 32u;   // slots the rows' regions are laid out over; the rest takes the overflow of the last cluster
constexpr uint32_t BT_H_NONE = 0xFFFFFFFFu, BT_H_COPY = 0xFFFFFFFEu;   // `state` of a lane without a product / of a copied product
// state of a hashed product: slot (12 bits) | creator of the block << 12 | first to set its mask bit << 13
constexpr uint32_t BT_ST_CREATOR = 1u << 12, BT_ST_OWNER = 1u << 13, BT_ST_HASHED_MAX = 0x10000000u;
constexpr size_t BT_OFF_KEYS = 256, BT_OFF_MASK = BT_OFF_KEYS + (size_t)BT_T * 4, BT_OFF_ENT = BT_OFF_MASK + (size_t)BT_T * 4,
                 BT_OFF_TAIL = BT_OFF_ENT + 8192, BT_OFF_ROWS = BT_OFF_TAIL + 256;
struct __attribute__((aligned(16))) BtRow {   // region of a row in the table: first slot, slots, first possible block, slots per block
    uint32_t s, g, bmin;
    float scale;
};
constexpr uint32_t BT_BINS = 256;   // equalised home slots (second attempt of a task whose blocks cluster): bins of the rows' column spans
constexpr uint32_t BT_PROBE_MAX = 24;   // slots a block may be displaced in the first attempt before the task starts over with equalised home slots (16 / 32: within noise; 64 / 128: R-MAT 18 + 40 ... 75 %)
constexpr size_t BT_OFF_BINS = BT_OFF_ROWS + (size_t)TK_RMAX * (sizeof(BtRow) + 4 + 4 + 4 + 4 + 4 + 4);
constexpr size_t BT_OFF_LIST = BT_OFF_BINS + (size_t)BT_BINS * 4;   // the displaced blocks of the task
constexpr uint32_t BT_LIST_CAP = (40960 - 128 - BT_OFF_LIST) / 4;   // (the last 128 bytes: the counters of SPADA_TASK_DBG builds)
__host__ __device__ constexpr size_t batch_lds() { return BT_OFF_LIST + (size_t)BT_LIST_CAP * 4; }
static_assert(batch_lds() <= 40960, "four workgroups per CU");
static_assert(BT_T == 6 * BW && BT_PMAX == 4u * BW && BT_EMAX == (uint32_t)BW && BT_T * 8 == BT_PMAX * 12,
              "the LDS map and the per-thread arrays are written for these sizes");
static_assert(TK_RMAX <= 128 && BT_T <= 4096 && BT_PMAX <= 4096, "field widths of the product state");
static_assert(BT_LAYOUT + BT_PROBE_MAX < BT_T, "the first attempt's walk stays inside the table");

// The task loop runs every batch task in two parts.  batch_prologue -- descriptor -> row records / A entries -> (range: narrowing
// searches) -> scans -> entry records, tail bits and row regions in LDS -- touches neither the table nor the outputs of the task
// before; the loop runs it for the NEXT task at the point where the current one has nothing left to do but wait for its position in
// C (the chain) and store: three dependent round trips and three barriers that then lie under the chain's latency (a hand-off
// between loaded CUs takes 3 - 5 us: MI355X_MICROARCH.md, handoff-1to1) instead of in front of the next task.  Only these uniform
// words cross over in registers:
struct BatchHead {
    uint32_t P, nent;       // products, A entries with products
    uint32_t NBK, ncopy;    // hashed products (by the rows' records), copied products
};
#define BT_LDS_MAP(smem)                                                                                                             \
    uint32_t *hdr = (uint32_t *)(smem);                                                                                              \
    uint32_t *keys = (uint32_t *)((smem) + BT_OFF_KEYS);                                                                             \
    uint32_t *masks = (uint32_t *)((smem) + BT_OFF_MASK);                                                                            \
    EntryRecNum *w_ent = (EntryRecNum *)((smem) + BT_OFF_ENT);                                                                       \
    uint16_t *fo = (uint16_t *)((smem) + BT_OFF_ENT);           /* first output of every slot (the entry records are dead by then) */ \
    uint32_t *fo32 = (uint32_t *)((smem) + BT_OFF_ENT);                                                                              \
    uint32_t *bm32 = (uint32_t *)((smem) + BT_OFF_TAIL);        /* tail bits (last product of every entry): 64 words */               \
    const unsigned long long *bm64 = (const unsigned long long *)((smem) + BT_OFF_TAIL);                                             \
    double *vals = (double *)((smem) + BT_OFF_KEYS);            /* 2048 values in the order of the task's slice of C */               \
    uint32_t *cols = (uint32_t *)((smem) + BT_OFF_KEYS + (size_t)BT_PMAX * 8);   /* ... and their composite keys */                  \
    BtRow *s_emit = (BtRow *)((smem) + BT_OFF_ROWS);                                                                                 \
    int32_t *s_delta = (int32_t *)(s_emit + TK_RMAX);           /* outputs of the task before the row - what the row's outputs are numbered from */ \
    uint32_t *s_cpo = (uint32_t *)(s_delta + TK_RMAX);          /* COPY rows: number of the row's first product */                    \
    uint32_t *s_hoff = s_cpo + TK_RMAX;                         /* hashed outputs before the row (COUNT: outputs of the row) */       \
    uint32_t *s_info = s_hoff + TK_RMAX;                        /* class | products << 3 of the row */                                \
    uint32_t *s_bin = s_info + TK_RMAX;                         /* equalised mapping: first bin | bins << 16 of the row */            \
    uint32_t *s_span = s_bin + TK_RMAX;                         /* ... blocks between its first and last possible block */            \
    uint32_t *bins = (uint32_t *)((smem) + BT_OFF_BINS);        /* equalised mapping: products per bin, then first slot | slots << 16 */ \
    uint32_t *dlist = (uint32_t *)((smem) + BT_OFF_LIST);       /* displaced blocks: slot | slots above the home << 12; hdr[41] of them */ \
    /* scan slots (eight words each) in the header; hdr[40] = a probe sequence reached the end of the table; hdr[48 .. 50] belong  \
       to the chain and the ticket, hdr[52 .. 53] to the numeric base */                                                             \
    uint32_t *slot_rows = hdr, *slot_ent = hdr + 8, *slot_sp = hdr + 16, *slot_cnt = hdr + 24, *slot_pc = hdr + 32;                  \
    (void)keys; (void)masks; (void)w_ent; (void)fo; (void)fo32; (void)bm32; (void)bm64; (void)vals; (void)cols; (void)s_emit;        \
    (void)s_delta; (void)s_cpo; (void)s_hoff; (void)s_info; (void)dlist; (void)bins; (void)s_bin; (void)s_span; (void)slot_rows; (void)slot_ent; (void)slot_sp; (void)slot_cnt; (void)slot_pc

// keys (EMPTY), then masks (0): 1536 uint4, three per thread (a dense task has no keys)
template <bool DENSE>
__device__ inline void batch_clear_table(uint32_t *keys, int tid)
{
    uint4 *k4 = (uint4 *)keys;
    uint32_t zero = 0u;   // (opaque: a constant vector would be built before the task loop and kept -- spilled -- across it)
    asm volatile("" : "+v"(zero));
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const uint32_t i = (uint32_t)tid + (uint32_t)s * BW;
        const uint32_t fill = i < BT_T / 4 ? ~zero : zero;
        if (!DENSE || i >= BT_T / 4) k4[i] = make_uint4(fill, fill, fill, fill);
    }
}

template <int MODE, bool DENSE, bool SPILL = false, class ARGS>
__device__ inline BatchHead batch_prologue(const ARGS &g, const TaskDesc &td, uint32_t t, unsigned char *smem)
{
    constexpr bool VALUES = MODE != MODE_COUNT;
    constexpr uint32_t T = BT_T;
    BT_LDS_MAP(smem);
    // (the thread number is made opaque per task: what is derived from it -- LDS addresses, lane numbers, masks -- is then computed
    // where it is used instead of once before the task loop, where it would be held in registers, i.e. spilled, across every task;
    // a scratch reload waits for ALL vector loads in flight (one counter) and cuts the one-round-trip prologue into pieces)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    asm volatile("; BT_MARK p0" ::: "memory");
    constexpr bool spill = SPILL;
    const bool range = td.kind != TASK_BATCH;
    const uint32_t rb = td.row, R = range ? 1u : (td.np & 0xFFu), E = spill ? 0u : range ? (td.first >> 1) : ((td.np >> 8) & 0x3FFu),
                   PT = range ? td.np : ((td.np >> 18) & 0xFFFu);

    // Only the waves that hold a row or an entry of the task do any of this (a task of the web input has 22 rows and 108 entries:
    // two of the eight waves); the others meet them at the barrier and read the totals.  Every instruction of a stage is paid once
    // per WAVE, and with eight waves per task the stages' fixed costs, not the products, were most of the kernel's instructions.
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t nw = max(max((R + 63u) >> 6, (E + 63u) >> 6), 1u);
    const bool busy = wave_u < nw;   // (uniform)

    // ---- everything the walk needs in ONE round trip ------------------------------------------------------------------------
    RowRec rr{0u, 0u, 0u, (uint32_t)CLS_EMPTY};
    if (range) {
        if (tid == 0) rr = RowRec{td.col_lo, td.col_hi, td.np, (uint32_t)CLS_SOLO};
    } else if ((uint32_t)tid < R) {
        rr = (g.row_rec + rb)[(uint32_t)tid];   // (uniform base + 32-bit thread offset: one shift per address)
    }
    uint64_t b0 = 0;
    uint32_t len = 0, elr = 0;
    double av = 0.0;
    bool ecopy = false;
    if (busy && PT && (uint32_t)tid < E) {
        const uint32_t ti = (uint32_t)tid;   // (the entries of a task are contiguous: uniform base + 32-bit thread offset)
        b0 = (g.eb0 + td.src)[ti];
        len = (g.elen + td.src)[ti];
        if constexpr (VALUES) av = (g.aval + td.src)[ti];
        if (range) {
            // DIRECT RANGE: every selected B row narrowed to the range's columns -- B rows are ascending, and k_big_cuts has left
            // the positions of the range's bounds in every one of them: the same round trip as the entry itself
            if (g.cuts && td.cut != ~0ull) {
                const uint32_t c_lo = (g.cuts + td.cut)[ti], c_hi = (g.cuts + td.cut + E)[ti];
                b0 += c_lo;
                len = c_hi - c_lo;
            } else {
                // (a row without a cut table -- k_big_plan -- or SPADA_CUT_TABLE=0: the searches inside the task -- l1 = first position with column >= lo, l2 = first with column
                // > hi, all searches of the wave in lock step)
                const uint32_t lo = td.col_lo, hi = td.col_hi;
                const uint32_t *__restrict__ bidx = g.bidx;
                uint32_t l1 = 0, l2 = 0, n1 = len, n2 = len;
                while (n1 | n2) {
                    const uint32_t c1 = n1 ? bidx[b0 + l1 + (n1 >> 1)] : 0u, c2 = n2 ? bidx[b0 + l2 + (n2 >> 1)] : 0u;
                    if (n1) {
                        const uint32_t hh = n1 >> 1;
                        if (c1 < lo) {
                            l1 += hh + 1;
                            n1 -= hh + 1;
                        } else {
                            n1 = hh;
                        }
                    }
                    if (n2) {
                        const uint32_t hh = n2 >> 1;
                        if (c2 <= hi) {
                            l2 += hh + 1;
                            n2 -= hh + 1;
                        } else {
                            n2 = hh;
                        }
                    }
                }
                b0 += l1;
                len = l2 - l1;
            }
        }
        if (!range) {
            // a COPY row is a row with ONE entry (row_class): the entry's neighbours belong to other rows (batches hold whole rows)
            const uint32_t *ar = g.arow + td.src;
            const uint32_t row = ar[ti];
            const uint32_t prev = tid > 0 ? ar[ti - 1u] : 0xFFFFFFFFu, nxt = ti + 1u < E ? ar[ti + 1u] : 0xFFFFFFFFu;
            elr = row - (uint32_t)g.r0 - rb;
            ecopy = prev != row && nxt != row;
        }
    }
    if (MODE == MODE_NUMERIC && tid == 0) {
        const uint64_t c0 = range ? g.range_out[t] : g.cptr[rb];
        hdr[52] = (uint32_t)c0;
        hdr[53] = (uint32_t)(c0 >> 32);
    }
    if (tid < 64) bm32[tid] = 0u;
    if (tid == 0) {
        hdr[40] = 0u;
        hdr[41] = 0u;
        hdr[42] = 0u;
    }
    // one-pass mode: the table of the task before is free by now (its stores are done): cleared here, under the loads above
    if constexpr (MODE == MODE_FUSED) batch_clear_table<DENSE>(keys, tid);
    asm volatile("; BT_MARK p1" ::: "memory");
    // rows: the hashed products before every row (the table's slots are laid out over the rows in proportion to them) and the
    // outputs of COPY rows before it -- both known before anything is expanded.  Entries: numbered densely, and their products
    const bool row_hashed = rr.cls == CLS_SMALL || rr.cls == CLS_SOLO;
    const uint32_t row_pr = row_hashed ? rr.nprod : 0u, row_cp = rr.cls == CLS_COPY ? rr.nprod : 0u;
    const uint32_t bmin = rr.kmin >> BT_BSHIFT, spanb = row_hashed && rr.nprod ? (rr.kmax >> BT_BSHIFT) - bmin + 1u : 0u;
    const uint32_t row_v = row_pr | (row_cp << 16), ent_v = len ? ((1u << 16) | len) : 0u, sp_v = min(spanb, 2u * T);
    uint32_t row_inc = 0, ent_inc = 0, sp_inc = 0;
    if (busy) {
        row_inc = scan_part(row_v, slot_rows, tid);
        if (!spill) ent_inc = scan_part(ent_v, slot_ent, tid);
        if constexpr (DENSE) sp_inc = scan_part(sp_v, slot_sp, tid);
    }
    __syncthreads();   // (the tail bits are cleared)
    uint32_t row_tot, tot32 = PT, ex32 = 0, sp_tot = 0, soff = 0;   // (a spilled range: PT <= BT_PMAX products, no entries -- the dispatch in k_task sees to it)
    const uint32_t row_ex = scan_done<BT_NWAVE>(row_inc, row_v, slot_rows, &row_tot, tid, nw);
    if (!spill) ex32 = scan_done<BT_NWAVE>(ent_inc, ent_v, slot_ent, &tot32, tid, nw);
    if constexpr (DENSE) soff = scan_done<BT_NWAVE>(sp_inc, sp_v, slot_sp, &sp_tot, tid, nw);
    const uint32_t boff = row_ex & 0xFFFFu, cpre = row_ex >> 16, NBK = row_tot & 0xFFFFu;   // hashed products before the row, copied outputs before it
    BatchHead hd{tot32 & 0xFFFFu, tot32 >> 16, NBK, row_tot >> 16};
    if ((uint32_t)tid < R) {
        uint32_t S, G;   // first slot of the row's region, its slots
        float scale = 1.0f;
        if constexpr (DENSE) {
            S = soff;
            G = spanb;
        } else {
            // (floor(x * f) is monotone in x, and the row behind starts where this one ends: the same expression of the same number.
            // f >= 1.49: a row's region has at least as many slots as the row has products, i.e. blocks)
            const float f = (float)BT_LAYOUT / (float)max(NBK, 1u);
            S = min((uint32_t)((float)boff * f), BT_LAYOUT);
            G = max(min((uint32_t)((float)(boff + row_pr) * f), BT_LAYOUT) - S, 1u);
            if (spanb > G) scale = (float)G / (float)spanb;
        }
        s_emit[tid] = BtRow{S, G, bmin, scale};
        if constexpr (!DENSE) s_span[tid] = max(spanb, 1u);   // (for the second attempt of a task whose blocks cluster, batch_main)
        s_delta[tid] = (int32_t)cpre;
        s_hoff[tid] = 0u;
        s_info[tid] = rr.cls | (min(rr.nprod, 0xFFFFFFu) << 3);
    }
    if (hd.P > BT_PMAX || (DENSE && sp_tot > T)) {   // the cut / the dispatch guarantee it; a batch that does not fit is an internal error, not a memory fault
        if (tid == 0 && atomicOr(&g.ctr->abort_flag, 32u) == 0u) {   // what did not fit (reported by the host)
            g.ctr->dbg[0] = td.kind | ((unsigned long long)E << 8) | ((unsigned long long)R << 32);
            g.ctr->dbg[1] = tot32;
            g.ctr->dbg[2] = td.np;
            g.ctr->dbg[3] = t;
        }
        hd.P = 0;
    }
    if (busy && hd.P && len) {
        // record: (begin - first product) mod 2^48 | local row << 48 | copy << 55, A value
        const uint32_t ci = ex32 >> 16, po = ex32 & 0xFFFFu;
        w_ent[ci] = EntryRecNum{((b0 - po) & M48) | ((uint64_t)elr << 48) | ((uint64_t)(ecopy ? 1u : 0u) << 55), av};
        atomicOr(&bm32[(po + len - 1u) >> 5], 1u << ((po + len - 1u) & 31));   // TAIL bit: the entry's last product
        if (ecopy) s_cpo[elr] = po;   // (the row's one entry: product number - first product = place in the row)
    }
    asm volatile("; BT_MARK p2" ::: "memory");
    return hd;   // (no barrier: batch_main starts with one)
}

// `next()` -- called once, by all threads, where the task has nothing left to do but wait for its position and store: takes the
// next ticket and runs the next task's prologue
// `early()` -- called once, by all threads, where the products have arrived and the task turns to LDS for a long while: the modes
// without a chain take the next ticket here (the atomic's round trip lies under the insertion; see k_task)
template <int MODE, bool DENSE, bool SPILL = false, class ARGS, class NEXT, class EARLY>
__device__ inline void batch_main(const ARGS &g, const TaskDesc &td, uint32_t t, uint32_t ntasks, unsigned char *smem,
                                  uint32_t *dbg_ph /* LDS: SPADA_TASK_DBG builds */, const BatchHead hd, NEXT &&next, EARLY &&early)
{
    constexpr bool VALUES = MODE != MODE_COUNT;
    constexpr uint32_t T = BT_T;
    BT_LDS_MAP(smem);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    // stage boundaries: a comment in the ISA (static instruction counts per stage: scripts/dev/isa_stages.py) and, in SPADA_TASK_DBG
    // builds, the clock ticks of thread 0 per stage, summed per workgroup in `dbg_ph` (k_task adds them to TaskCounters::dbg at its end)
    unsigned long long ph_prev = SPADA_TASK_DBG ? __builtin_amdgcn_s_memtime() : 0;
#define BMARK(i)                                                                \
    do {                                                                        \
        asm volatile("; BT_MARK " #i ::: "memory");                             \
        if (SPADA_TASK_DBG && tid == 0 && (i) > 0) {                            \
            const unsigned long long n_ = __builtin_amdgcn_s_memtime();         \
            dbg_ph[(i) - 1] += (uint32_t)(n_ - ph_prev);                        \
            ph_prev = n_;                                                       \
        }                                                                       \
    } while (0)
    if (SPADA_TASK_DBG && tid == 0) dbg_ph[8] += 1;
    uint32_t dbg_t_start = SPADA_TASK_DBG ? (uint32_t)__builtin_amdgcn_s_memtime() : 0u, dbg_t_gath = 0u;
    (void)dbg_t_start;
    (void)dbg_t_gath;
    // (development: the numeric mode cut short behind stage k; `sink` keeps what the stage computed alive)
#define BSTOP(k, sink)                                                              \
    do {                                                                            \
        if (MODE == MODE_NUMERIC && ((SPADA_ABLATE >> 8) & 15) == (k)) {            \
            if ((sink) == 0x9E3779B1u) hdr[60] = 1u;                                \
            next();                                                                 \
            __syncthreads();                                                        \
            return;                                                                 \
        }                                                                           \
    } while (0)
    BMARK(0);
    constexpr bool spill = SPILL;
    const bool range = td.kind != TASK_BATCH;
    const uint32_t rb = td.row, R = range ? 1u : (td.np & 0xFFu), E = spill ? 0u : range ? (td.first >> 1) : ((td.np >> 8) & 0x3FFu);
    (void)E;
    const uint32_t colbits = g.colbits;                       // >= BT_BSHIFT + 1 (the engine sees to it)
    const uint32_t colmask = colbits >= 32 ? 0xFFFFFFFFu : ((1u << colbits) - 1u);
    const uint32_t hshift = colbits >= 32 ? 0u : colbits - BT_BSHIFT;   // block key = composite key >> 5 = local row << hshift | block
    auto lr_of_ck = [&](uint32_t ck) { return colbits >= 32 ? 0u : ck >> colbits; };
    const uint32_t P = hd.P, nent = hd.nent, NBK = hd.NBK;
    // (one-pass mode: the prologue has cleared the table under its loads -- it runs behind the stores of the task before, when the table
    // is free; in the other modes it runs UNDER those stores, and the table is cleared here)
    if constexpr (MODE != MODE_FUSED) batch_clear_table<DENSE>(keys, tid);
    BMARK(1);
    __syncthreads();   // the table is cleared; the prologue's records, tail bits and row regions are written
    // tails before every 64-bit word of the bitmap (32 words): every wave scans them for itself and keeps the prefixes in the
    // lanes of one register (word w in lane w): the per-segment values are then scalar reads, and no further barrier is needed
    uint32_t tail_pre = 0;
    if (!spill) {
        const uint32_t c = (uint32_t)__popcll(bm64[lane & 31]);
        const uint32_t inc = wave_scan_incl_u32(lane < 32 ? c : 0u);
        tail_pre = inc - c;
    }
    BSTOP(1, tail_pre);

    BMARK(2);
    // ---- expand - scale - accumulate blocks (scheduler.rs:482-606, simulator.rs:892-953, :86-111) ---------------------------
    uint32_t r_ck[4], r_st[4];   // the products of this thread: composite key (local row << colbits | column), state (above)
    double r_v[4];
    uint32_t mynew = 0, mykeys = 0;   // mask bits this lane has set | blocks (table keys) this lane has created
    uint32_t NO = 0, NBt = 0, total = 0;   // hashed outputs, blocks, outputs of the task
    {
        // lane l of a segment holds product seg + l, i.e. bit l of one bitmap word: the word and its prefix are wave-uniform
        // reads; the entry of a product = the entries that END before it = tails before the word + tails below the lane (a
        // v_mbcnt pair).  Lanes past the end take the last product (a valid address; only their atomics are switched off).
        uint32_t pp[4];
        bool act[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t p = ((uint32_t)u * BT_NWAVE + wave_u) * 64u + lane;
            act[u] = p < P;
            pp[u] = min(p, P ? P - 1u : 0u);
        }
        uint32_t col[4], lrc[4];   // column | local row, copy << 7
        if (P == 0 || (SPADA_ABLATE & 32)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                col[u] = (SPADA_ABLATE & 32) ? pp[u] * 7u : 0u;
                lrc[u] = 0u;
                r_v[u] = 0.0;
            }
        } else if (spill) {
            // the slice holds (column, a * b) of the range's products in any order: product p is element p
#pragma unroll
            for (int u = 0; u < 4; ++u) col[u] = g.scr_col[td.src + pp[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                lrc[u] = 0u;
                r_v[u] = 0.0;
                if constexpr (VALUES) r_v[u] = g.scr_val[td.src + pp[u]];
            }
        } else {
            uint32_t j[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t w = (uint32_t)u * BT_NWAVE + wave_u;   // = segment / 64
                const unsigned long long bits = bm64[w];
                const uint32_t bp = (uint32_t)__builtin_amdgcn_readlane((int)tail_pre, (int)w);
                // (a clamped lane sits past the last tail of its word: it counts every tail of the word, i.e. one entry too
                // many whenever the last product is in this word -- min() with the last entry puts it back)
                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bits, 0u));
                j[u] = min(bp + below, nent - 1u);
            }
            double a_[4];
            if (g.b_off32) {
                // nnz(B) < 2^29: the byte offsets of both gathers fit 32 bits -- one shift per address on top of the scalar base
                // instead of 64-bit adds and shifts (eight vector instructions per product less)
                uint32_t q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const EntryRecNum er = w_ent[j[u]];
                    q[u] = (uint32_t)er.pack + pp[u];
                    lrc[u] = (uint32_t)(er.pack >> 48);
                    a_[u] = er.av;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) col[u] = *(const uint32_t *)((const char *)g.bidx + (uint32_t)(q[u] << 2));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    r_v[u] = 0.0;
                    if constexpr (VALUES) r_v[u] = a_[u] * *(const double *)((const char *)g.bval + (uint32_t)(q[u] << 3));   // simulator.rs:100-101
                }
            } else {
                uint64_t q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const EntryRecNum er = w_ent[j[u]];
                    q[u] = ((er.pack & M48) + pp[u]) & M48;
                    lrc[u] = (uint32_t)(er.pack >> 48);
                    a_[u] = er.av;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) col[u] = g.bidx[q[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    r_v[u] = 0.0;
                    if constexpr (VALUES) r_v[u] = a_[u] * g.bval[q[u]];   // simulator.rs:100-101
                }
            }
        }
        if (SPADA_TASK_DBG) dbg_t_gath = (uint32_t)__builtin_amdgcn_s_memtime() + (col[0] & 0u);
        BSTOP(2, col[0] ^ col[1] ^ col[2] ^ col[3] ^ lrc[0] ^ lrc[1] ^ lrc[2] ^ lrc[3] ^ (uint32_t)__double2loint(r_v[0] + r_v[1] + r_v[2] + r_v[3]));
        early();
        // The home slot of a block is a LINEAR function of its column inside the row's span: when the columns cluster -- the rows of
        // an R-MAT graph: a third of a row's blocks on a twentieth of its span -- the blocks of a cluster share a few home slots and
        // linear probing pays for it quadratically (such tasks took 60 - 100 us to count their outputs, and a thousand tasks of the
        // chain waited for each of them).  So the first attempt gives up where a block is displaced by BT_PROBE_MAX slots, and the
        // task starts over with EQUALISED home slots: it counts its products in BT_BINS bins of the rows' spans (every row its own
        // bins, s_bin), gives every bin slots in proportion to its products -- at least as many: a bin cannot overflow by itself --
        // and interpolates inside the bin.  Still monotone in (row, column), and as even as the task's own histogram makes it.
        // (the insertion of the hashed products: EQ = the second attempt, with equalised home slots)
        auto insert_hashed = [&](auto EQ) {
            constexpr bool equalised = decltype(EQ)::value;
            constexpr bool COUNT_HASHED = MODE == MODE_COUNT;   // (web count kernel 0.52 -> 0.42 ms against monotone home slots)
            mynew = mykeys = 0u;
            // the home slots of the thread's four products first -- their row parameters are four independent LDS reads, in flight together --
            // then the four insertions: a step's chain of dependent LDS round trips is the CAS (and its probes) and the OR, not the row read
            // in front of them (the insertions are what a task's count, i.e. everything behind it in the chain, waits for)
            uint32_t homes[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t lr = lrc[u] & 127u;
                const bool copy = (lrc[u] & 128u) != 0u, hashed = act[u] && !copy;
                const uint32_t ck = compose_key(lr, col[u], colbits);
                r_ck[u] = ck;
                r_st[u] = act[u] ? BT_H_COPY : BT_H_NONE;
                uint32_t home = 0u;
                if (hashed) {
                    const uint32_t hk = ck >> BT_BSHIFT;
                    if constexpr (COUNT_HASHED) {
                        // the counting mode orders nothing: a hashed home slot -- no clusters whatever the columns are, no second
                        // attempt, no list of displaced blocks, no row record to read
                        home = __umulhi(hk * 0x9E3779B1u, T);
                    } else {
                    const BtRow e = s_emit[lr];
                    const uint32_t d = (col[u] >> BT_BSHIFT) - e.bmin;
                    home = e.s + min((uint32_t)((float)d * e.scale), e.g - 1u);
                    if (equalised) {
                        const uint32_t rb_ = s_bin[lr], nb = rb_ >> 16;
                        const float x = (float)d * ((float)nb / (float)s_span[lr]);
                        const uint32_t k = min((uint32_t)x, nb - 1u), bw = bins[(rb_ & 0xFFFFu) + k], bg = bw >> 16;
                        home = (bw & 0xFFFFu) + min((uint32_t)((x - (float)k) * (float)bg), bg - 1u);
                    }
                    }
                }
                homes[u] = home;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool hashed = r_st[u] == BT_H_COPY && !(lrc[u] & 128u);
                if (hashed) {
                    const uint32_t hk = r_ck[u] >> BT_BSHIFT, home = homes[u];
                    uint32_t h = home, st = 0;
                    uint32_t old = (SPADA_ABLATE & 16) ? EMPTY_KEY : atomicCAS(&keys[h], EMPTY_KEY, hk);
                    if (old != EMPTY_KEY && old != hk) {
                        // upwards; at the end of the table the free slot BELOW the home takes the block (the order stage then looks at
                        // whole clusters).  First attempt: BT_PROBE_MAX slots above its home the lane gives up -- the blocks cluster,
                        // the task starts over with equalised home slots -- and leaves the loop by "finding" the key of its home slot
                        // (no exit of its own: the common iteration is the one of a loop without a budget)
                        uint32_t hkx = hk;
                        if constexpr (!equalised && !COUNT_HASHED) {
                            // (first attempt: home + BT_PROBE_MAX lies inside the table -- the rows' regions end BT_T - BT_LAYOUT slots before
                            // its end -- so the walk never wraps and needs no direction: one compare per step besides the CAS)
                            const uint32_t hlim = home + BT_PROBE_MAX;
                            do {
                                if (++h == hlim) {
                                    hdr[42] = 1u;
                                    h = home;
                                    hkx = keys[home];
                                }
                                old = atomicCAS(&keys[h], EMPTY_KEY, hkx);
                            } while (old != EMPTY_KEY && old != hkx);
                        } else {
                            do {
                                if (h >= home) {
                                    if (++h == T) {
                                        h = home - 1u;
                                        hdr[40] = 1u;
                                    }
                                } else {
                                    --h;
                                }
                                old = atomicCAS(&keys[h], EMPTY_KEY, hkx);
                            } while (old != EMPTY_KEY && old != hkx);
                        }
                        if (!COUNT_HASHED && old == EMPTY_KEY && h > home) {   // displaced: the order stage looks at the slots between its home and its place
                            const uint32_t li = atomicAdd(&hdr[41], 1u);
                            if (li < BT_LIST_CAP) dlist[li] = h | ((h - home) << 12);
                        }
                    }
                    if (old == EMPTY_KEY) {
                        ++mykeys;
                        st = BT_ST_CREATOR;
                    }
                    const uint32_t bit = 1u << (col[u] & 31u);
                    const uint32_t was = (SPADA_ABLATE & 16) ? 0u : atomicOr(&masks[h], bit);
                    if (!(was & bit)) {
                        ++mynew;
                        st |= BT_ST_OWNER;
                    }
                    r_st[u] = h | st;
                }
            }
        };
        auto count_outputs = [&]() {
            // ---- the count of the task: new mask bits + copied products (known from the rows) -----------------------------------------
        {
            uint32_t tot;
            (void)block_scan_excl_dpp_n<BT_NWAVE>(mynew | (mykeys << 16), slot_cnt, &tot, tid);   // (its barrier: the table is complete)
            NO = tot & 0xFFFFu;
            NBt = tot >> 16;
            total = NO + hd.ncopy;   // + the products of the COPY rows
        }
        };
        if constexpr (DENSE) {
            // no keys: the slot is the block's place in the row.  The four products of a thread go through the two steps -- row
            // parameters, mask -- together, so that the four LDS operations of a step are in flight at once
            bool hashed[4];
            BtRow e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t lr = lrc[u] & 127u;
                hashed[u] = act[u] && !(lrc[u] & 128u);
                r_ck[u] = compose_key(lr, col[u], colbits);
                r_st[u] = act[u] ? BT_H_COPY : BT_H_NONE;
                e[u] = s_emit[hashed[u] ? lr : 0u];
            }
            uint32_t was[4], h[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                h[u] = e[u].s + (col[u] >> BT_BSHIFT) - e[u].bmin;
                was[u] = 0xFFFFFFFFu;
                if (hashed[u]) was[u] = atomicOr(&masks[h[u]], 1u << (col[u] & 31u));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t fresh = ((was[u] >> (col[u] & 31u)) & 1u) ^ 1u;
                mynew += fresh;
                if (hashed[u]) r_st[u] = h[u] | (fresh ? BT_ST_OWNER : 0u);
            }
        } else {
            insert_hashed(std::false_type{});
        }
        count_outputs();
        if constexpr (!DENSE) {
            if (MODE != MODE_COUNT && hdr[42] != 0u) {   // (uniform, rare: a block of the first attempt was displaced too far)
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // (the products again from their keys and states: nothing else stays live across the count)
                const uint32_t st = r_st[u], ck = r_ck[u];
                act[u] = st != BT_H_NONE;
                col[u] = colbits >= 32 ? ck : ck & ((1u << colbits) - 1u);
                lrc[u] = (colbits >= 32 ? 0u : ck >> colbits) | (st == BT_H_COPY ? 128u : 0u);
            }
            {   // the rows' bins: every hashed row at least one, the rest in proportion to the products -- disjoint, ascending with the rows
                const uint32_t R = td.kind != TASK_BATCH ? 1u : (td.np & 0xFFu);
                uint32_t row_pr = 0;
                if ((uint32_t)tid < R) {
                    const uint32_t info = s_info[tid], cls = info & 7u;
                    if (cls == CLS_SMALL || cls == CLS_SOLO) row_pr = info >> 3;
                }
                uint32_t rtot;
                const uint32_t rex = block_scan_excl_dpp_n<BT_NWAVE>(row_pr | ((row_pr ? 1u : 0u) << 16), slot_rows, &rtot, tid);
                if ((uint32_t)tid < R) {
                    const uint32_t nh = rtot >> 16, hrb = rex >> 16, boff = rex & 0xFFFFu;
                    const float fb = (float)(BT_BINS - nh) / (float)max(rtot & 0xFFFFu, 1u);
                    const uint32_t b0 = hrb + (uint32_t)((float)boff * fb), b1 = hrb + (row_pr ? 1u : 0u) + (uint32_t)((float)(boff + row_pr) * fb);
                    s_bin[tid] = min(b0, BT_BINS - 1u) | (max(min(b1, BT_BINS) - min(b0, BT_BINS - 1u), 1u) << 16);
                }
            }
            {   // the table again, and the marks of the first attempt
                uint4 *k4 = (uint4 *)keys;
                uint32_t zero = 0u;
                asm volatile("" : "+v"(zero));
#pragma unroll
                for (int s2 = 0; s2 < 3; ++s2) {
                    const uint32_t i = (uint32_t)tid + (uint32_t)s2 * BW;
                    const uint32_t fill = i < T / 4 ? ~zero : zero;
                    k4[i] = make_uint4(fill, fill, fill, fill);
                }
                if (tid < (int)BT_BINS) bins[tid] = 0u;
                if (tid == 0) {
                    hdr[40] = 0u;
                    hdr[41] = 0u;
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // (the lanes of a wave hold consecutive products of ascending columns: runs of lanes share a bin, one atomic per run)
                const bool on = act[u] && !(lrc[u] & 128u);
                uint32_t b = 0xFFFFFFFFu;
                if (on) {
                    const uint32_t lr = lrc[u] & 127u, rb_ = s_bin[lr], nb = rb_ >> 16;
                    const float x = (float)((col[u] >> BT_BSHIFT) - s_emit[lr].bmin) * ((float)nb / (float)s_span[lr]);
                    b = (rb_ & 0xFFFFu) + min((uint32_t)x, nb - 1u);
                }
                const uint32_t prev = (uint32_t)__shfl_up((int)b, 1);
                const bool head = lane == 0 || prev != b;
                const unsigned long long heads = __ballot(head), above = (heads >> lane) >> 1;
                const uint32_t len = above ? (uint32_t)__ffsll((long long)above) : 64u - (uint32_t)lane;
                if (head && on) atomicAdd(&bins[b], len);
            }
            __syncthreads();
            if (tid < 64) {   // first slot of every bin: floor(products before it * slots / products), four bins per lane
                const uint4 c = ((const uint4 *)bins)[lane];
                const uint32_t sum = c.x + c.y + c.z + c.w, inc = wave_scan_incl_u32(sum), ex = inc - sum;
                const float f = (float)BT_LAYOUT / (float)max(NBK, 1u);
                const uint32_t p0 = ex, p1 = p0 + c.x, p2 = p1 + c.y, p3 = p2 + c.z, p4 = p3 + c.w;
                const uint32_t s0 = min((uint32_t)((float)p0 * f), BT_LAYOUT), s1 = min((uint32_t)((float)p1 * f), BT_LAYOUT),
                               s2 = min((uint32_t)((float)p2 * f), BT_LAYOUT), s3 = min((uint32_t)((float)p3 * f), BT_LAYOUT),
                               s4 = min((uint32_t)((float)p4 * f), BT_LAYOUT);
                ((uint4 *)bins)[lane] = make_uint4(s0 | (max(s1 - s0, 1u) << 16), s1 | (max(s2 - s1, 1u) << 16), s2 | (max(s3 - s2, 1u) << 16),
                                                   s3 | (max(s4 - s3, 1u) << 16));
            }
            __syncthreads();
                insert_hashed(std::true_type{});
                count_outputs();
            }
        }
    }
    BMARK(3);
    BSTOP(3, r_st[0] ^ r_st[1] ^ r_st[2] ^ r_st[3] ^ r_ck[0] ^ r_ck[1] ^ r_ck[2] ^ r_ck[3] ^ total ^ (uint32_t)__double2loint(r_v[0] + r_v[1] + r_v[2] + r_v[3]));
    (void)NBt;
    if constexpr (MODE != MODE_NUMERIC) task_publish<MODE>(g, t, total);
    if (SPADA_TASK_DBG && tid == 0) {   // ticks from the task's ticket (dbg_ph[16], k_task) to its publication, per kind: sum, tasks, maximum
        const uint32_t d = (uint32_t)__builtin_amdgcn_s_memtime() - dbg_ph[16], k = range ? 20u : 17u;
        dbg_ph[k] += d >> 4;
        dbg_ph[k + 1] += 1u;
        dbg_ph[k + 2] = max(dbg_ph[k + 2], d);
        {   // histogram of that latency (bins end at 16 / 20 / 24 / 28 / 32 / 40 / 60 thousand ticks), and the tasks that took the second attempt
            const uint32_t b = d < 16000u ? 0u : d < 20000u ? 1u : d < 24000u ? 2u : d < 28000u ? 3u : d < 32000u ? 4u : d < 40000u ? 5u : d < 60000u ? 6u : 7u;
            dbg_ph[24 + b] += 1u;
            if (!DENSE && hdr[42]) dbg_ph[23] += 1u;
        }
#if SPADA_TASK_DBG
        for (int grp = d > 30000u ? 0 : 1; grp < 2; ++grp) {   // late tasks | all tasks: what they are, and where their time went
            unsigned long long *q = g.ctr->dbgs[blockIdx.x & 2047u][grp];
            q[0] += 1ull;
            q[1] += P;
            q[2] += E;
            q[3] += R;
            q[4] += hdr[41];
            q[5] += NO;
            q[6] += !DENSE && hdr[42] ? 1u : 0u;
            q[7] += DENSE ? 1u : 0u;
            q[8] += range ? 1u : 0u;
            q[9] += dbg_t_start - dbg_ph[16];
            q[10] += dbg_t_gath - dbg_ph[16];
            q[11] += d;
            q[12] += t + 1000u >= ntasks ? 1u : 0u;
        }
        if (d > 60000u) {   // the slowest tasks: what they are
            const unsigned long long n = atomicAdd(&g.ctr->dbgh[2][0], 1ull);
            if (n < 7) {
                g.ctr->dbgh[2][1 + 3 * n] = t | ((unsigned long long)td.kind << 32) | ((unsigned long long)(DENSE ? 1 : 0) << 40);
                g.ctr->dbgh[2][2 + 3 * n] = P | ((unsigned long long)NBt << 16) | ((unsigned long long)hdr[41] << 32) | ((unsigned long long)E << 48);
                g.ctr->dbgh[2][3 + 3 * n] = d | ((unsigned long long)NO << 32);
            }
        }
#endif
    }
    if (SPADA_TASK_DBG) {   // shape of the tasks: products, hashed outputs, blocks, slots between home and place, entries
        if (tid == 0) {
            dbg_ph[9 + 3] += hdr[41];   // displaced blocks
            dbg_ph[9 + 0] += P;
            dbg_ph[9 + 1] += NO;
            dbg_ph[9 + 2] += NBt;
            dbg_ph[9 + 4] += E;
            dbg_ph[9 + 5] += 1u;
        }
    }
    if constexpr (MODE == MODE_FUSED) __builtin_amdgcn_s_setprio(0);
    if constexpr (MODE == MODE_COUNT) {
        if (range) {   // (the count of a range task is all the position kernels need)
            next();
            return;
        }
        // the symbolic phase wants the outputs of every ROW: every hashed product that set a mask bit counts for its row; offsets
        // of the rows inside the batch -- k_pos4 adds the position of the batch afterwards
        if (NO) {
            if (R > 1) {
                if constexpr (DENSE) {
                    // the slots are in (row, block) order: outputs of a row = sum of the popcounts of its slots -- every thread adds
                    // up its six consecutive slots, a run of one row with one LDS atomic
                    const uint2 *m2 = (const uint2 *)(masks + 6u * (uint32_t)tid);
                    const uint2 wa = m2[0], wb = m2[1], wc = m2[2];
                    const uint32_t c[6] = {(uint32_t)__popc(wa.x), (uint32_t)__popc(wa.y), (uint32_t)__popc(wb.x), (uint32_t)__popc(wb.y),
                                           (uint32_t)__popc(wc.x), (uint32_t)__popc(wc.y)};
                    if (c[0] | c[1] | c[2] | c[3] | c[4] | c[5]) {
                        // row of slot x: the last row whose region starts at or before x (the regions of the hashed rows tile the slots)
                        uint32_t r = 0;
                        for (uint32_t stp = 64u; stp; stp >>= 1)
                            if (r + stp < R && s_emit[r + stp].s <= 6u * (uint32_t)tid) r += stp;
                        uint32_t acc = 0;
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            const uint32_t x = 6u * (uint32_t)tid + (uint32_t)k;
                            while (r + 1u < R && s_emit[r + 1u].s <= x) {
                                if (acc) atomicAdd(&s_hoff[r], acc);
                                acc = 0;
                                ++r;
                            }
                            acc += c[k];
                        }
                        if (acc) atomicAdd(&s_hoff[r], acc);
                    }
                } else {
                    // every block was created by exactly one product: it adds the block's columns to its row
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (r_st[u] < BT_ST_HASHED_MAX && (r_st[u] & BT_ST_CREATOR))
                            atomicAdd(&s_hoff[lr_of_ck(r_ck[u])], (uint32_t)__popc(masks[r_st[u] & 0xFFFu]));
                }
            } else if (tid == 0) {
                s_hoff[0] = NO;
            }
        }
        __syncthreads();
        uint32_t n = 0;
        if ((uint32_t)tid < R) {
            const uint32_t info = s_info[tid], cls = info & 7u;
            n = (cls == CLS_SMALL || cls == CLS_SOLO) ? s_hoff[tid] : (cls == CLS_COPY ? info >> 3 : 0u);
        }
        uint32_t tot;
        const uint32_t ooff = block_scan_excl_dpp_n<BT_NWAVE>(n, slot_pc, &tot, tid);
        if ((uint32_t)tid < R) g.cptr[rb + tid] = ooff;
        next();   // (the row arrays are read)
        return;
    }

    BMARK(4);
    // ---- order: first output of every block = outputs of the blocks before it in (row, block) order ----------------------------
    uint32_t hoff = 0;   // hashed outputs of the batch before this thread's row
    uint32_t rank[4] = {0u, 0u, 0u, 0u};
    if (NO && !(SPADA_ABLATE & 2)) {
        {
            // the slots are in (row, block) order up to the clusters: one prefix sum over the popcounts of the masks, six consecutive
            // slots per thread
            const uint2 *m2 = (const uint2 *)(masks + 6u * (uint32_t)tid);
            const uint2 wa = m2[0], wb = m2[1], wc = m2[2];
            const uint32_t c0 = (uint32_t)__popc(wa.x), c1 = (uint32_t)__popc(wa.y), c2 = (uint32_t)__popc(wb.x), c3 = (uint32_t)__popc(wb.y),
                           c4 = (uint32_t)__popc(wc.x), c5 = (uint32_t)__popc(wc.y);
            uint32_t tot;
            const uint32_t ex = block_scan_excl_dpp_n<BT_NWAVE>(c0 + c1 + c2 + c3 + c4 + c5, slot_pc, &tot, tid);
            const uint32_t f0 = ex, f1 = f0 + c0, f2 = f1 + c1, f3 = f2 + c2, f4 = f3 + c3, f5 = f4 + c4;
            uint32_t *f32 = fo32 + 3u * (uint32_t)tid;
            f32[0] = f0 | (f1 << 16);
            f32[1] = f2 | (f3 << 16);
            f32[2] = f4 | (f5 << 16);
        }
        __syncthreads();
        if constexpr (!DENSE) {
            // Inside a cluster of occupied slots the keys may be out of order.  Probing upwards only, an inversion -- x below y in
            // the table, key(x) > key(y) -- has home(y) <= home(x) <= slot(x) < slot(y): y was displaced, and x lies between y's
            // home and y's place.  So every displaced block looks at those slots: a larger key there moves behind it (its first
            // output grows by this block's outputs), and this block moves in front of it.  The first outputs are 16-bit halves of
            // 32-bit words: an atomic add of a (possibly negative) amount to the word is exact for the half it is meant for as
            // long as the final values fit, whatever the intermediate carries are.
            const uint32_t ndisp = hdr[41];
            const bool whole_clusters = hdr[40] != 0u || ndisp > BT_LIST_CAP;   // (uniform) a block was placed BELOW its home, or the list is full
            if (SPADA_ABLATE & 1) {
            } else if (!whole_clusters) {
                // (the list spread over the waves: entry e to lane e / 8 of wave e % 8 -- a wave waits for the longest walk among its lanes)
                for (uint32_t e = (uint32_t)lane * BT_NWAVE + wave_u; e < ndisp; e += BW) {
                    const uint32_t ent = dlist[e], h = ent & 0xFFFu, k = keys[h], mine = (uint32_t)__popc(masks[h]);
                    uint32_t before = 0;
                    for (uint32_t j = h - (ent >> 12); j < h; ++j)
                        if (keys[j] > k) {
                            atomicAdd(&fo32[j >> 1], mine << ((j & 1u) * 16u));
                            before += (uint32_t)__popc(masks[j]);
                        }
                    if (before) atomicAdd(&fo32[h >> 1], (0u - before) << ((h & 1u) * 16u));
                }
            } else {
                // any arrangement inside a cluster: first output = outputs before the cluster + outputs of its smaller keys
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t st = r_st[u];
                    if (st < BT_ST_HASHED_MAX && (st & BT_ST_CREATOR)) {
                        const uint32_t h = st & 0xFFFu, k = r_ck[u] >> BT_BSHIFT;
                        uint32_t adj = 0;
                        for (uint32_t j = h; j-- > 0u;) {
                            const uint32_t kj = keys[j];
                            if (kj == EMPTY_KEY) break;
                            if (kj > k) adj -= (uint32_t)__popc(masks[j]);
                        }
                        for (uint32_t j = h + 1u; j < T; ++j) {
                            const uint32_t kj = keys[j];
                            if (kj == EMPTY_KEY) break;
                            if (kj < k) adj += (uint32_t)__popc(masks[j]);
                        }
                        if (adj) atomicAdd(&fo32[h >> 1], adj << ((h & 1u) * 16u));   // (its own half: the neighbour's creator may be at work on the other one)
                    }
                }
            }
            __syncthreads();
        }
        BMARK(5);
        BSTOP(5, r_st[0] ^ r_st[1] ^ r_st[2] ^ r_st[3] ^ r_ck[0] ^ r_ck[1] ^ r_ck[2] ^ r_ck[3] ^ (uint32_t)fo[tid] ^ (uint32_t)__double2loint(r_v[0] + r_v[1] + r_v[2] + r_v[3]));
        // output of a product = first output of its block + mask bits below its own
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r_st[u] < BT_ST_HASHED_MAX) {
                const uint32_t h = r_st[u] & 0xFFFu;
                rank[u] = (uint32_t)fo[h] + (uint32_t)__popc(masks[h] & ((1u << (r_ck[u] & 31u)) - 1u));
            }
        }
        // the rows' first hashed outputs: the row's first possible block (the block of kmin: some product has that column) has the
        // row's first slot as its home and no smaller key behind it
        if (R > 1 && (uint32_t)tid < R) {
            uint32_t v = 0xFFFFFFFFu;
            const uint32_t info = s_info[tid], cls = info & 7u;
            if ((cls == CLS_SMALL || cls == CLS_SOLO) && (info >> 3)) {
                const BtRow e = s_emit[tid];
                uint32_t j = e.s;
                if constexpr (!DENSE) {
                    const uint32_t target = ((uint32_t)tid << hshift) | e.bmin;
                    while (j < T && keys[j] != target) ++j;
                    if (j >= T) {   // (placed below its home)
                        j = e.s;
                        while (j > 0u && keys[j] != target) --j;
                    }
                }
                v = fo[j];
            }
            s_hoff[tid] = v;
        }
    }
    __syncthreads();   // keys, masks and first outputs are read: the outputs may take their place
    if (NO && R > 1 && (uint32_t)tid < R) {
        // rows without hashed products take the value of the next row that has some (the values ascend with the rows)
        uint32_t v = s_hoff[tid];
        for (uint32_t r2 = (uint32_t)tid + 1u; v == 0xFFFFFFFFu && r2 < R; ++r2) v = s_hoff[r2];
        hoff = v == 0xFFFFFFFFu ? NO : v;
    }
    // the rows' first outputs inside the batch: hashed outputs before the row + copied outputs before it
    uint32_t ooff = 0;
    if ((uint32_t)tid < R) {
        ooff = hoff + (uint32_t)s_delta[tid];   // (still the copied outputs before the row)
        if ((s_info[tid] & 7u) == CLS_COPY) s_delta[tid] += (int32_t)hoff - (int32_t)s_cpo[tid];
    }
    // ---- scale - add: every retained product puts its value at its output (simulator.rs:213-218; order differs) ---------------
    // The task's outputs are assembled in LDS in the order of its slice of C: a hashed output at its rank + the copied outputs of
    // the rows before its row, a copied product at its number in its row + the outputs before the row.  The product that set a
    // mask bit first stores its value, the others of the same output add theirs afterwards.
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (!(SPADA_ABLATE & 4) && r_st[u] < BT_ST_HASHED_MAX && (r_st[u] & BT_ST_OWNER)) {
            const uint32_t i = rank[u] + (uint32_t)s_delta[lr_of_ck(r_ck[u])];   // (a hashed row's delta is final since the prologue)
            if constexpr (VALUES) vals[i] = r_v[u];
            cols[i] = r_ck[u];
        }
    if (!(SPADA_ABLATE & 4) && (NO != NBK || total > NO)) {   // (uniform) some products share their output with another one, or some are copied
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r_st[u] < BT_ST_HASHED_MAX) {
                if (!(r_st[u] & BT_ST_OWNER)) {
                    if constexpr (VALUES) atomicAdd(&vals[rank[u] + (uint32_t)s_delta[lr_of_ck(r_ck[u])]], r_v[u]);
                }
            } else if (r_st[u] == BT_H_COPY) {
                // COPY rows: C_i = a * B_k, already ascending: product number - first product of the row = place in the row
                const uint32_t p = ((uint32_t)u * BT_NWAVE + wave_u) * 64u + lane;
                const uint32_t i = (uint32_t)((int32_t)p + s_delta[lr_of_ck(r_ck[u])]);
                if constexpr (VALUES) vals[i] = r_v[u];
                cols[i] = r_ck[u];
            }
        }
    }
    BMARK(6);
    BSTOP(6, ooff ^ cols[tid]);
    unsigned long long base = 0;
    if constexpr (MODE == MODE_NUMERIC) base = ((unsigned long long)hdr[53] << 32) | hdr[52];   // (before the next task's prologue writes its own)
    const uint32_t my_ooff = ooff;

    // ---- position of the task's slice of C, then the stores --------------------------------------------------------------------
    // Where the next ticket is taken and the next task's prologue runs.  Without a chain (NUMERIC): here, under the stores.  With
    // it (FUSED): only after the stores -- a ticket taken while this task still waits for its position puts a task into the chain that
    // cannot publish its count before this wait is over, and every task behind it waits for that (measured: web 0.82 -> 0.95 ms,
    // R-MAT 16 4.9 -> 5.6 ms with the prologue under the wait; a wait of 2 - 12 thousand ticks before the early ticket: web 1.16 ... 0.81
    // against 0.77 ms; parking the finished task in registers or in a global staging slice instead of waiting: no gain either --
    // profiles/r04_experiments.txt)
    constexpr bool NEXT_EARLY = MODE == MODE_NUMERIC;
    if constexpr (NEXT_EARLY) next();   // (starts with a barrier: the task's outputs are complete in LDS)
    else __syncthreads();
    if constexpr (MODE != MODE_NUMERIC) {
        base = task_position<MODE>(g, t, total, hdr);
        if (range) {
            if (tid == 0) {
                if (td.first & 1u) g.cptr[rb] = base;   // first range of its row
                g.range_out[t] = base;
            }
        } else if ((uint32_t)tid < R) {
            g.cptr[rb + tid] = base + my_ooff;
        }
        if (t == ntasks - 1 && tid == 0) {
            g.cptr[g.nrows] = base + total;
            g.ctr->nnz_c = base + total;
        }
        if (base + total > g.capacity) {
            if (tid == 0) atomicOr(&g.ctr->cap_overflow, 1u);
            next();
            return;
        }
    }
    BMARK(7);
    // the slice as it is: neighbouring lanes, neighbouring addresses -- shifted so that every wave's 64 outputs START on a 128-byte
    // line of C.indices (and on a 256-byte boundary of C.data): a non-temporal store does not wait in the L2 for its neighbours, and a
    // piece that straddles one more line than it fills wrote 1.18 x the bytes of C (WRITE_SIZE, round 3)
    {
        // (by the ADDRESS of the slice in C.indices, not by its number: a caller's index buffer need not start on a line -- bench.py's
        // followed its value buffer at 48 bytes past one until round 6, and every wave's piece straddled two lines: 684 MB written
        // against 656 MB into aligned buffers)
        const uint32_t shift = (uint32_t)(((unsigned long long)(uintptr_t)g.c_idx >> 2) + base) & 31u;
        for (uint32_t j = tid; j < ((SPADA_ABLATE & 8) ? 0u : total + shift); j += BW) {
            if (j >= shift) {
                const uint32_t i = j - shift;
                __builtin_nontemporal_store(cols[i] & colmask, &g.c_idx[base + i]);
                __builtin_nontemporal_store(vals[i], &g.c_val[base + i]);
            }
        }
    }
    if constexpr (NEXT_EARLY) __syncthreads();   // (the outputs are read: the next task may clear the table)
    else next();                                 // (starts with a barrier)
    BMARK(8);
#undef BMARK
#undef BSTOP
}

}  // namespace spada
