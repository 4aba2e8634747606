// gfx950 kernels, part 4: the BATCH task of the task kernel (included by spgemm_task.hip.hpp) -- consecutive non-BIG rows of C
// computed by one workgroup with one round trip per stage and the ordered emission done on BLOCKS of columns.
//
// What it replaces in the reference (citations into /root/reference/src): for the rows of one batch, the window fetch of A
// scalars (scheduler.rs:482-606, storage.rs:279-323), the B-fiber streaming (simulator.rs:892-953), the multiplier
// (simulator.rs:86-111), the sort by column (simulator.rs:143-171), the sum of equal columns (simulator.rs:199-230) and the
// result assembly (simulator.rs:1034-1062: columns ascending and unique, explicit zeros kept).
//
// Shape of a batch (k_cut1, batch_info): R <= 128 consecutive rows, E <= 512 A entries (contiguous: the entries of consecutive
// rows are), P <= 2048 products of which at most `limit` are hashed; the rest belong to COPY rows (one A entry: C_i = a * B_k is
// already ascending and needs no accumulator).  That shape is what makes the stages short:
//   prologue   the descriptor carries R, E, P and the first A entry: the entry loads (B-row begin / length, A value, row of the
//              entry) and the row records (one 16-byte RowRec per row) are issued together, the table is cleared under them
//   expand     ONE chunk: two entries per thread, one packed scan, head bits of the products (2048 bits), then two rounds of
//              four products per thread -- 64 consecutive products per wave and step (coalesced gathers from B).  The products
//              STAY IN REGISTERS (composite key, value, table slot): nothing is walked twice
//   accumulate the table is keyed by BLOCK -- (local row, column / 16) -- and holds a 16-bit mask of the columns seen in the block
//              (ds_cmpst on the key, ds_or on the mask).  The number of new mask bits is the task's output count: published
//              to the chain (one-pass mode) as soon as the expansion is done
//   order      only the BLOCKS are sorted (count / scan / scatter into monotone per-row buckets, rank inside the bucket): runs
//              of consecutive columns -- link lists into one site, mesh neighbourhoods -- collapse into a few blocks, so the
//              buckets stay short whatever the column pattern is (the rank loop over whole COLUMNS of round 2 waited for the
//              longest bucket of 64 lanes: 60 consecutive columns of one row fell into one bucket).  A prefix sum of the masks'
//              popcounts in block order gives every block its first output; the output of a product is
//              first(block) + popcount(mask below its bit): no search, no comparison
//   scale-add  the retained products add their value at that rank (ds_add_f64 into a dense array in OUTPUT order) and leave
//              their column there; simulator.rs:213-218 adds left to right, here the order is arbitrary (1e-9)
//   emit       once the task's position is known (the chain): the dense arrays are stored as they are (neighbouring lanes, neighbouring addresses), the
//              retained products of COPY rows go straight to their place
// The DIRECT RANGE tasks of BIG rows (columns [col_lo, col_hi] of one row whose products exceed a table; the row's entries are
// narrowed to the range by two binary searches each: B rows are ascending) run through the same stages whenever the row has at
// most BT_EMAX entries: one row, no COPY entries, the range as the row's column bounds.
// LDS (40 832 bytes, four workgroups per CU; regions are reused by the stages):
//   hdr 256 | MB u32[2048]: mask | first output << 16 | K 8 KB: keys, then (popcounts, slots) in block order | X 8 KB: entry
//   records, then bucket counters | Y 8 KB: head bits, then keys in bucket order | Z 4 KB: slots in bucket order | rows 3.2 KB;
//   the dense output arrays of the last stages lie over K + X (values) and Y (composite keys).
#pragma once

#ifndef SPADA_BT_STOP
#define SPADA_BT_STOP 0
#endif
#ifndef SPADA_DENSE_WIDE
#define SPADA_DENSE_WIDE 1
#endif
#ifndef SPADA_BT_FIRST_ROLLED
#define SPADA_BT_FIRST_ROLLED 1
#endif

namespace spada {

constexpr uint32_t BT_H_COPY = 0xFFFFu, BT_H_NONE = 0xFFFEu;  // `slot` of a retained product that is copied / of a lane without a product
constexpr size_t BT_OFF_MB = 256, BT_OFF_K = BT_OFF_MB + 8192, BT_OFF_X = BT_OFF_K + 8192, BT_OFF_Y = BT_OFF_X + 8192,
                 BT_OFF_Z = BT_OFF_Y + 8192, BT_OFF_ROWS = BT_OFF_Z + 4096;
struct BtRow {          // bucket parameters of a row (blocks): first bucket, buckets = blocks of the row, first block, buckets per block
    uint16_t boff, nb;
    uint32_t bmin;
    float scale;
};
__host__ __device__ constexpr size_t batch_lds()
{
    return BT_OFF_ROWS + (size_t)TK_RMAX * (sizeof(BtRow) + 4 + 4 + 1 + 4);
}
static_assert(batch_lds() <= 40960, "four workgroups per CU");
static_assert(TK_T == 2048 && BT_PMAX == 2048 && TK_BLOCK == 256, "the LDS map and the per-thread arrays are written for these sizes");

// exclusive scan of arr[0 .. N), N <= 2048 (u32 or u16 elements): every thread PER = ceil(N / 256) consecutive elements; the
// prefixes go to out[] (which may be arr itself).  Returns the total.  No barrier after the stores (the caller's next one covers them).
template <class T, uint32_t MAXPER>
__device__ inline uint32_t batch_scan_n(const T *arr, T *out, uint32_t N, uint32_t *slot)
{
    const int tid = threadIdx.x;
    const uint32_t PER = (N + 255u) >> 8;
    uint32_t loc[MAXPER], tot = 0;
#pragma unroll
    for (uint32_t j = 0; j < MAXPER; ++j) {
        const uint32_t idx = tid * PER + j;
        loc[j] = (j < PER && idx < N) ? (uint32_t)arr[idx] : 0u;
        tot += loc[j];
    }
    uint32_t total;
    uint32_t ex = block_scan_excl_dpp(tot, slot, &total);   // (its barrier comes after every thread has read its elements)
#pragma unroll
    for (uint32_t j = 0; j < MAXPER; ++j) {
        const uint32_t idx = tid * PER + j;
        if (j < PER && idx < N) {
            out[idx] = (T)ex;
            ex += loc[j];
        }
    }
    return total;
}
// (most tasks have fewer than 1024 blocks: half of the straight-line code is skipped for them)
template <class T>
__device__ inline uint32_t batch_scan(const T *arr, T *out, uint32_t N, uint32_t *slot)
{
#if SPADA_BT_FIRST_ROLLED
    if (N <= 1024u) return batch_scan_n<T, 4>(arr, out, N, slot);
#endif
    return batch_scan_n<T, 8>(arr, out, N, slot);
}

// DENSE (k_cut1 / the range's bounds decide): the blocks between the first and the last column of every hashed row, added up over
// the rows, are at most the table's slots.  Then the table needs no keys: block b of row r lives in slot soff[r] + b - bmin[r], the
// slots are in output order as they are, and the whole order stage is ONE prefix sum over the popcounts of the 2048 masks (meshes
// and banded matrices: half of the batches of the cop20k_A surrogate).
template <int MODE, bool DENSE, bool SPILL = false>
__device__ inline void batch_task(const TaskArgs &g, const TaskDesc &td, uint32_t t, uint32_t ntasks, unsigned char *smem,
                                  unsigned long long (&dbg_ph)[9])
{
    constexpr bool VALUES = MODE != MODE_COUNT;
    constexpr int BLOCK = TK_BLOCK, T = TK_T;
    uint32_t *hdr = (uint32_t *)smem;
    uint32_t *mb = (uint32_t *)(smem + BT_OFF_MB);
    uint32_t *keys = (uint32_t *)(smem + BT_OFF_K);
    uint16_t *pc = (uint16_t *)(smem + BT_OFF_K), *pcx = pc + T;   // outputs of the blocks in bucket order | outputs before them
    EntryRecNum *w_ent = (EntryRecNum *)(smem + BT_OFF_X);
    uint32_t *bcnt = (uint32_t *)(smem + BT_OFF_X);
    uint32_t *bm32 = (uint32_t *)(smem + BT_OFF_Y);                            // tail bits (last product of every entry): 64 words
    const unsigned long long *bm64 = (const unsigned long long *)(smem + BT_OFF_Y);
    uint32_t *lk = (uint32_t *)(smem + BT_OFF_Y);
    uint16_t *ls = (uint16_t *)(smem + BT_OFF_Z);
    double *vals = (double *)(smem + BT_OFF_K);                                 // K + X: 2048 values in output order
    uint32_t *cols = (uint32_t *)(smem + BT_OFF_Y);                             // composite keys in output order
    BtRow *s_emit = (BtRow *)(smem + BT_OFF_ROWS);
    int32_t *s_delta = (int32_t *)(s_emit + TK_RMAX);
    uint32_t *s_n = (uint32_t *)(s_delta + TK_RMAX);
    uint8_t *s_cls = (uint8_t *)(s_n + TK_RMAX);
    uint32_t *s_dense = (uint32_t *)(s_cls + TK_RMAX);   // DENSE: first slot of the row - its first block (mod 2^32)
    // scan slots (four words each) in the header; hdr[48 .. 50] belong to the chain and the ticket, hdr[52 .. 53] to the numeric base
    uint32_t *slot_rows = hdr + 4, *slot_ent = hdr + 8, *slot_cnt = hdr + 12, *slot_bk = hdr + 16, *slot_pc = hdr + 20, *slot_cr = hdr + 24,
             *slot_sp = hdr + 28;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane(wave);
    // stage boundaries: a comment in the ISA (static instruction counts per stage: scripts/dev/isa_stages.py) and, in SPADA_TASK_DBG
    // builds, the cycles of thread 0 per stage, summed per workgroup in `dbg_ph` (k_task adds them to TaskCounters::dbg at its end)
    unsigned long long ph_prev = SPADA_TASK_DBG ? __builtin_amdgcn_s_memtime() : 0;
#define BPH(i)                                                                  \
    do {                                                                        \
        if (MODE == MODE_NUMERIC && SPADA_BT_STOP == (i) + 1) return; /* development: numeric mode cut short here (counts per stage) */ \
        asm volatile("; BT_MARK " #i ::: "memory");                             \
        if (SPADA_TASK_DBG && tid == 0) {                                       \
            const unsigned long long n_ = __builtin_amdgcn_s_memtime();         \
            dbg_ph[i] += n_ - ph_prev;                                          \
            ph_prev = n_;                                                       \
        }                                                                       \
    } while (0)
    if (SPADA_TASK_DBG && tid == 0) dbg_ph[8] += 1;
#define BSTOP(id) do { if (MODE == MODE_NUMERIC && SPADA_BT_STOP == (id)) return; asm volatile("; BT_MARK s" #id ::: "memory"); } while (0)   /* development: finer cut points */
    // BATCH: rows, entries, products from batch_info | DIRECT RANGE: one row, its entries (descriptor), the products of the range
    // SPILLED RANGE (DENSE only): one row, no entries -- its products (column, scaled value) are a contiguous slice of the scratch arrays
    static_assert(!SPILL || DENSE, "spilled ranges take the dense path only");
    // a spilled range uses slots of 32 columns (the whole mask word; the slots' first outputs in an array of their own): twice the
    // column range fits the table.  DSH / DMASK: columns per slot of the DENSE layouts.
    constexpr bool WIDE = SPILL || (DENSE && SPADA_DENSE_WIDE != 0);   // (SPADA_DENSE_WIDE: every dense task, not only the spilled ranges)
    constexpr int DSH = WIDE ? BT_BSHIFT + 1 : BT_BSHIFT;
    constexpr uint32_t DMASK = WIDE ? 0xFFFFFFFFu : 0xFFFFu;
    uint16_t *fo = (uint16_t *)(smem + BT_OFF_Z);   // SPILL: first output of every slot (region Z: the dense paths do not sort)
    constexpr bool spill = SPILL;   // (td.kind == TASK_RANGE: the dispatch in k_task)
    const bool range = td.kind != TASK_BATCH;
    const uint32_t rb = td.row, R = range ? 1u : (td.np & 0xFFu), E = spill ? 0u : range ? (td.first >> 1) : ((td.np >> 8) & 0x3FFu),
                   PT = range ? td.np : ((td.np >> 18) & 0xFFFu);
    const uint64_t e0 = td.src;
    const uint32_t colbits = g.colbits;                       // >= BT_BSHIFT + 1 (the engine sees to it)
    const uint32_t colmask = colbits >= 32 ? 0xFFFFFFFFu : ((1u << colbits) - 1u);
    const uint32_t hshift = colbits >= 32 ? 0u : colbits - BT_BSHIFT;   // block key = composite key >> 4 = local row << hshift | block
    const uint32_t blkmask = colbits >= 32 ? 0xFFFFFFFFu : ((1u << hshift) - 1u);
    auto lr_of_ck = [&](uint32_t ck) { return colbits >= 32 ? 0u : ck >> colbits; };
    auto lr_of_hk = [&](uint32_t hk) { return colbits >= 32 ? 0u : hk >> hshift; };
    const bool wave_has_entries = wave_u * 128u < E;          // (two entries per thread)

    // ---- prologue: everything the walk needs in ONE round trip; the table is cleared while the loads are in flight ------------
    RowRec rr{0u, 0u, 0u, (uint32_t)CLS_EMPTY};
    if (range) {
        if (tid == 0) rr = RowRec{td.col_lo, td.col_hi, td.np, (uint32_t)CLS_SOLO};
    } else if ((uint32_t)tid < R) {
        rr = g.row_rec[rb + tid];
    }
    uint64_t b0[2] = {0, 0};
    uint32_t len[2] = {0, 0}, elr[2] = {0, 0};
    double av[2] = {0.0, 0.0};
    if (PT && wave_has_entries) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t ei = 2u * tid + i;
            if (ei < E) {
                const uint64_t q = e0 + ei;
                b0[i] = g.eb0[q];
                len[i] = g.elen[q];
                if (!range) elr[i] = g.arow[q] - (uint32_t)g.r0 - rb;
                if constexpr (VALUES) av[i] = g.aval[q];
            }
        }
    }
    if (MODE == MODE_NUMERIC && tid == 0) {
        const uint64_t c0 = range ? g.range_out[t] : g.cptr[rb];
        hdr[52] = (uint32_t)c0;
        hdr[53] = (uint32_t)(c0 >> 32);
    }
    {
        uint4 *k4 = (uint4 *)keys, *m4 = (uint4 *)mb;
#pragma unroll
        for (int s = 0; s < T / 4 / BLOCK; ++s) {
            if constexpr (!DENSE) k4[tid + s * BLOCK] = make_uint4(EMPTY_KEY, EMPTY_KEY, EMPTY_KEY, EMPTY_KEY);
            m4[tid + s * BLOCK] = make_uint4(0u, 0u, 0u, 0u);
        }
        if (tid < 64) bm32[tid] = 0u;
    }
    BSTOP(10);
    // rows: the hashed products before every row (the buckets of the block order are laid out in proportion to them, below) and
    // the outputs of COPY rows before it -- both known before anything is expanded
    const bool row_hashed = rr.cls == CLS_SMALL || rr.cls == CLS_SOLO;
    const uint32_t row_pr = row_hashed ? rr.nprod : 0u, row_cp = rr.cls == CLS_COPY ? rr.nprod : 0u;
    uint32_t row_tot;
    const uint32_t row_ex = block_scan_excl_dpp(row_pr | (row_cp << 16), slot_rows, &row_tot);   // (barrier: the table is cleared)
    const uint32_t boff = row_ex & 0xFFFFu, cpre = row_ex >> 16, NBK = row_tot & 0xFFFFu;         // buckets before the row, copied outputs before it
    uint32_t soff = 0, spanb = 0;   // DENSE: first slot of the row, its slots
    bool dense_ok = true;
    if constexpr (DENSE) {
        spanb = row_hashed && rr.nprod ? (rr.kmax >> DSH) - (rr.kmin >> DSH) + 1u : 0u;
        uint32_t sp_tot;
        soff = block_scan_excl_dpp(min(spanb, 2u * (uint32_t)T), slot_sp, &sp_tot);
        dense_ok = sp_tot <= (uint32_t)T;   // (the cut / the dispatch guarantee it)
        if ((uint32_t)tid < R) s_dense[tid] = soff - (rr.kmin >> DSH);
    }
    if ((uint32_t)tid < R) {
        s_cls[tid] = (uint8_t)rr.cls;
        s_n[tid] = 0u;
        // position of an output = position of the task + its number among the hashed outputs (or among the products) + delta
        s_delta[tid] = (int32_t)cpre;
    }
    __syncthreads();
    BPH(0);

    // ---- expand - scale - accumulate keys (scheduler.rs:482-606, simulator.rs:892-953, :86-111) ---------------------------
    uint32_t r_ck[8], r_h[8];   // the products of this thread: composite key (local row << colbits | column), table slot
    double r_v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r_ck[i] = 0u;
        r_h[i] = BT_H_NONE;
        r_v[i] = 0.0;
    }
    uint32_t mynew = 0, mykeys = 0;   // mask bits this lane has set | blocks (table keys) this lane has created
    uint32_t P = 0;
    if (PT) {
        bool ecopy[2] = {false, false};
        uint32_t mine = 0;
        if (range && wave_has_entries) {
            // DIRECT RANGE: every selected B row narrowed to [col_lo, col_hi]: l1 = first position with column >= lo, l2 = first with
            // column > hi, all searches of the wave in lock step (binary: K-ary searches were slower -- more scattered loads)
            const uint32_t lo = td.col_lo, hi = td.col_hi;
            const uint32_t *__restrict__ bidx = g.bidx;
            uint32_t l1[2], n1[2], l2[2], n2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                l1[i] = l2[i] = 0;
                n1[i] = n2[i] = len[i];
            }
            for (;;) {
                uint32_t any = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i) any |= n1[i] | n2[i];
                if (!any) break;
                uint32_t c1[2], c2[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    c1[i] = n1[i] ? bidx[b0[i] + l1[i] + (n1[i] >> 1)] : 0u;
                    c2[i] = n2[i] ? bidx[b0[i] + l2[i] + (n2[i] >> 1)] : 0u;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (n1[i]) {
                        const uint32_t hh = n1[i] >> 1;
                        if (c1[i] < lo) {
                            l1[i] += hh + 1;
                            n1[i] -= hh + 1;
                        } else {
                            n1[i] = hh;
                        }
                    }
                    if (n2[i]) {
                        const uint32_t hh = n2[i] >> 1;
                        if (c2[i] <= hi) {
                            l2[i] += hh + 1;
                            n2[i] -= hh + 1;
                        } else {
                            n2[i] = hh;
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                b0[i] += l1[i];
                len[i] = l2[i] - l1[i];
            }
        }
        if (wave_has_entries) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (len[i]) {
                    const uint8_t cls = s_cls[elr[i] < (uint32_t)TK_RMAX ? elr[i] : 0u];
                    ecopy[i] = cls == CLS_COPY;
                    if (!(ecopy[i] || cls == CLS_SMALL || cls == CLS_SOLO)) len[i] = 0u;   // (cannot happen: such rows have no products)
                }
#pragma unroll
            for (int i = 0; i < 2; ++i) mine += len[i] ? ((1u << 16) | len[i]) : 0u;   // entries with products << 16 | products
        }
        BSTOP(11);
        uint32_t tot32 = PT, ex32 = 0;   // (a spilled range: PT <= BT_PMAX products, no entries -- the dispatch in k_task sees to it)
        if (!spill) ex32 = block_scan_excl_dpp(mine, slot_ent, &tot32);
        P = tot32 & 0xFFFFu;
        const uint32_t nent = tot32 >> 16;   // entries with products
        BSTOP(12);
        if (P > BT_PMAX || !dense_ok) {   // the cut guarantees it; a batch that does not fit is an internal error, not a memory fault
            if (tid == 0 && atomicOr(&g.ctr->abort_flag, 32u) == 0u) {   // what did not fit (reported by the host)
                g.ctr->dbg[0] = td.kind | ((unsigned long long)E << 8) | ((unsigned long long)R << 32);
                g.ctr->dbg[1] = tot32;
                g.ctr->dbg[2] = td.np;
                g.ctr->dbg[3] = t;
            }
            P = 0;
        }
        if (P && wave_has_entries) {
            uint32_t ci = ex32 >> 16, po = ex32 & 0xFFFFu;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (len[i]) {
                    // record: (begin - first product) mod 2^48 | local row << 48 | copy << 55, A value
                    const uint64_t pack = ((b0[i] - po) & M48) | ((uint64_t)elr[i] << 48) | ((uint64_t)(ecopy[i] ? 1u : 0u) << 55);
                    w_ent[ci] = EntryRecNum{pack, av[i]};
                    atomicOr(&bm32[(po + len[i] - 1u) >> 5], 1u << ((po + len[i] - 1u) & 31));   // TAIL bit: the entry's last product
                    if (ecopy[i]) s_delta[elr[i]] -= (int32_t)po;   // (the row's one entry: product number - first product = place in the row)
                    ++ci;
                    po += len[i];
                }
        }
        BSTOP(13);
        if (!spill) __syncthreads();
        // tails before every 64-bit word of the bitmap (32 words): every wave scans them for itself and keeps the prefixes in the
        // lanes of one register (word w in lane w): the per-segment values are then scalar reads, and no second barrier is needed
        uint32_t tail_pre = 0;
        if (!spill) {
            const uint32_t c = (uint32_t)__popcll(bm64[lane & 31]);
            const uint32_t inc = wave_scan_incl_u32(lane < 32 ? c : 0u);
            tail_pre = inc - c;
        }
        BPH(1);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if ((uint32_t)r * 1024u < P) {   // (uniform)
                // lane l of a segment holds product seg + l, i.e. bit l of one bitmap word: the word and its prefix are wave-uniform
                // reads; the entry of a product = the entries that END before it = tails before the word + tails below the lane (a
                // v_mbcnt pair).  Lanes past the end take the last product (a valid address; only their atomics are switched off).
                uint32_t pp[4];
                bool act[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t p = (uint32_t)r * 1024u + ((uint32_t)u * 4u + wave_u) * 64u + lane;
                    act[u] = p < P;
                    pp[u] = min(p, P - 1u);
                }
                uint64_t pack[4] = {0, 0, 0, 0}, q[4];   // (a spilled range: local row 0, not copied)
                uint32_t col[4];
                if (spill) {
                    // the slice holds (column, a * b) of the range's products in any order: product p is element p
#pragma unroll
                    for (int u = 0; u < 4; ++u) q[u] = td.src + pp[u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) col[u] = g.scr_col[q[u]];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if constexpr (VALUES) r_v[r * 4 + u] = g.scr_val[q[u]];
                } else {
                    unsigned long long bits[4];
                    uint32_t bp[4], j[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t w = (uint32_t)r * 16u + (uint32_t)u * 4u + wave_u;   // = segment / 64
                        bits[u] = bm64[w];
                        bp[u] = (uint32_t)__builtin_amdgcn_readlane((int)tail_pre, (int)w);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(bits[u]));
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        // (a clamped lane sits past the last tail of its word: it counts every tail of the word, i.e. one entry too
                        // many whenever the last product is in this word -- min() with the last entry puts it back)
                        const uint32_t below =
                            __builtin_amdgcn_mbcnt_hi((uint32_t)(bits[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bits[u], 0u));
                        j[u] = min(bp[u] + below, nent - 1u);
                    }
                    double a_[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const EntryRecNum er = w_ent[j[u]];
                        pack[u] = er.pack;
                        a_[u] = er.av;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        asm volatile("" : "+v"(pack[u]));
                        if constexpr (VALUES) asm volatile("" : "+v"(a_[u]));
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) q[u] = ((pack[u] & M48) + pp[u]) & M48;
#pragma unroll
                    for (int u = 0; u < 4; ++u) col[u] = g.bidx[q[u]];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if constexpr (VALUES) r_v[r * 4 + u] = a_[u] * g.bval[q[u]];   // simulator.rs:100-101
                }
                uint32_t hk[4], h[4], old[4];
                bool hashed[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t lr = (uint32_t)(pack[u] >> 48) & 127u;
                    const bool copy = ((pack[u] >> 55) & 1ull) != 0;
                    const uint32_t ck = compose_key(lr, col[u], colbits);
                    r_ck[r * 4 + u] = ck;
                    hashed[u] = act[u] && !copy;
                    if (act[u] && copy) r_h[r * 4 + u] = BT_H_COPY;
                    hk[u] = ck >> BT_BSHIFT;
                    if constexpr (DENSE) {
                        // the slot is the block's place in the row: nothing to insert, nothing to probe (masked: a lane without a hashed
                        // product computes one from a stale record)
                        h[u] = (s_dense[R == 1 ? 0u : lr] + (col[u] >> DSH)) & (uint32_t)(T - 1);
                        old[u] = hk[u];
                    } else {
                        h[u] = hash_slot<TK_LOG_T>(hk[u]);
                        old[u] = hk[u];
                        if (hashed[u]) old[u] = atomicCAS(&keys[h[u]], EMPTY_KEY, hk[u]);
                    }
                }
                if constexpr (!DENSE) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (hashed[u] && old[u] != EMPTY_KEY && old[u] != hk[u]) {
                            const uint32_t step = probe_step(hk[u]);
                            for (;;) {
                                h[u] = (h[u] + step) & (T - 1);
                                old[u] = atomicCAS(&keys[h[u]], EMPTY_KEY, hk[u]);
                                if (old[u] == EMPTY_KEY || old[u] == hk[u]) break;
                            }
                        }
                        mykeys += (hashed[u] && old[u] == EMPTY_KEY) ? 1u : 0u;
                    }
                }
                if (r == 0) BSTOP(14);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (hashed[u]) {
                        const uint32_t bit = 1u << (col[u] & (DENSE ? (1u << DSH) - 1u : 15u));
                        const uint32_t was = atomicOr(&mb[h[u]], bit);
                        mynew += (was & bit) ? 0u : 1u;
                        r_h[r * 4 + u] = h[u];
                    }
            }
        }
        __syncthreads();
        BPH(2);
    }

    // ---- the count of the task: new mask bits + copied products (known from the rows); published as soon as it is known ----------
    uint32_t NO, NBt, total;   // hashed outputs, blocks, outputs of the task
    {
        uint32_t tot32;
        (void)block_scan_excl_dpp(mynew | (mykeys << 16), slot_cnt, &tot32);
        NO = tot32 & 0xFFFFu;
        NBt = tot32 >> 16;
        total = NO + (row_tot >> 16);   // + the products of the COPY rows
    }
    if constexpr (MODE != MODE_NUMERIC) task_publish<MODE>(g, t, total);
#if SPADA_PRIO
    if constexpr (MODE == MODE_FUSED) __builtin_amdgcn_s_setprio(0);
#endif
    if (SPADA_TASK_DBG && tid == 0) {   // shape of the tasks: products, hashed outputs, blocks, rows, entries
        atomicAdd(&g.ctr->dbg[0], (unsigned long long)P);
        atomicAdd(&g.ctr->dbg[1], (unsigned long long)NO);
        atomicAdd(&g.ctr->dbg[2], (unsigned long long)NBt);
        atomicAdd(&g.ctr->dbg[3], (unsigned long long)R);
        atomicAdd(&g.ctr->dbg[4], (unsigned long long)E);
        atomicAdd(&g.ctr->dbg[5], 1ull);
    }
    uint32_t myk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) myk[i] = (!DENSE && NO) ? keys[tid + i * BLOCK] : EMPTY_KEY;
    // DENSE: the slots ARE in (row, block) order: first output of a slot = outputs of the slots before it, one prefix sum over the
    // popcounts (eight consecutive slots per thread), written into the upper halves like the sorted path does
    auto dense_first_outputs = [&]() {
        uint4 *m4 = (uint4 *)mb;
        uint4 wa = m4[2 * tid], wb = m4[2 * tid + 1];
        uint32_t w[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w}, sum = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += (uint32_t)__popc(w[j] & DMASK);
        uint32_t tot;
        uint32_t ex = block_scan_excl_dpp(sum, slot_bk, &tot);
        uint32_t f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t c = (uint32_t)__popc(w[j] & DMASK);
            f[j] = ex;
            w[j] |= ex << 16;
            ex += c;
        }
        if constexpr (WIDE) {
            ((uint4 *)fo)[tid] = make_uint4(f[0] | (f[1] << 16), f[2] | (f[3] << 16), f[4] | (f[5] << 16), f[6] | (f[7] << 16));
        } else {
            m4[2 * tid] = make_uint4(w[0], w[1], w[2], w[3]);
            m4[2 * tid + 1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
        __syncthreads();
    };
    // hashed outputs before a slot
    auto dense_first_of = [&](uint32_t slot) { return slot < (uint32_t)T ? (WIDE ? (uint32_t)fo[slot] : mb[slot] >> 16) : NO; };

    BSTOP(15);
    if constexpr (MODE == MODE_COUNT) {
        if (range) return;   // (the count of a range task is all the position kernels need)
        // the symbolic phase wants the outputs of every ROW: popcounts of the masks summed per row (the other modes get them from
        // the scans of the block order); offsets of the rows inside the batch -- k_pos4 adds the position of the batch afterwards
        if constexpr (DENSE) {
            if (NO && R > 1) {
                dense_first_outputs();
                if ((uint32_t)tid < R && row_hashed) s_n[tid] = dense_first_of(soff + spanb) - dense_first_of(soff);
            } else if (tid == 0) {
                s_n[0] = NO;
            }
        } else if (NO) {
            if (R > 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (myk[i] != EMPTY_KEY) atomicAdd(&s_n[lr_of_hk(myk[i])], (uint32_t)__popc(mb[tid + i * BLOCK] & 0xFFFFu));
            } else if (tid == 0) {
                s_n[0] = NO;
            }
        }
        __syncthreads();
        const uint32_t n = (uint32_t)tid < R ? (row_hashed ? s_n[tid] : row_cp) : 0u;
        uint32_t tot32;
        const uint32_t ooff = block_scan_excl_dpp(n, slot_cr, &tot32);
        if ((uint32_t)tid < R) g.cptr[rb + tid] = ooff;
        return;
    }
    BPH(3);

    // ---- order: first output of every block = outputs of the blocks before it in (row, block) order ----------------------------
    // NBt buckets (one block per bucket on average), laid out over the rows in proportion to their products; a row that gets no
    // bucket of its own shares one with its neighbours (the keys decide inside a bucket).  The blocks are scattered into bucket
    // order together with their popcounts; one prefix sum in that order, then every block adds the popcounts of the smaller
    // keys of its own bucket: no exact rank, no second ordering pass.
    uint32_t hoff = 0;   // hashed outputs of the batch before this thread's row
    if (NO) {
      if constexpr (DENSE) {
        dense_first_outputs();
        if (R > 1 && (uint32_t)tid < R) hoff = dense_first_of(soff);
      } else {
        uint32_t b_lo = 0;
        if ((uint32_t)tid < R) {
            // (floor(x * f) is monotone in x, and the row behind starts where this one ends: the same expression of the same number)
            const float f = (float)NBt / (float)NBK;
            b_lo = min((uint32_t)((float)boff * f), NBt);
            const uint32_t b_hi = min((uint32_t)((float)(boff + row_pr) * f), NBt), cnt = max(b_hi - b_lo, 1u);
            const uint32_t bmin = rr.kmin >> BT_BSHIFT, bmax = rr.kmax >> BT_BSHIFT;
            s_emit[tid] = BtRow{(uint16_t)b_lo, (uint16_t)cnt, bmin, (float)cnt / ((float)(bmax - bmin) + 1.0f)};
        }
        for (uint32_t s2 = tid; s2 < NBt; s2 += BLOCK) bcnt[s2] = 0u;
        __syncthreads();
        auto bucket_of = [&](uint32_t k, const BtRow &e) {
            uint32_t bk = (uint32_t)((float)((k & blkmask) - e.bmin) * e.scale);
            bk = bk < e.nb ? bk : (uint32_t)e.nb - 1u;
            return e.boff + bk;
        };
        const bool one_row = R == 1;
        const BtRow e_one = s_emit[0];
        uint16_t myb[8];
        if (one_row) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                myb[i] = 0;
                if (myk[i] != EMPTY_KEY) {
                    myb[i] = (uint16_t)bucket_of(myk[i], e_one);
                    atomicAdd(&bcnt[myb[i]], 1u);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                myb[i] = 0;
                if (myk[i] != EMPTY_KEY) {
                    myb[i] = (uint16_t)bucket_of(myk[i], s_emit[lr_of_hk(myk[i])]);
                    atomicAdd(&bcnt[myb[i]], 1u);
                }
            }
        }
        BSTOP(16);
        __syncthreads();
        (void)batch_scan<uint32_t>(bcnt, bcnt, NBt, slot_bk);
        __syncthreads();
        BSTOP(17);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (myk[i] != EMPTY_KEY) {
                const uint32_t p = atomicAdd(&bcnt[myb[i]], 1u);   // afterwards bcnt[b] = end of bucket b
                lk[p] = myk[i];
                ls[p] = (uint16_t)(tid + i * BLOCK);
                pc[p] = (uint16_t)__popc(mb[tid + i * BLOCK] & 0xFFFFu);   // (region K: the table's keys are in registers by now)
            }
        BSTOP(18);
        __syncthreads();
        (void)batch_scan<uint16_t>(pc, pcx, NBt, slot_pc);
        __syncthreads();
        BSTOP(19);
        // outputs before key k: those of the buckets before its bucket + those of the smaller keys inside it.  The loads of all
        // the blocks of a thread are issued together (bucket bounds, then the prefix at the bucket's start); the walk over the bucket
        // is left to the blocks that share theirs (one block per bucket on average)
        auto smaller_in_bucket = [&](uint32_t k, uint32_t lo, uint32_t hi) {
            uint32_t add = 0;
#pragma clang loop unroll(disable) vectorize(disable)
            for (uint32_t jj = lo; jj < hi; ++jj) add += (lk[jj] < k) ? (uint32_t)pc[jj] : 0u;
            return add;
        };
#if SPADA_BT_FIRST_ROLLED
        // (a loop over pairs of blocks, as many as the task has: the straight-line version for 8 blocks per thread spent more on the
        // skeleton of the blocks a task does not have -- a third of the slots on the web input -- than it gained from issuing all loads together)
        const uint32_t nper = (NBt + (uint32_t)BLOCK - 1u) / (uint32_t)BLOCK;   // (uniform)
#pragma clang loop unroll(disable)
        for (uint32_t w0 = 0; w0 < nper; w0 += 2) {
            uint32_t kk[2], blo[2], bhi[2], fst[2];
            uint16_t slt[2];
            bool on[2];
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const uint32_t p = tid + (w0 + w) * BLOCK;
                on[w] = p < NBt;
                kk[w] = 0;
                slt[w] = 0;
                if (on[w]) {
                    kk[w] = lk[p];
                    slt[w] = ls[p];
                }
            }
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                blo[w] = bhi[w] = 0;
                if (on[w]) {
                    const uint32_t bk = one_row ? bucket_of(kk[w], e_one) : bucket_of(kk[w], s_emit[lr_of_hk(kk[w])]);
                    blo[w] = bk ? bcnt[bk - 1] : 0u;
                    bhi[w] = bcnt[bk];
                }
            }
#pragma unroll
            for (int w = 0; w < 2; ++w) fst[w] = on[w] ? (uint32_t)pcx[blo[w]] : 0u;
#pragma unroll
            for (int w = 0; w < 2; ++w)
                if (on[w]) {
                    if (bhi[w] - blo[w] > 1u) fst[w] += smaller_in_bucket(kk[w], blo[w], bhi[w]);
                    mb[slt[w]] |= fst[w] << 16;   // (one thread per slot)
                }
        }
#else
        {
            uint32_t kk[8], blo[8], bhi[8], fst[8];
            uint16_t slt[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const uint32_t p = tid + w * BLOCK;
                kk[w] = 0;
                slt[w] = 0;
                if (p < NBt) {
                    kk[w] = lk[p];
                    slt[w] = ls[p];
                }
            }
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                blo[w] = bhi[w] = 0;
                if (tid + w * BLOCK < NBt) {
                    const uint32_t bk = one_row ? bucket_of(kk[w], e_one) : bucket_of(kk[w], s_emit[lr_of_hk(kk[w])]);
                    blo[w] = bk ? bcnt[bk - 1] : 0u;
                    bhi[w] = bcnt[bk];
                }
            }
#pragma unroll
            for (int w = 0; w < 8; ++w) fst[w] = (tid + w * BLOCK < NBt) ? (uint32_t)pcx[blo[w]] : 0u;
#pragma unroll
            for (int w = 0; w < 8; ++w)
                if (tid + w * BLOCK < NBt) {
                    if (bhi[w] - blo[w] > 1u) fst[w] += smaller_in_bucket(kk[w], blo[w], bhi[w]);
                    mb[slt[w]] |= fst[w] << 16;   // (one thread per slot)
                }
        }
#endif
        // the rows' first hashed outputs: the outputs before the (virtual) smallest key of the row
        if (!one_row && (uint32_t)tid < R) {
            if (b_lo < NBt) {
                const uint32_t lo = b_lo ? bcnt[b_lo - 1] : 0u, hi = bcnt[b_lo];
                hoff = (lo < NBt ? (uint32_t)pcx[lo] : NO) + smaller_in_bucket((uint32_t)tid << hshift, lo, hi);
            } else {
                hoff = NO;
            }
        }
      }
        __syncthreads();   // pc / pcx (region K) are read: the values may take their place
        BPH(4);
        // ---- scale - add: every retained product adds its value at its output (simulator.rs:213-218; order differs) -----------
        for (uint32_t s = tid; s < NO; s += BLOCK) vals[s] = 0.0;
        __syncthreads();
        BSTOP(20);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (r_h[i] < BT_H_NONE) {
                const uint32_t w = mb[r_h[i]];
                const uint32_t rank = WIDE ? (uint32_t)fo[r_h[i]] + (uint32_t)__popc(w & ((1u << (r_ck[i] & 31u)) - 1u))
                                            : (w >> 16) + (uint32_t)__popc(w & ((1u << (r_ck[i] & 15u)) - 1u));
                atomicAdd(&vals[rank], r_v[i]);
                cols[rank] = r_ck[i];
            }
    }
    // the rows' first outputs inside the batch: hashed outputs before the row + copied outputs before it
    const uint32_t ooff = hoff + cpre;
    if ((uint32_t)tid < R && rr.cls == CLS_COPY) s_delta[tid] += (int32_t)hoff;
    __syncthreads();
    BPH(5);

    // ---- position of the task's slice of C (the chain), then the stores -------------------------------------------------------
    unsigned long long base;
    if constexpr (MODE == MODE_NUMERIC) {
        base = ((unsigned long long)hdr[53] << 32) | hdr[52];
    } else {
        base = task_position<MODE>(g, t, total, hdr);
        if (range) {
            if (tid == 0) {
                if (td.first & 1u) g.cptr[rb] = base;   // first range of its row
                g.range_out[t] = base;
            }
        } else if ((uint32_t)tid < R) {
            g.cptr[rb + tid] = base + ooff;
        }
        if (t == ntasks - 1 && tid == 0) {
            g.cptr[g.nrows] = base + total;
            g.ctr->nnz_c = base + total;
        }
        if (base + total > g.capacity) {
            if (tid == 0) atomicOr(&g.ctr->cap_overflow, 1u);
            return;
        }
    }
    BPH(6);
    for (uint32_t p = tid; p < NO; p += BLOCK) {
        const uint32_t ck = cols[p];
        const unsigned long long pos = base + p + (long long)s_delta[lr_of_ck(ck)];
#if SPADA_NT_STORE
        __builtin_nontemporal_store(ck & colmask, &g.c_idx[pos]);
        __builtin_nontemporal_store(vals[p], &g.c_val[pos]);
#else
        g.c_idx[pos] = ck & colmask;
        g.c_val[pos] = vals[p];
#endif
    }
    BSTOP(21);
    if (total > NO) {   // COPY rows: C_i = a * B_k, already ascending: product number - first product of the row = place in the row
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (r_h[i] == BT_H_COPY) {
                const uint32_t p = (uint32_t)(i >> 2) * 1024u + ((uint32_t)(i & 3) * 4u + wave_u) * 64u + lane;
                const unsigned long long pos = base + (long long)((int32_t)p + s_delta[lr_of_ck(r_ck[i])]);
#if SPADA_NT_STORE
                __builtin_nontemporal_store(r_ck[i] & colmask, &g.c_idx[pos]);
                __builtin_nontemporal_store(r_v[i], &g.c_val[pos]);
#else
                g.c_idx[pos] = r_ck[i] & colmask;
                g.c_val[pos] = r_v[i];
#endif
            }
    }
    BPH(7);
#undef BPH
#undef BSTOP
}

}  // namespace spada
