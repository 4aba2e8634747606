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
//
// The pipeline is split over five headers (round 6), included here in order:
//   spgemm_defs.hip.hpp       constants and records: task shapes, TaskDesc, TaskCounters, RowRec, row classes, modes, TaskArgs
//   spgemm_bigrow.hip.hpp     the BIG-row stage: parts, histograms, plan, cut table, scatter
//   spgemm_prekernel.hip.hpp  entry descriptors + row statistics, row classes + cut, the one launch behind the plan, positions
//   (this file)               the chain, the older range path, k_task / k_task_range; spgemm_batch.hip.hpp: the batch task
//   spgemm_sortmerge.hip.hpp  k_task_sm
#pragma once
#include "spgemm_prekernel.hip.hpp"

namespace spada {

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
                    if (__ballot(gave_up)) break;   // (the run is given up: any position will do)
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
    if (tid == 0) hdr[50] = g.task_lo + atomicAdd(my_ticket, 1u) * TK_NQ + task_queue();
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
        // (behind a wait that gave itself up -- chain_gave_up -- the workgroup's remaining tasks each give up at their first look at the flag: no
        // check here, where an LDS read per task would make the loop wait for the next task's scalar loads)
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


}  // namespace spada
#include "spgemm_sortmerge.hip.hpp"
