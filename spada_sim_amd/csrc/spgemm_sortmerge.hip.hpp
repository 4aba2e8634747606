// gfx950 kernels of the task pipeline, part 5 of 5: the task kernel with the SORT-MERGE accumulator (included by spgemm_task.hip.hpp).
#pragma once

namespace spada {

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
