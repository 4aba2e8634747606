// gfx950 MEASUREMENT kernels (not part of the product path; include/spada_probe.h): what the expansion of the products ALONE costs on
// this part -- the floor under any accumulator (profiles/r06_floor.txt).
//
// k_floor runs over the task list a pipeline run has left (batch tasks, direct range tasks with a cut table, single-pass spilled
// ranges; the tasks of the older range path are skipped) with everything the task kernel's latency consists of taken away:
//   * STATIC assignment -- workgroup b takes tasks b, b + G, b + 2 G, ...: no ticket, no chain, no publication
//   * the descriptor of task i + 2 and the entry loads of task i + 1 are in flight while task i's products are gathered: no
//     dependent round trip in front of a task except the B gathers themselves
//   * expand + scale only (scheduler.rs:482-606 window fetch, simulator.rs:892-953 B-fiber streaming, simulator.rs:86-111
//     multiplier): WRITE = 0 folds the products into a word nobody reads, WRITE = 1 stores 12 bytes per product (column, a * b) at
//     task * 2048 + product number -- contiguous, line-aligned runs, the best case of any output layout
// Workgroups of 512 threads, four products per thread, eight waves per SIMD: the task kernel's shape.
#pragma once

namespace spada {

constexpr size_t FLOOR_LDS = 256 + 2 * ((size_t)BT_EMAX * 16 + 256);

template <int WRITE>
__global__ __launch_bounds__(TKW, 8) void k_floor(const TaskArgs *__restrict__ gp_, uint32_t *__restrict__ out_idx,
                                                  double *__restrict__ out_val, unsigned long long *__restrict__ sink)
{
    TaskArgsC &g = *(TaskArgsC *)gp_;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hdr = (uint32_t *)smem;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t ntasks = g.ctr->ntasks, G = gridDim.x;
    uint32_t t = blockIdx.x;
    if (t >= ntasks) return;
    // what a task needs before its products: descriptor (scalar), then one entry per thread
    struct Ent {
        uint64_t b0;
        uint32_t len;
        double av;
    };
    auto usable = [](const TaskDesc &d) { return task_is_batch(d); };
    auto entries_of = [](const TaskDesc &d) -> uint32_t {
        if (task_spill_batch(d)) return 0u;
        return d.kind == TASK_BATCH ? ((d.np >> 8) & 0x3FFu) : (d.first >> 1);
    };
    auto issue_entries = [&](const TaskDesc &d) -> Ent {
        Ent e{0ull, 0u, 0.0};
        if (!usable(d)) return e;
        const uint32_t E = entries_of(d);
        if (tid < E) {
            e.b0 = (g.eb0 + d.src)[tid];
            e.len = (g.elen + d.src)[tid];
            e.av = (g.aval + d.src)[tid];
            if (d.kind == TASK_RANGE_DIRECT && g.cuts && d.cut != ~0ull) {
                const uint32_t c_lo = (g.cuts + d.cut)[tid], c_hi = (g.cuts + d.cut + E)[tid];
                e.b0 += c_lo;
                e.len = c_hi - c_lo;
            }
        }
        return e;
    };
    TaskDesc td = load_task(g.tasks, t);
    TaskDesc td1 = t + G < ntasks ? load_task(g.tasks, t + G) : TaskDesc{};
    Ent e = issue_entries(td);
    unsigned long long acc = 0;
    uint32_t it = 0;
    for (; t < ntasks; t += G, ++it) {
        unsigned char *buf = smem + 256 + (size_t)(it & 1u) * ((size_t)BT_EMAX * 16 + 256);
        EntryRecNum *w_ent = (EntryRecNum *)buf;
        uint32_t *bm32 = (uint32_t *)(buf + (size_t)BT_EMAX * 16);
        const unsigned long long *bm64 = (const unsigned long long *)bm32;
        const bool ok = usable(td), spill = task_spill_batch(td);
        const uint32_t E = entries_of(td);
        // entries numbered densely, and their products (the entry loads of this task were issued an iteration ago)
        if (tid < 64u) bm32[tid] = 0u;
        const uint32_t ent_v = e.len ? ((1u << 16) | min(e.len, 0xFFFFu)) : 0u;
        uint32_t *slot = hdr + (it & 1u) * 8u;
        const uint32_t inc = scan_part(ent_v, slot, tid);
        __syncthreads();
        uint32_t tot32;
        const uint32_t ex32 = scan_done<BT_NWAVE>(inc, ent_v, slot, &tot32, tid);
        const uint32_t P = ok ? (spill ? td.np : (tot32 & 0xFFFFu)) : 0u, nent = tot32 >> 16;
        if (ok && !spill && e.len && P <= BT_PMAX) {
            const uint32_t ci = ex32 >> 16, po = ex32 & 0xFFFFu;
            w_ent[ci] = EntryRecNum{(e.b0 - po) & M48, e.av};
            atomicOr(&bm32[(po + e.len - 1u) >> 5], 1u << ((po + e.len - 1u) & 31u));
        }
        (void)E;
        // the NEXT task's entries and the descriptor of the one after it: in flight under this task's gathers
        const TaskDesc td2 = t + 2u * G < ntasks ? load_task(g.tasks, t + 2u * G) : TaskDesc{};
        e = t + G < ntasks ? issue_entries(td1) : Ent{0ull, 0u, 0.0};
        __syncthreads();
        if (P && P <= BT_PMAX) {
            uint32_t col[4];
            double v[4];
            uint32_t pp[4];
            bool act[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t p = ((uint32_t)u * BT_NWAVE + wave_u) * 64u + lane;
                act[u] = p < P;
                pp[u] = min(p, P - 1u);
            }
            if (spill) {
#pragma unroll
                for (int u = 0; u < 4; ++u) col[u] = g.scr_col[td.src + pp[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = g.scr_val[td.src + pp[u]];
            } else {
                const uint32_t c = (uint32_t)__popcll(bm64[lane & 31u]);
                const uint32_t tinc = wave_scan_incl_u32(lane < 32u ? c : 0u);
                const uint32_t tail_pre = tinc - c;
                uint64_t q[4];
                double a_[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t w = (uint32_t)u * BT_NWAVE + wave_u;
                    const unsigned long long bits = bm64[w];
                    const uint32_t bp = (uint32_t)__builtin_amdgcn_readlane((int)tail_pre, (int)w);
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bits, 0u));
                    const uint32_t j = min(bp + below, nent - 1u);
                    const EntryRecNum er = w_ent[j];
                    q[u] = ((er.pack & M48) + pp[u]) & M48;
                    a_[u] = er.av;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) col[u] = g.bidx[q[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = a_[u] * g.bval[q[u]];   // simulator.rs:100-101
            }
            if constexpr (WRITE) {
                const size_t base = (size_t)t * BT_PMAX;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (act[u]) {
                        const uint32_t p = ((uint32_t)u * BT_NWAVE + wave_u) * 64u + lane;
                        __builtin_nontemporal_store(col[u], &out_idx[base + p]);
                        __builtin_nontemporal_store(v[u], &out_val[base + p]);
                    }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (act[u]) acc += (unsigned long long)col[u] ^ (unsigned long long)__double_as_longlong(v[u]);
            }
        }
        td = td1;
        td1 = td2;
    }
    if (acc == 0x9E3779B97F4A7C15ull) sink[0] = acc;   // (keeps the products alive; never true in practice)
}

}  // namespace spada
