// development aid: latency of dependent atomic operations at agent scope and at workgroup scope (performed in the XCD's L2),
// and visibility of workgroup-scope atomics between workgroups of the same / of different XCDs.   hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) ; }

template <int SCOPE, int OP>
__global__ void k_lat(unsigned long long *w, int iters, unsigned long long *out)
{
    if (threadIdx.x != 0) return;
    unsigned long long idx = 0, acc = 0;
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        unsigned long long v;
        if (OP == 0) v = __hip_atomic_load(&w[idx * 16], __ATOMIC_RELAXED, SCOPE);
        else v = __hip_atomic_fetch_add(&w[idx * 16], 0ull, __ATOMIC_RELAXED, SCOPE);
        idx = (idx + 1 + v) & 1023;   // dependent
        acc += v;
    }
    const long long t1 = wall_clock64();
    out[0] = (unsigned long long)(t1 - t0);
    out[1] = acc;
    out[2] = xcc_id();
}

// ping-pong between two workgroups through one word: each side waits for the other's value, then bumps it
template <int SCOPE>
__global__ void k_pingpong(unsigned long long *w, int rounds, int other_block, unsigned long long *out)
{
    if (blockIdx.x != 0 && (int)blockIdx.x != other_block) return;
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x == 0 ? 0 : 1;
    const long long t0 = wall_clock64();
    long long spins = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned long long want = 2ull * r + me;
        while (__hip_atomic_load(w, __ATOMIC_RELAXED, SCOPE) != want) { if (++spins > 50000000) { out[4 + me] = 1; return; } }
        __hip_atomic_store(w, want + 1, __ATOMIC_RELAXED, SCOPE);
    }
    const long long t1 = wall_clock64();
    out[me * 2] = (unsigned long long)(t1 - t0);
    out[me * 2 + 1] = xcc_id();
}

int main()
{
    unsigned long long *w, *out;
    CHECK(hipMalloc(&w, 1024 * 16 * 8));
    CHECK(hipMemset(w, 0, 1024 * 16 * 8));
    CHECK(hipMalloc(&out, 64));
    unsigned long long h[8];
    const int iters = 20000;
    const double tick_ns = 10.0;   // wall_clock64: 100 MHz
    auto report = [&](const char *what) {
        hipDeviceSynchronize();
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("%-46s %8.1f ns per op  (xcc %llu)\n", what, h[0] * tick_ns / iters, h[2]);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_lat<__HIP_MEMORY_SCOPE_AGENT, 0>), dim3(1), dim3(64), 0, 0, w, iters, out); report("atomic load, agent scope");
        hipLaunchKernelGGL((k_lat<__HIP_MEMORY_SCOPE_WORKGROUP, 0>), dim3(1), dim3(64), 0, 0, w, iters, out); report("atomic load, workgroup scope");
        hipLaunchKernelGGL((k_lat<__HIP_MEMORY_SCOPE_AGENT, 1>), dim3(1), dim3(64), 0, 0, w, iters, out); report("atomic fetch_add, agent scope");
        hipLaunchKernelGGL((k_lat<__HIP_MEMORY_SCOPE_WORKGROUP, 1>), dim3(1), dim3(64), 0, 0, w, iters, out); report("atomic fetch_add, workgroup scope");
    }
    const int rounds = 2000;
    for (int other : {8, 16, 1, 2, 7}) {
        for (int sc = 0; sc < 2; ++sc) {
            CHECK(hipMemset(w, 0, 64));
            CHECK(hipMemset(out, 0, 64));
            if (sc == 0) hipLaunchKernelGGL((k_pingpong<__HIP_MEMORY_SCOPE_AGENT>), dim3(32), dim3(64), 0, 0, w, rounds, other, out);
            else hipLaunchKernelGGL((k_pingpong<__HIP_MEMORY_SCOPE_WORKGROUP>), dim3(32), dim3(64), 0, 0, w, rounds, other, out);
            hipDeviceSynchronize();
            hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
            printf("ping-pong block 0 (xcc %llu) <-> block %d (xcc %llu), %s scope: %8.1f ns per hand-over%s\n", h[1], other, h[3],
                   sc == 0 ? "agent" : "workgroup", h[0] * tick_ns / (2.0 * rounds), (h[4] || h[5]) ? "  ** TIMED OUT: not visible **" : "");
        }
    }
    return 0;
}
