// development aid (GPU box): does the time of the scatter's store pattern on a (column, value) pair of arrays depend on their RELATIVE address?
// One block holds both arrays; the value array starts X bytes behind the end of the column array, X swept.  Build: hipcc --offload-arch=gfx950 -O3 -o place_offset_probe place_offset_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <algorithm>
__global__ __launch_bounds__(256) void k_probe(uint32_t *__restrict__ col, double *__restrict__ val, unsigned long long nprod, uint32_t per_wave)
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
static float probe(uint32_t *col, double *val, unsigned long long np)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe, dim3(4096), dim3(256), 0, 0, col, val, np, 256u);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return best;
}
int main(int argc, char **argv)
{
    const unsigned long long np = argc > 1 ? strtoull(argv[1], nullptr, 10) : 800000000ull;   // products
    const size_t slack = 64ull << 20;
    // (a) separate allocations, several times
    for (int t = 0; t < 6; ++t) {
        uint32_t *c;
        double *v;
        if (hipMalloc(&c, np * 4) != hipSuccess || hipMalloc(&v, np * 8) != hipSuccess) return 1;
        printf("separate blocks %d: col %p val %p  probe %.3f ms\n", t, (void *)c, (void *)v, probe(c, v, np));
        uint32_t *c2;
        if (hipMalloc(&c2, np * 4) == hipSuccess) {
            printf("   other col block %p  probe %.3f ms\n", (void *)c2, probe(c2, v, np));
            hipFree(c2);
        }
        hipFree(c);
        hipFree(v);
    }
    // (a2) does a pair's time change when the arrays are overwritten front to back?
    for (int t = 0; t < 6; ++t) {
        uint32_t *c;
        double *v;
        if (hipMalloc(&c, np * 4) != hipSuccess || hipMalloc(&v, np * 8) != hipSuccess) return 1;
        const float p0 = probe(c, v, np);
        hipMemset(c, 0, np * 4);
        hipMemset(v, 0, np * 8);
        hipDeviceSynchronize();
        const float p1 = probe(c, v, np);
        hipMemset(c, 0xFF, np * 4);
        hipMemset(v, 0xFF, np * 8);
        hipDeviceSynchronize();
        const float p2 = probe(c, v, np);
        const float p3 = probe(c, v, np);
        printf("fresh pair %d: probe %.3f ms, after memset 0 %.3f, after memset 0xFF %.3f, again %.3f\n", t, p0, p1, p2, p3);
        hipFree(c);
        hipFree(v);
    }
    // (b) one block, the value array X bytes behind the column array's end
    char *blk;
    if (hipMalloc(&blk, np * 12 + slack) != hipSuccess) return 1;
    const size_t xs[] = {0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 4 << 20, (4 << 20) + 4096, 8 << 20, 16 << 20, 32 << 20, (32 << 20) + 65536, 48 << 20};
    for (size_t x : xs) {
        const size_t off = (np * 4 + 255) / 256 * 256 + x;
        printf("one block %p, value array at +%zu (+%zu behind the column array's end): probe %.3f ms\n", (void *)blk, off, x, probe((uint32_t *)blk, (double *)(blk + off), np));
    }
    hipFree(blk);
    return 0;
}
