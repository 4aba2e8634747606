// LDS operation throughput on gfx950: random-address reads, CAS, u32 add, f64 add (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int T = 8192;   // table entries
template <int OP, int ILP>
__global__ __launch_bounds__(1024) void k(int iters, unsigned *out, int spread)
{
    __shared__ unsigned keys[T];
    __shared__ double vals[T];
    for (int i = threadIdx.x; i < T; i += blockDim.x) { keys[i] = 0xFFFFFFFFu; vals[i] = 0.0; }
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned h[ILP];
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            x = x * 1664525u + 1013904223u;
            h[u] = spread ? ((x >> 12) & (T - 1)) : ((threadIdx.x * ILP + u + it) & (T - 1));
        }
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            if (OP == 0) acc += keys[h[u]];
            if (OP == 1) acc += atomicCAS(&keys[h[u]], 0xFFFFFFFFu, x);
            if (OP == 2) atomicAdd(&keys[h[u]], 1u);
            if (OP == 3) acc += atomicAdd(&keys[h[u]], 1u);
            if (OP == 4) atomicAdd(&vals[h[u]], 1.0);
            if (OP == 5) acc += (unsigned)vals[h[u]];
            if (OP == 6) atomicOr(&keys[h[u]], 1u << (x & 31));
        }
    }
    __syncthreads();
    if (acc == 0x12345678u) out[0] = acc + keys[threadIdx.x] + (unsigned)vals[threadIdx.x];
}
template <int OP, int ILP>
void run(const char *name, int block, int nblk_per_cu, int spread)
{
    unsigned *out; CK(hipMalloc(&out, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 2000, grid = 256 * nblk_per_cu;
    k<OP, ILP><<<grid, block>>>(10, out, spread);
    CK(hipEventRecord(a));
    k<OP, ILP><<<grid, block>>>(iters, out, spread);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double ops = (double)grid * block * iters * ILP;
    // assume 2.4 GHz
    printf("%-22s block %4d x%d/CU %s: %7.3f ms  %6.2f lane-ops/clk/CU (at 2.4 GHz)\n", name, block, nblk_per_cu, spread ? "random" : "linear",
           ms, ops / 256.0 / (ms * 1e-3 * 2.4e9));
    CK(hipFree(out));
}
int main()
{
    for (int spread = 0; spread < 2; ++spread)
        for (int blk : {256, 1024}) {
            int per = blk == 256 ? 4 : 1;
            run<0, 8>("ds_read_b32", blk, per, spread);
            run<5, 8>("ds_read_b64", blk, per, spread);
            run<1, 8>("ds_cmpst_rtn_b32", blk, per, spread);
            run<2, 8>("ds_add_u32 (noret)", blk, per, spread);
            run<3, 8>("ds_add_rtn_u32", blk, per, spread);
            run<4, 8>("ds_add_f64 (noret)", blk, per, spread);
            run<6, 8>("ds_or_b32 (noret)", blk, per, spread);
        }
    return 0;
}
