// micro-benchmark: what a dependent float64 addition costs a lone wavefront on gfx950, with and without the LDS traffic of
// rt_integral_kernel's row wave.  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off chain.hip -o chain && ./chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__global__ void k_regs(double *out, unsigned long long *ticks, double seed)
{
    double c = seed, x = seed * 0.5;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; i++) c = __dadd_rn(c, x);
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = c;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
template <int LANES, int AHEAD>
__global__ void k_lds(double *out, unsigned long long *ticks, double seed)
{
    __shared__ __align__(16) double tile[16][66 * 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < 16 * 264; i += 64) (&tile[0][0])[i] = seed + i;
    __syncthreads();
    double c = seed;
    unsigned long long t0 = 0, t1 = 0;
    if (lane < LANES) {
        double2 *row = reinterpret_cast<double2 *>(tile[lane & 15]);
        t0 = __builtin_readcyclecounter();
        for (int rep = 0; rep < N / 256; rep++) {
            double2 x[AHEAD][4];
#pragma unroll
            for (int u = 0; u < AHEAD; u++)
#pragma unroll
                for (int q = 0; q < 4; q++) x[u][q] = row[4 * u + q];
            for (int j = 0; j < 128; j += 4 * AHEAD) {
#pragma unroll
                for (int u = 0; u < AHEAD; u++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) { c = __dadd_rn(c, x[u][q].x); x[u][q].x = c; c = __dadd_rn(c, x[u][q].y); x[u][q].y = c; }
#pragma unroll
                    for (int q = 0; q < 4; q++) row[j + 4 * u + q] = x[u][q];
                    if (j + 4 * (AHEAD + u) < 128) {
#pragma unroll
                        for (int q = 0; q < 4; q++) x[u][q] = row[j + 4 * (AHEAD + u) + q];
                    }
                }
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    out[threadIdx.x] = c;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
int main()
{
    double *out; unsigned long long *tk, h;
    hipMalloc(&out, 64 * 8); hipMalloc(&tk, 8);
    auto run = [&](const char *name, auto launch) {
        for (int r = 0; r < 3; r++) { launch(); hipDeviceSynchronize(); }
        hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
        printf("%-46s %8llu clock ticks for %d additions = %.2f per addition\n", name, h, N, (double)h / N);
    };
    run("registers only, 64 lanes", [&] { hipLaunchKernelGGL(k_regs, dim3(1), dim3(64), 0, 0, out, tk, 1.0); });
    run("LDS b128 read/write, 16 lanes, 1 batch ahead", [&] { hipLaunchKernelGGL((k_lds<16, 1>), dim3(1), dim3(64), 0, 0, out, tk, 1.0); });
    run("LDS b128 read/write, 16 lanes, 4 batches ahead", [&] { hipLaunchKernelGGL((k_lds<16, 4>), dim3(1), dim3(64), 0, 0, out, tk, 1.0); });
    run("LDS b128 read/write, 64 lanes, 4 batches ahead", [&] { hipLaunchKernelGGL((k_lds<64, 4>), dim3(1), dim3(64), 0, 0, out, tk, 1.0); });
    // the counter's unit: s_memtime / readcyclecounter against wall time
    return 0;
}
