// f64 / f32 VALU issue rates on gfx950 (one wave64 instruction = how many cycles?): N independent accumulator chains per thread
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
template <typename T, int OP>
__global__ __launch_bounds__(256) void k(T *out, T a, T b, int iters)
{
    T x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { x0 += b; x1 += b; x2 += b; x3 += b; x4 += b; x5 += b; x6 += b; x7 += b; }
        if (OP == 1) { x0 *= b; x1 *= b; x2 *= b; x3 *= b; x4 *= b; x5 *= b; x6 *= b; x7 *= b; }
        if (OP == 2) { x0 = fma(x0, b, a); x1 = fma(x1, b, a); x2 = fma(x2, b, a); x3 = fma(x3, b, a); x4 = fma(x4, b, a); x5 = fma(x5, b, a); x6 = fma(x6, b, a); x7 = fma(x7, b, a); }
        if (OP == 3) { x0 = fmax(x0 + b, a); x1 = fmax(x1 + b, a); x2 = fmax(x2 + b, a); x3 = fmax(x3 + b, a); x4 = fmax(x4 + b, a); x5 = fmax(x5 + b, a); x6 = fmax(x6 + b, a); x7 = fmax(x7 + b, a); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <typename T, int OP> void run(const char *name, int ops_per_iter)
{
    T *d; hipMalloc(&d, sizeof(T) * 256 * 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4096, blocks = 256 * 16;      // 16 workgroups of 4 waves per CU
    k<T, OP><<<blocks, 256>>>(d, (T)1.0000001, (T)0.9999999, 16);
    hipEventRecord(a);
    k<T, OP><<<blocks, 256>>>(d, (T)1.0000001, (T)0.9999999, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_instr = (double)blocks * 4 * iters * ops_per_iter;          // per launch
    const double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;
    printf("%-14s %.3f ms: %.2f cycles per wave64 instruction (at 2.4 GHz, 1024 SIMDs)\n", name, ms, simd_cycles / wave_instr);
    hipFree(d);
}
int main()
{
    run<float, 0>("v_add_f32", 8); run<float, 2>("v_fma_f32", 8);
    run<double, 0>("v_add_f64", 8); run<double, 1>("v_mul_f64", 8); run<double, 2>("v_fma_f64", 8); run<double, 3>("add+max f64", 16);
    return 0;
}
