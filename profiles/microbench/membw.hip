// HBM streaming micro-benchmark (calibration for the roofline fractions in DESIGN.md):
// read+write copy at 1 / 4 / 16 bytes per lane, and read-only reduction, over a 1.5 GB buffer.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <typename T> __global__ void copy_k(const T *__restrict__ a, T *__restrict__ b, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}
template <typename T> __global__ void read_k(const T *__restrict__ a, uint32_t *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += st) { T v = a[i]; const uint32_t *p = (const uint32_t *)&v; for (unsigned k = 0; k < sizeof(T) / 4; k++) acc += p[k]; }
    if (acc == 0x12345678) out[0] = acc;
}
__global__ void read_u8_k(const uint8_t *__restrict__ a, uint32_t *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += st) acc += a[i];
    if (acc == 0x12345678) out[0] = acc;
}
template <typename F> float timeit(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main()
{
    const size_t bytes = (size_t)1536 << 20;
    uint8_t *a, *b; uint32_t *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 64);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    for (int blocks : {2048, 8192, 65536}) {
        float ms;
        ms = timeit([&] { hipLaunchKernelGGL(copy_k<uint4>, dim3(blocks), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, bytes / 16); }, 5);
        printf("blocks %6d copy 16B/lane: %7.1f GB/s (r+w)\n", blocks, 2.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(copy_k<uint32_t>, dim3(blocks), dim3(256), 0, 0, (const uint32_t *)a, (uint32_t *)b, bytes / 4); }, 5);
        printf("blocks %6d copy  4B/lane: %7.1f GB/s (r+w)\n", blocks, 2.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(copy_k<uint8_t>, dim3(blocks), dim3(256), 0, 0, (const uint8_t *)a, (uint8_t *)b, bytes / 4); }, 3);
        printf("blocks %6d copy  1B/lane: %7.1f GB/s (r+w, quarter buffer)\n", blocks, 2.0 * (bytes / 4) / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(read_k<uint4>, dim3(blocks), dim3(256), 0, 0, (const uint4 *)a, o, bytes / 16); }, 5);
        printf("blocks %6d read 16B/lane: %7.1f GB/s\n", blocks, 1.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(read_k<uint32_t>, dim3(blocks), dim3(256), 0, 0, (const uint32_t *)a, o, bytes / 4); }, 5);
        printf("blocks %6d read  4B/lane: %7.1f GB/s\n", blocks, 1.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(read_u8_k, dim3(blocks), dim3(256), 0, 0, a, o, bytes / 4); }, 3);
        printf("blocks %6d read  1B/lane: %7.1f GB/s (quarter buffer)\n", blocks, 1.0 * (bytes / 4) / ms / 1e6);
    }
    return 0;
}
