// micro-benchmark: LDS instruction throughput of one CU on gfx950 - W wavefronts of one workgroup each issue N independent LDS reads of
// one kind (byte / dword / 2 x dword / 8 bytes / 16 bytes; lane addresses conflict-free or random bytes), cycles per instruction and CU.
// hipcc -O3 --offload-arch=gfx950 ldsrate.hip -o ldsrate && ./ldsrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 2048
template <int KIND>
__global__ void k(unsigned long long *ticks, uint32_t *sink, int seed)
{
    __shared__ __align__(16) uint32_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 2654435761u + seed;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    // byte offsets: KIND 0 random bytes (as the taps of the integral kernel), others lane-linear
    uint32_t off = KIND == 0 ? ((lane * 2654435761u + seed) >> 19) & 0x1fff : (KIND == 4 ? lane * 16 : (KIND == 3 ? lane * 8 : lane * 4));
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; i++) {
        const uint32_t o = (off + i * 64) & 0x7fc0 | (off & 63);
        const uint8_t *p = reinterpret_cast<const uint8_t *>(lds) + (KIND == 0 ? ((off + i * 37) & 0x7fff) : o);
        if (KIND == 0) acc += *p;
        else if (KIND == 1) acc += *reinterpret_cast<const uint32_t *>(p);
        else if (KIND == 2) { acc += reinterpret_cast<const uint32_t *>(p)[0] + reinterpret_cast<const uint32_t *>(p)[33]; }
        else if (KIND == 3) { const uint2 v = *reinterpret_cast<const uint2 *>(p); acc += v.x + v.y; }
        else { const uint4 v = *reinterpret_cast<const uint4 *>(p); acc += v.x + v.y + v.z + v.w; }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    sink[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
int main()
{
    unsigned long long *tk, h; uint32_t *sink;
    hipMalloc(&tk, 8); hipMalloc(&sink, 1024 * 4);
    const char *names[5] = {"ds_read_u8, random bytes", "ds_read_b32, lane-linear", "ds_read2_b32 (2 dwords)", "ds_read_b64", "ds_read_b128"};
    for (int w = 1; w <= 16; w *= 2) {
        for (int kd = 0; kd < 5; kd++) {
            for (int r = 0; r < 3; r++) {
                switch (kd) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * w), 0, 0, tk, sink, r); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * w), 0, 0, tk, sink, r); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * w), 0, 0, tk, sink, r); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64 * w), 0, 0, tk, sink, r); break;
                default: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64 * w), 0, 0, tk, sink, r); break;
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
            printf("%2d waves  %-26s %7.2f cycles per instruction and wave, %6.2f per instruction of the CU\n", w, names[kd], (double)h / N, (double)h / N / w);
        }
    }
    return 0;
}
