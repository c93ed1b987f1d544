// Gaussian pyramid + pyramidal Lucas-Kanade tracking (a7).
//
// Replaces cv2.calcOpticalFlowPyrLK as called by getTransformKLT.getTrackedPointsKLT
// (reference getTransformKLT.py:317-381; LK_PARAMS :77-81: winSize (15,15), maxLevel 3,
// criteria (EPS|COUNT, 10, 0.03); default minEigThreshold 1e-4).  Algorithm statement and
// what is (un)pinned: oracle/c/warp_klt.c.  Integer window sums are exact (int64), every
// float operation is an explicit round-to-nearest intrinsic => bit-identical to the oracle.
//
// pyr_down: the live shapes stream whole rows - pyr_down_rows_kernel (2024 -> 1012) and
//   pyr_down2_rows_kernel (1012 -> 506 -> 253 in one pass), see the comments at those kernels;
//   other shapes use the tiled kernel: 256-thread block -> 64x16 output tile, the (131 x 35) u8
//   input tile staged in LDS with REFLECT_101 addressing, filtered horizontally into LDS (u16),
//   then vertically.
// klt: ONE WAVEFRONT PER FEATURE (64-thread workgroup), all 4 levels in one launch.  Per
//   level the 18x18 neighbourhood of the previous image is staged in LDS, Scharr
//   derivatives are formed on the fly (never materialised in HBM: saves 2 x int16 x 4.1 MP
//   per scan), the 15x15 patch (I, Ix, Iy) lives in registers (4 samples per lane), the
//   2x2 normal matrix and the per-iteration mismatch vector are wave-wide integer
//   reductions (32-bit DPP sums per 16-lane row, 64-bit scalar sum of the four rows), and a
//   32x32 neighbourhood of the next image is cached in LDS per level (reloaded only when the
//   window leaves it).  Control flow is wave-uniform.
#include "roam_internal.h"

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

// ------------------------------------------------------------------------------ pyrDown
#define PD_TW 64
#define PD_TH 32
#define PD_IW (2 * PD_TW + 3)   // 131
#define PD_IH (2 * PD_TH + 3)   // 35
#define PD_IWP 132

#define PD_FW 136               // fast-path tile row: 34 dwords starting at global column 2*ox0-4
__global__ __launch_bounds__(256) void pyr_down_kernel(const uint8_t *__restrict__ src, int64_t src_lane_stride,
                                                       int w, int h, uint8_t *__restrict__ dst,
                                                       int64_t dst_lane_stride, int dw, int dh)
{
    __shared__ __align__(16) uint8_t tin[PD_IH][PD_FW];
    __shared__ __align__(16) uint16_t hb[PD_IH][PD_TW];
    const int b = blockIdx.z;
    const int ox0 = blockIdx.x * PD_TW, oy0 = blockIdx.y * PD_TH;
    const uint8_t *s = src + (int64_t)b * src_lane_stride;
    uint8_t *d = dst + (int64_t)b * dst_lane_stride;
    const int t = threadIdx.x;
    const int gx0 = 2 * ox0 - 4, gy0 = 2 * oy0 - 2;
    // interior tile with dword-aligned rows: wide loads, no border arithmetic (block-uniform)
    const bool fast = ((w & 3) == 0) && ((src_lane_stride & 3) == 0) && gx0 >= 0 && gx0 + PD_FW <= w && gy0 >= 0 &&
                      gy0 + PD_IH <= h && ox0 + PD_TW <= dw && oy0 + PD_TH <= dh && ((dw & 1) == 0) &&
                      ((dst_lane_stride & 3) == 0);
    if (fast) {
        // 35 rows x 34 dwords, coalesced 136-byte row segments
        for (int i = t; i < PD_IH * (PD_FW / 4); i += 256) {
            const int ty = i / (PD_FW / 4), q = i - ty * (PD_FW / 4);
            reinterpret_cast<uint32_t *>(&tin[ty][0])[q] =
                *reinterpret_cast<const uint32_t *>(s + (int64_t)(gy0 + ty) * w + gx0 + 4 * q);
        }
        __syncthreads();
        // horizontal 5-tap: output ox uses tile bytes 2+2ox .. 6+2ox; a thread makes the pair (2q, 2q+1)
        // from bytes 4q+2 .. 4q+8, i.e. dwords q, q+1, q+2
        for (int i = t; i < PD_IH * (PD_TW / 2); i += 256) {
            const int ty = i / (PD_TW / 2), q = i - ty * (PD_TW / 2);
            const uint32_t *row = reinterpret_cast<const uint32_t *>(&tin[ty][0]);
            const uint32_t d0 = row[q], d1 = row[q + 1], d2 = row[q + 2];
            const int b2 = (d0 >> 16) & 255, b3 = d0 >> 24, b4 = d1 & 255, b5 = (d1 >> 8) & 255, b6 = (d1 >> 16) & 255,
                      b7 = d1 >> 24, b8 = d2 & 255;
            const uint32_t h0 = (uint32_t)(b2 + 4 * b3 + 6 * b4 + 4 * b5 + b6);
            const uint32_t h1 = (uint32_t)(b4 + 4 * b5 + 6 * b6 + 4 * b7 + b8);
            reinterpret_cast<uint32_t *>(&hb[ty][0])[q] = h0 | (h1 << 16);
        }
        __syncthreads();
        // vertical 5-tap on 4 adjacent outputs, one packed store per thread
        for (int i = t; i < PD_TH * (PD_TW / 4); i += 256) {
            const int oy = i >> 4, q4 = (i & 15) * 4;
            uint32_t acc[4] = {0, 0, 0, 0};
            const int kw[5] = {1, 4, 6, 4, 1};
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const uint2 v = *reinterpret_cast<const uint2 *>(&hb[2 * oy + j][q4]);
                acc[0] += kw[j] * (v.x & 0xffff); acc[1] += kw[j] * (v.x >> 16);
                acc[2] += kw[j] * (v.y & 0xffff); acc[3] += kw[j] * (v.y >> 16);
            }
            const uint32_t o0 = (acc[0] + 128) >> 8, o1 = (acc[1] + 128) >> 8, o2 = (acc[2] + 128) >> 8, o3 = (acc[3] + 128) >> 8;
            uint8_t *o = d + (int64_t)(oy0 + oy) * dw + ox0 + q4;
            if ((dw & 3) == 0) *reinterpret_cast<uint32_t *>(o) = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
            else {
                *reinterpret_cast<uint16_t *>(o) = (uint16_t)(o0 | (o1 << 8));
                *reinterpret_cast<uint16_t *>(o + 2) = (uint16_t)(o2 | (o3 << 8));
            }
        }
        return;
    }
    for (int i = t; i < PD_IH * PD_IW; i += 256) {
        int ty = i / PD_IW, tx = i - ty * PD_IW;
        int sy = reflect101(2 * oy0 - 2 + ty, h), sx = reflect101(2 * ox0 - 2 + tx, w);
        tin[ty][tx] = s[(int64_t)sy * w + sx];
    }
    __syncthreads();
    for (int i = t; i < PD_IH * PD_TW; i += 256) {
        int ty = i / PD_TW, ox = i - ty * PD_TW;
        const uint8_t *r = &tin[ty][2 * ox];
        hb[ty][ox] = (uint16_t)(r[0] + 4 * r[1] + 6 * r[2] + 4 * r[3] + r[4]);
    }
    __syncthreads();
    for (int i = t; i < PD_TH * PD_TW; i += 256) {
        int oy = i / PD_TW, ox = i - oy * PD_TW;
        int X = ox0 + ox, Y = oy0 + oy;
        if (X < dw && Y < dh) {
            int acc = hb[2 * oy][ox] + 4 * hb[2 * oy + 1][ox] + 6 * hb[2 * oy + 2][ox] +
                      4 * hb[2 * oy + 3][ox] + hb[2 * oy + 4][ox];
            d[(int64_t)Y * dw + X] = (uint8_t)((acc + 128) >> 8);
        }
    }
}

__device__ __forceinline__ uint32_t pyr_hpair(uint32_t d0, uint32_t d1, uint32_t d2)
{
    // bytes b0..b3 = d0, b4..b7 = d1, b8.. = d2; left output = b2+4b3+6b4+4b5+b6, right = b4+4b5+6b6+4b7+b8
    const uint32_t mid = __builtin_amdgcn_alignbyte(d1, d0, 2);                           // b2 b3 b4 b5
    const uint32_t lo = __builtin_amdgcn_udot4(d1, 0x00010000u, __builtin_amdgcn_udot4(mid, 0x04060401u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(d2, 0x00000001u, __builtin_amdgcn_udot4(d1, 0x04060401u, 0u, false), false);
    return lo | (hi << 16);
}

// ---- streaming variant for dword-aligned images up to 2048 wide (pyramid levels 0->1, 1->2)
// One block owns PR_CH output rows over the FULL image width and streams the 2*PR_CH+3 input rows
// through LDS: every input row is read once, as one contiguous w-byte burst (DRAM-page friendly:
// the tiled variant above touches 136-byte segments and measured 0.87 TB/s), 4 rows of loads are
// kept in flight in registers ahead of the row being filtered, the horizontally filtered rows
// live in a 5-deep LDS ring (u16 pairs), and one output row is emitted every second input row.
#ifndef PR_CH
#define PR_CH 32
#endif
#define PR_MAXW 2048
#define PR_AHEAD 4
__global__ __launch_bounds__(256) void pyr_down_rows_kernel(const uint8_t *__restrict__ src, int64_t src_lane_stride,
                                                            int w, int h, uint8_t *__restrict__ dst,
                                                            int64_t dst_lane_stride, int dw, int dh)
{
    __shared__ __align__(16) uint32_t rowbuf[PR_MAXW / 4 + 2];            // dword 0 = left pad, 1.. = pixels, then right pad
    __shared__ __align__(16) uint32_t hb[5][PR_MAXW / 4];                  // horizontally filtered rows, 2 x u16 per dword
    const int b = blockIdx.y, t = threadIdx.x;
    const int oy0 = blockIdx.x * PR_CH;
    const int nout = min(PR_CH, dh - oy0);
    const int nin = 2 * nout + 3;
    const uint8_t *s = src + (int64_t)b * src_lane_stride;
    uint8_t *d = dst + (int64_t)b * dst_lane_stride;
    const int wq = w >> 2;                       // dwords per input row (<= 512)
    const int npair = dw >> 1;                   // output pairs per row (dw even)
    uint32_t pre[PR_AHEAD][2];
    auto load_row = [&](int i, uint32_t (&r)[2]) {
        const int sy = reflect101(2 * oy0 - 2 + i, h);
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(s + (int64_t)sy * w);
        r[0] = (t < wq) ? rp[t] : 0u;
        r[1] = (t + 256 < wq) ? rp[t + 256] : 0u;
    };
#pragma unroll
    for (int k = 0; k < PR_AHEAD; k++) if (k < nin) load_row(k, pre[k]);
    for (int i0 = 0; i0 < nin; i0 += PR_AHEAD) {
#pragma unroll
        for (int k = 0; k < PR_AHEAD; k++) {
            const int i = i0 + k;
            if (i < nin) {
                // registers -> LDS row, with REFLECT_101 pads (pixel -2,-1 and w, w+1)
                if (t < wq) rowbuf[1 + t] = pre[k][0];
                if (t + 256 < wq) rowbuf[1 + t + 256] = pre[k][1];
                if (t == 0) {
                    const uint32_t d0 = pre[k][0];                       // pixels 0..3
                    rowbuf[0] = (((d0 >> 16) & 255) << 16) | (((d0 >> 8) & 255) << 24);   // bytes 2,3 of the pad = px[2], px[1]
                }
                if (t == ((wq - 1) & 255)) {
                    const uint32_t dl = ((wq - 1) < 256) ? pre[k][0] : pre[k][1];          // pixels w-4..w-1
                    rowbuf[1 + wq] = ((dl >> 16) & 255) | (((dl >> 8) & 255) << 8);        // px[w] = px[w-2], px[w+1] = px[w-3]
                }
                if (i + PR_AHEAD < nin) load_row(i + PR_AHEAD, pre[k]);
                __syncthreads();
                // horizontal 5-tap for output pairs (2q, 2q+1): pixels 4q-2 .. 4q+4 = LDS dwords q, q+1, q+2
                uint32_t *hrow = hb[i % 5];
                for (int q = t; q < npair; q += 256) {
                    const uint32_t d0 = rowbuf[q], d1 = rowbuf[q + 1], d2 = rowbuf[q + 2];
                    // bytes b0..b3 = d0, b4..b7 = d1, b8.. = d2; left output = b2+4b3+6b4+4b5+b6, right = b4+4b5+6b6+4b7+b8
                    const uint32_t mid = __builtin_amdgcn_alignbyte(d1, d0, 2);                  // b2 b3 b4 b5
                    const uint32_t lo = __builtin_amdgcn_udot4(d1, 0x00010000u, __builtin_amdgcn_udot4(mid, 0x04060401u, 0u, false), false);
                    const uint32_t hi = __builtin_amdgcn_udot4(d2, 0x00000001u, __builtin_amdgcn_udot4(d1, 0x04060401u, 0u, false), false);
                    hrow[q] = lo | (hi << 16);
                }
                __syncthreads();
                if (i >= 4 && !(i & 1)) {
                    const int ko = (i - 4) >> 1;                                          // output row oy0 + ko uses input rows i-4 .. i
                    const uint32_t *r0 = hb[(i - 4) % 5], *r1 = hb[(i - 3) % 5], *r2 = hb[(i - 2) % 5], *r3 = hb[(i - 1) % 5], *r4 = hb[i % 5];
                    uint16_t *orow = reinterpret_cast<uint16_t *>(d + (int64_t)(oy0 + ko) * dw);
                    for (int q = t; q < npair; q += 256) {
                        // two u16 column sums per dword; 16 * 16 * 255 + 128 < 65536, so the halves never carry
                        // into each other and the 5-tap sum is plain 32-bit arithmetic on the packed pair
                        const uint32_t v0 = r0[q], v1 = r1[q], v2 = r2[q], v3 = r3[q], v4 = r4[q];
                        const uint32_t sum = (v0 + v4) + 4u * (v1 + v3) + 6u * v2 + 0x00800080u;
                        orow[q] = (uint16_t)__builtin_amdgcn_perm(0u, sum, 0x0c0c0301u);       // (lo >> 8) | ((hi >> 8) << 8)
                    }
                }
            }
        }
    }
}

// ---- barrier-free streaming variant for the widest level (2024 -> 1012): ONE WAVEFRONT per band of output rows.
// A lane owns 8 consecutive pixel dwords (32 pixels) of every input row: two 16-byte loads; the horizontal 5-tap
// needs the neighbouring lanes' edge dwords only (DPP wave shifts), the REFLECT_101 pads are synthesised in the
// edge lanes; the five most recent horizontally filtered rows live in REGISTERS (5 x 8 packed u16 pairs per lane,
// the row loop is unrolled by ten so that every ring index is static), and an output row leaves as one 16-byte
// store per lane.  No LDS, no barriers; two input rows are prefetched ahead of the one being filtered.
#define PW_DPL 8              // pixel dwords per lane -> rows up to 64 * 8 * 4 = 2048 pixels
#define PW_CH 32              // output rows per wavefront
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

struct PwRow { uint32_t d[PW_DPL]; };

// dk: this lane's pixels are zero for every row of the band (pyr_dark_kernel) - its loads are pointed at the first bytes of the image,
// which lie beyond the maximum range too: zeros out of the cache instead of zeros out of HBM, with the same instructions on both paths
__device__ __forceinline__ void pw_load(const uint8_t *__restrict__ s, int w, int h, int wq, int sy, int b0, PwRow &r, bool dk = false)
{
    const uint32_t *rp = dk ? reinterpret_cast<const uint32_t *>(s) : reinterpret_cast<const uint32_t *>(s + (int64_t)sy * w) + b0;
    if (sy < h - 1) {                                    // reading past the row end stays inside the image
        const u32x4_a4 a = *reinterpret_cast<const u32x4_a4 *>(rp), c = *reinterpret_cast<const u32x4_a4 *>(rp + 4);
        r.d[0] = a.x; r.d[1] = a.y; r.d[2] = a.z; r.d[3] = a.w; r.d[4] = c.x; r.d[5] = c.y; r.d[6] = c.z; r.d[7] = c.w;
    } else {
#pragma unroll
        for (int j = 0; j < PW_DPL; j++) r.d[j] = (b0 + j < wq) ? rp[j] : 0u;
    }
}

// horizontal 5-tap of one row held in registers -> PW_DPL packed output pairs
__device__ __forceinline__ void pw_hfilter(const PwRow &r, int lane, int jstar, uint32_t (&out)[PW_DPL])
{
    uint32_t D[PW_DPL + 2];
    D[0] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r.d[PW_DPL - 1], 0x138, 0xf, 0xf, false);   // lane - 1
    D[PW_DPL + 1] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r.d[0], 0x130, 0xf, 0xf, false);   // lane + 1
#pragma unroll
    for (int j = 0; j < PW_DPL; j++) D[1 + j] = r.d[j];
    if (lane == 0) D[0] = (((r.d[0] >> 16) & 255u) << 16) | (((r.d[0] >> 8) & 255u) << 24);          // px[-2] = px[2], px[-1] = px[1]
#pragma unroll
    for (int j = 0; j < PW_DPL; j++)                                                                 // px[w] = px[w-2], px[w+1] = px[w-3]
        if (j == jstar) D[2 + j] = ((D[1 + j] >> 16) & 255u) | (((D[1 + j] >> 8) & 255u) << 8);
#pragma unroll
    for (int j = 0; j < PW_DPL; j++) out[j] = pyr_hpair(D[j], D[1 + j], D[2 + j]);
}

// which lanes of pyr_down_wave_kernel see nothing in a band: every pixel of the lane's 32-pixel segment (with its filter taps) in the
// band's input rows lies beyond the maximum range of the sampling map - geometry only, once per engine.  Bit l of dark[band].
__global__ __launch_bounds__(64) void pyr_dark_kernel(const uint32_t *__restrict__ map, int w, int h, int cols, int dh, unsigned long long *__restrict__ dark)
{
    const int band = blockIdx.x, lane = threadIdx.x, oy0 = band * PW_CH, nout = min(PW_CH, dh - oy0);
    const int r0 = max(0, 2 * oy0 - 2), r1 = min(h - 1, 2 * oy0 + 2 * nout + 1);
    const int c0 = max(0, 32 * lane - 4), c1 = min(w - 1, 32 * lane + 35);
    bool dk = c0 <= c1;
    for (int r = r0; r <= r1 && dk; r++)
        for (int c = c0; c <= c1; c++)
            if ((int)(map[(int64_t)r * w + c] & 4095u) < cols) { dk = false; break; }
    const unsigned long long m = __ballot(dk);
    if (lane == 0) dark[band] = m;
}

size_t pyr_dark_words(int h) { return (size_t)(((h + 1) / 2 + PW_CH - 1) / PW_CH); }

// fills dark (pyr_dark_words(h) 64-bit words) for a w x h level-0 image sampled through `map`; the table is only valid - and only then
// left non-zero - when the image's first 32 bytes are themselves beyond the maximum range (the dummy source of the dark lanes)
hipError_t launch_pyr_dark(hipStream_t st, const uint32_t *map, int w, int h, int cols, unsigned long long *dark)
{
    const int dh = (h + 1) / 2, bands = (dh + PW_CH - 1) / PW_CH;
    hipLaunchKernelGGL(pyr_dark_kernel, dim3(bands), dim3(64), 0, st, map, w, h, cols, dh, dark);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    unsigned long long first = 0;
    if ((e = hipMemcpyAsync(&first, dark, sizeof(first), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    if (!(first & 1ull)) return hipMemsetAsync(dark, 0, sizeof(unsigned long long) * bands, st);
    return hipSuccess;
}

__global__ __launch_bounds__(256) void pyr_down_wave_kernel(const uint8_t *__restrict__ src, int64_t src_lane_stride,
                                                            int w, int h, uint8_t *__restrict__ dst,
                                                            int64_t dst_lane_stride, int dw, int dh, int bands,
                                                            const unsigned long long *__restrict__ dark)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int band = blockIdx.x * 4 + wv;
    if (band >= bands) return;
    const int b = blockIdx.y;
    const int oy0 = band * PW_CH;
    const int nout = min(PW_CH, dh - oy0);
    const int nin = 2 * nout + 3;
    const uint8_t *s = src + (int64_t)b * src_lane_stride;
    uint8_t *d = dst + (int64_t)b * dst_lane_stride;
    const int wq = w >> 2, npair = dw >> 1;
    const int b0 = lane * PW_DPL;
    const int jstar = wq - 1 - b0;                       // local index of the last pixel dword of the row (if in this lane)
    uint32_t R[5][PW_DPL];
    PwRow pre[2];
    // (dark: destinations that were zero-filled once and are written by this kernel only - the engine's pyramids)
    const bool dk = dark && ((dark[band] >> lane) & 1ull);
    pw_load(s, w, h, wq, reflect101(2 * oy0 - 2, h), b0, pre[0], dk);
    if (nin > 1) pw_load(s, w, h, wq, reflect101(2 * oy0 - 1, h), b0, pre[1], dk);

#define PW_STEP(K)                                                                                              \
    if (i0 + (K) < nin) {                                                                                       \
        const int i = i0 + (K);                                                                                 \
        pw_hfilter(pre[(K) & 1], lane, jstar, R[(K) % 5]);                                                      \
        if (i + 2 < nin) pw_load(s, w, h, wq, reflect101(2 * oy0 - 2 + i + 2, h), b0, pre[(K) & 1], dk);        \
        if (i >= 4 && !((K) & 1)) {                                                                             \
            uint32_t o[PW_DPL / 2];                                                                             \
            _Pragma("unroll") for (int m = 0; m < PW_DPL / 2; m++) {                                            \
                const uint32_t sa = (R[((K) + 1) % 5][2 * m] + R[(K) % 5][2 * m]) + 4u * (R[((K) + 2) % 5][2 * m] + R[((K) + 4) % 5][2 * m]) + \
                                    6u * R[((K) + 3) % 5][2 * m] + 0x00800080u;                                 \
                const uint32_t sb = (R[((K) + 1) % 5][2 * m + 1] + R[(K) % 5][2 * m + 1]) +                      \
                                    4u * (R[((K) + 2) % 5][2 * m + 1] + R[((K) + 4) % 5][2 * m + 1]) + 6u * R[((K) + 3) % 5][2 * m + 1] + 0x00800080u; \
                o[m] = __builtin_amdgcn_perm(sb, sa, 0x07050301u);                                              \
            }                                                                                                   \
            uint8_t *orow = d + (int64_t)(oy0 + ((i - 4) >> 1)) * dw + 2 * b0;                                  \
            if (dk) { }                                                                                         \
            else if (b0 + PW_DPL <= npair) *reinterpret_cast<u32x4_a4 *>(orow) = u32x4_a4{o[0], o[1], o[2], o[3]}; \
            else {                                                                                              \
                _Pragma("unroll") for (int m = 0; m < PW_DPL / 2; m++)                                          \
                    if (b0 + 2 * m + 1 < npair) reinterpret_cast<uint32_t *>(orow)[m] = o[m];                   \
                    else if (b0 + 2 * m < npair) reinterpret_cast<uint16_t *>(orow)[2 * m] = (uint16_t)o[m];    \
            }                                                                                                   \
        }                                                                                                       \
    }
    // ring slot of input row i is i % 5; the loop advances ten rows so that K % 5 == i % 5 and K & 1 == i & 1:
    // row i - 4 is in slot (K + 1) % 5, i - 3 in (K + 2) % 5, i - 2 in (K + 3) % 5, i - 1 in (K + 4) % 5
    for (int i0 = 0; i0 < nin; i0 += 10) {
        PW_STEP(0) PW_STEP(1) PW_STEP(2) PW_STEP(3) PW_STEP(4) PW_STEP(5) PW_STEP(6) PW_STEP(7) PW_STEP(8) PW_STEP(9)
    }
#undef PW_STEP
}

// ---- two levels in one pass (1012 -> 506 -> 253): the 506-wide level is neither dword aligned nor large
// enough to stream well on its own (the tiled kernel above reaches 0.8 TB/s on it).  A block produces PF_CC
// rows of the SECOND output level: it streams the input rows exactly like pyr_down_rows_kernel to make the
// 2 * PF_CC + 3 rows of the first output level those need (the 3 halo rows are recomputed by the neighbouring
// block: 9 % more input reads, no re-read of the intermediate level), keeps them in LDS with their
// REFLECT_101 pads, writes the ones it owns, and filters the kept rows once more.
#ifndef PF_CC
#define PF_CC 12
#endif
#define PF_NB (2 * PF_CC + 3)
#define PF_PITCH 528              // 4 pad bytes + up to 512 pixels + 2 pad bytes, multiple of 16
#define PF_MAXW 1024
__global__ __launch_bounds__(256) void pyr_down2_rows_kernel(const uint8_t *__restrict__ src, int64_t src_lane_stride,
                                                             int w, int h, uint8_t *__restrict__ dst,
                                                             int64_t dst_lane_stride, int dw, int dh,
                                                             uint8_t *__restrict__ dst2, int64_t dst2_lane_stride,
                                                             int dw2, int dh2)
{
    __shared__ __align__(16) uint32_t rowbuf[PF_MAXW / 4 + 2];
    __shared__ __align__(16) uint32_t hb[5][PF_MAXW / 4];
    __shared__ __align__(16) uint8_t brow[PF_NB][PF_PITCH];
    const int b = blockIdx.y, t = threadIdx.x;
    const int c0 = blockIdx.x * PF_CC;
    const int nc = min(PF_CC, dh2 - c0);
    const int lo = max(2 * c0 - 2, 0), hi = min(2 * (c0 + nc - 1) + 2, dh - 1);      // first-level rows kept
    const int nout = hi - lo + 1;
    const int own0 = 2 * c0, own1 = min(2 * (c0 + nc), dh);                           // first-level rows written
    const int oy0 = lo;
    const int nin = 2 * nout + 3;
    const uint8_t *s = src + (int64_t)b * src_lane_stride;
    uint8_t *d = dst + (int64_t)b * dst_lane_stride;
    const int wq = w >> 2;                       // dwords per input row (<= 256)
    const int npair = dw >> 1;                   // output pairs per row (dw even)
    uint32_t pre[PR_AHEAD];
    auto load_row = [&](int i, uint32_t &r) {
        const int sy = reflect101(2 * oy0 - 2 + i, h);
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(s + (int64_t)sy * w);
        r = (t < wq) ? rp[t] : 0u;
    };
#pragma unroll
    for (int k = 0; k < PR_AHEAD; k++) if (k < nin) load_row(k, pre[k]);
    for (int i0 = 0; i0 < nin; i0 += PR_AHEAD) {
#pragma unroll
        for (int k = 0; k < PR_AHEAD; k++) {
            const int i = i0 + k;
            if (i < nin) {
                if (t < wq) rowbuf[1 + t] = pre[k];
                if (t == 0) rowbuf[0] = (((pre[k] >> 16) & 255) << 16) | (((pre[k] >> 8) & 255) << 24);
                if (t == wq - 1) rowbuf[1 + wq] = ((pre[k] >> 16) & 255) | (((pre[k] >> 8) & 255) << 8);
                if (i + PR_AHEAD < nin) load_row(i + PR_AHEAD, pre[k]);
                __syncthreads();
                uint32_t *hrow = hb[i % 5];
                if (t < npair) hrow[t] = pyr_hpair(rowbuf[t], rowbuf[t + 1], rowbuf[t + 2]);
                __syncthreads();
                if (i >= 4 && !(i & 1)) {
                    const int ko = (i - 4) >> 1, row = oy0 + ko;
                    const uint32_t *r0 = hb[(i - 4) % 5], *r1 = hb[(i - 3) % 5], *r2 = hb[(i - 2) % 5], *r3 = hb[(i - 1) % 5], *r4 = hb[i % 5];
                    if (t < npair) {
                        const uint32_t sum = (r0[t] + r4[t]) + 4u * (r1[t] + r3[t]) + 6u * r2[t] + 0x00800080u;
                        const uint16_t o2 = (uint16_t)__builtin_amdgcn_perm(0u, sum, 0x0c0c0301u);
                        *reinterpret_cast<uint16_t *>(&brow[ko][4 + 2 * t]) = o2;
                        if (row >= own0 && row < own1) reinterpret_cast<uint16_t *>(d + (int64_t)row * dw)[t] = o2;
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int r = t; r < nout; r += 256) {                      // REFLECT_101 pads of the kept rows
        brow[r][2] = brow[r][4 + 2]; brow[r][3] = brow[r][4 + 1];
        brow[r][4 + dw] = brow[r][4 + dw - 2]; brow[r][4 + dw + 1] = brow[r][4 + dw - 3];
    }
    __syncthreads();
    // second level: two rows at a time (128 threads each), one output pair per thread
    uint8_t *d2 = dst2 + (int64_t)b * dst2_lane_stride;
    const int npair2 = (dw2 + 1) >> 1;
    const int half = t >> 7, tq = t & 127;
    for (int q0 = 0; q0 < npair2; q0 += 128) {
        const int q = q0 + tq;
        for (int ci = half; ci < nc; ci += 2) {
            const int c = c0 + ci;
            if (q < npair2) {
                uint32_t hv[5];
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const uint32_t *rw = reinterpret_cast<const uint32_t *>(&brow[reflect101(2 * c - 2 + j, dh) - lo][0]);
                    hv[j] = pyr_hpair(rw[q], rw[q + 1], rw[q + 2]);
                }
                const uint32_t sum = (hv[0] + hv[4]) + 4u * (hv[1] + hv[3]) + 6u * hv[2] + 0x00800080u;
                const uint32_t o2 = __builtin_amdgcn_perm(0u, sum, 0x0c0c0301u);
                uint8_t *orow = d2 + (int64_t)c * dw2;
                orow[2 * q] = (uint8_t)o2;
                if (2 * q + 1 < dw2) orow[2 * q + 1] = (uint8_t)(o2 >> 8);
            }
        }
    }
}

hipError_t launch_pyr_down(hipStream_t st, const uint8_t *src, int64_t src_lane_stride, int w, int h,
                           uint8_t *dst, int64_t dst_lane_stride, int B, const unsigned long long *dark)
{
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    const bool rows_ok = ((w & 3) == 0) && w >= 16 && w <= PR_MAXW && ((dw & 1) == 0) && ((src_lane_stride & 3) == 0) &&
                         ((dst_lane_stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(src) & 3) == 0) &&
                         ((reinterpret_cast<uintptr_t>(dst) & 1) == 0) && h >= 4;
    const bool wave_ok = rows_ok && ((w & 3) == 0) && (w >> 2) > 32 * PW_DPL && (w >> 2) <= 64 * PW_DPL && ((dw & 3) == 0) && h >= 8 &&
                         ((dst_lane_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 3) == 0);
    if (wave_ok) {
        const int bands = (dh + PW_CH - 1) / PW_CH;
        hipLaunchKernelGGL(pyr_down_wave_kernel, dim3((bands + 3) / 4, B), dim3(256), 0, st, src, src_lane_stride, w, h, dst,
                           dst_lane_stride, dw, dh, bands, dark);
        return hipGetLastError();
    }
    if (rows_ok) {
        dim3 grid((dh + PR_CH - 1) / PR_CH, B);
        hipLaunchKernelGGL(pyr_down_rows_kernel, grid, dim3(256), 0, st, src, src_lane_stride, w, h, dst, dst_lane_stride, dw, dh);
        return hipGetLastError();
    }
    dim3 grid((dw + PD_TW - 1) / PD_TW, (dh + PD_TH - 1) / PD_TH, B);
    hipLaunchKernelGGL(pyr_down_kernel, grid, dim3(256), 0, st, src, src_lane_stride, w, h, dst, dst_lane_stride, dw, dh);
    return hipGetLastError();
}

void pyr_desc_init(PyrDesc *d, int w, int h)
{
    int64_t off = 0;
    for (int l = 0; l < ROAM_PYR_LEVELS; l++) {
        d->w[l] = w; d->h[l] = h; d->off[l] = off;
        off += (((int64_t)w * h) + 255) & ~(int64_t)255;      // keep levels 256-B aligned
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    d->lane_stride = off;
}

hipError_t launch_build_pyramid(hipStream_t st, uint8_t *pyr, const PyrDesc &d, int B, const unsigned long long *dark_l0)
{
    for (int l = 0; l + 1 < ROAM_PYR_LEVELS; l++) {
        const int w = d.w[l], h = d.h[l], dw = d.w[l + 1], dh = d.h[l + 1];
        // the last two levels in one pass when the shapes allow it (1012 -> 506 -> 253)
        if (l + 2 < ROAM_PYR_LEVELS && (w & 3) == 0 && w >= 64 && w <= PF_MAXW && (dw & 1) == 0 && h >= 16 && dh >= 8 &&
            (d.lane_stride & 3) == 0 && (d.off[l] & 3) == 0 && (d.off[l + 1] & 1) == 0 && (reinterpret_cast<uintptr_t>(pyr) & 3) == 0) {
            const int dw2 = d.w[l + 2], dh2 = d.h[l + 2];
            dim3 grid((dh2 + PF_CC - 1) / PF_CC, B);
            hipLaunchKernelGGL(pyr_down2_rows_kernel, grid, dim3(256), 0, st, pyr + d.off[l], d.lane_stride, w, h,
                               pyr + d.off[l + 1], d.lane_stride, dw, dh, pyr + d.off[l + 2], d.lane_stride, dw2, dh2);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            l++;
            continue;
        }
        hipError_t e = launch_pyr_down(st, pyr + d.off[l], d.lane_stride, w, h, pyr + d.off[l + 1], d.lane_stride, B, l == 0 ? dark_l0 : nullptr);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------------------ KLT
#define KW 15
#define W_BITS 14
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

// All window products have operands of at most 15 bits (pixels <= 255, weights <= 2^14, |Ix|, |Iy| <= 4080, |diff| <= 8160):
// they are written with __mul24 so that they compile to the full-rate v_mad_i32_i24 instead of the quarter-rate
// 32-bit multiply.
// Window sums whose 64-pixel partial sums fit in 32 bits: |Ix|, |Iy| <= 4080 (Scharr of u8, bilinear average),
// |diff| <= 8160 (u8 << 5), so a 16-lane row (4 pixels per lane) stays below 64 * 8160 * 4080 = 2 130 739 200 < 2^31.
// The four DPP steps then run on single registers (the adds fold into the DPP instruction) and only the four
// row sums are widened to 64 bits, as scalars.  Exact integer arithmetic either way.
__device__ __forceinline__ long long wave_sum_rows_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);        // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);       // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);       // row_mirror
    return ((long long)__builtin_amdgcn_readlane(v, 0) + (long long)__builtin_amdgcn_readlane(v, 16)) +
           ((long long)__builtin_amdgcn_readlane(v, 32) + (long long)__builtin_amdgcn_readlane(v, 48));
}

typedef uint32_t u32_a1 __attribute__((aligned(1)));

// 32 x 32 neighbourhood of the next image with origin (tx0, ty0): 4 bytes per lane and pass, one dword LDS
// write; interior tiles use misaligned dword loads, border tiles the REFLECT_101 byte gather
__device__ __forceinline__ void load_j_tile(const uint8_t *__restrict__ J, int w, int h, int tx0, int ty0, int lane,
                                            uint8_t (*Jt8)[36])
{
    if (tx0 >= 0 && tx0 + 32 <= w && ty0 >= 0 && ty0 + 32 <= h) {
#pragma unroll
        for (int i = lane; i < 32 * 8; i += 64) {
            const int y = i >> 3, c4 = (i & 7) * 4;
            *reinterpret_cast<uint32_t *>(&Jt8[y][c4]) = *reinterpret_cast<const u32_a1 *>(J + (int64_t)(ty0 + y) * w + (tx0 + c4));
        }
    } else {
        for (int i = lane; i < 32 * 8; i += 64) {
            const int y = i >> 3, c4 = (i & 7) * 4;
            const uint8_t *jr = J + (int64_t)reflect101(ty0 + y, h) * w;
            const uint32_t pk = (uint32_t)jr[reflect101(tx0 + c4, w)] | ((uint32_t)jr[reflect101(tx0 + c4 + 1, w)] << 8) |
                                ((uint32_t)jr[reflect101(tx0 + c4 + 2, w)] << 16) | ((uint32_t)jr[reflect101(tx0 + c4 + 3, w)] << 24);
            *reinterpret_cast<uint32_t *>(&Jt8[y][c4]) = pk;
        }
    }
}

__device__ __forceinline__ void bilin_weights(float a, float b, int &iw00, int &iw01, int &iw10, int &iw11)
{
    const float S = (float)(1 << W_BITS);
    const float na = __fsub_rn(1.f, a), nb = __fsub_rn(1.f, b);
    iw00 = __float2int_rn(__fmul_rn(__fmul_rn(na, nb), S));
    iw01 = __float2int_rn(__fmul_rn(__fmul_rn(a, nb), S));
    iw10 = __float2int_rn(__fmul_rn(__fmul_rn(na, b), S));
    iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
}

__global__ __launch_bounds__(64) void klt_kernel(const uint8_t *__restrict__ prev_pyr,
                                                 const uint8_t *__restrict__ next_pyr, PyrDesc d,
                                                 const float *__restrict__ pts, const int32_t *__restrict__ count,
                                                 int kstride, float *__restrict__ next_out,
                                                 uint8_t *__restrict__ status_out, float *__restrict__ err_out)
{
    __shared__ __align__(16) uint8_t It8[18][20];               // 18x18 neighbourhood of the previous image (u8, dword rows)
    __shared__ __align__(16) short Dx[16][20], Dy[16][20];      // Scharr derivatives, rows padded to 40 B
    __shared__ __align__(16) uint8_t Jt8[32][36];   // cached 32x32 u8 neighbourhood of the next image (reloaded only when the window leaves it)
    const int k = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    if (count && k >= count[b]) return;
    const int64_t pidx = ((int64_t)b * kstride + k) * 2;
    const float ptx = pts[pidx], pty = pts[pidx + 1];
    const float halfWin = 7.f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float eps2 = __fmul_rn(0.03f, 0.03f);
    const float min_eig_thr = 1e-4f;
    float out_x = 0.f, out_y = 0.f, er = 0.f;
    int st = 1;

    // window pixels owned by this lane: row ry, columns 4g .. 4g+3 (15 x 15 window -> lanes 0..59, the
    // last group holds 3 pixels).  Four CONSECUTIVE pixels per lane let one pair of dword LDS reads per
    // image row feed all their bilinear taps (the first version read 16 shorts per lane and iteration
    // and was LDS-bound: PMC SQ_LDS_IDX_ACTIVE 80 % of the kernel).
    const int ry = lane >> 2, g4 = (lane & 3) * 4;
    int py_[4], px_[4];
    bool pv_[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { py_[q] = ry; px_[q] = g4 + q; pv_[q] = (ry < KW) && (g4 + q < KW); }
    // five consecutive bytes starting at column x0 of tile row y (two aligned dword reads)
    auto row5 = [&](int y, int x0, int (&out)[5]) {
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(&Jt8[y][x0 & ~3]);
        const unsigned long long v = (((unsigned long long)rp[1] << 32) | rp[0]) >> (8 * (x0 & 3));
#pragma unroll
        for (int i = 0; i < 5; i++) out[i] = (int)((v >> (8 * i)) & 255ull);
    };

    for (int level = ROAM_PYR_LEVELS - 1; level >= 0; level--) {
        const int w = d.w[level], h = d.h[level];
        const uint8_t *I = prev_pyr + (int64_t)b * d.lane_stride + d.off[level];
        const uint8_t *J = next_pyr + (int64_t)b * d.lane_stride + d.off[level];
        const float scale = 1.f / (float)(1 << level);
        float px = __fmul_rn(ptx, scale), py = __fmul_rn(pty, scale);
        float nx, ny;
        if (level == ROAM_PYR_LEVELS - 1) { nx = px; ny = py; }
        else { nx = __fmul_rn(out_x, 2.f); ny = __fmul_rn(out_y, 2.f); }
        out_x = nx; out_y = ny;
        px = __fsub_rn(px, halfWin); py = __fsub_rn(py, halfWin);
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        if (ipx < -KW || ipx >= w || ipy < -KW || ipy >= h) {
            if (level == 0) { st = 0; er = 0.f; }
            continue;
        }
        int iw00, iw01, iw10, iw11;
        bilin_weights(__fsub_rn(px, (float)ipx), __fsub_rn(py, (float)ipy), iw00, iw01, iw10, iw11);

        __syncthreads();
        if (ipx >= 1 && ipx + 19 <= w && ipy >= 1 && ipy + 17 <= h) {
            // interior window: 18 rows x 5 dwords (the rows of the tile are 20 bytes), misaligned dword loads
            for (int i = lane; i < 18 * 5; i += 64) {
                const int ty = i / 5, q = i - ty * 5;
                reinterpret_cast<uint32_t *>(&It8[ty][0])[q] =
                    *reinterpret_cast<const u32_a1 *>(I + (int64_t)(ipy - 1 + ty) * w + (ipx - 1 + 4 * q));
            }
        } else {
            for (int i = lane; i < 18 * 18; i += 64) {
                int ty = i / 18, tx = i - ty * 18;
                It8[ty][tx] = I[(int64_t)reflect101(ipy - 1 + ty, h) * w + reflect101(ipx - 1 + tx, w)];
            }
        }
        __syncthreads();
        {   // Scharr derivatives of a 16x16 block: lane -> row y, columns xg..xg+3, three rows of 6 bytes each
            const int y = lane >> 2, xg = (lane & 3) * 4;
            int tr[3][6];
#pragma unroll
            for (int rr = 0; rr < 3; rr++) {
                const uint32_t lo = *reinterpret_cast<const uint32_t *>(&It8[y + rr][xg]);
                const uint32_t hi = *reinterpret_cast<const uint32_t *>(&It8[y + rr][xg + 4]);
                tr[rr][0] = lo & 255; tr[rr][1] = (lo >> 8) & 255; tr[rr][2] = (lo >> 16) & 255; tr[rr][3] = lo >> 24;
                tr[rr][4] = hi & 255; tr[rr][5] = (hi >> 8) & 255;
            }
            short dxs[4], dys[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int X = ipx + xg + i, Y = ipy + y;
                int dx = 0, dy = 0;
                if (X >= 0 && Y >= 0 && X < w && Y < h) {
                    const int a00 = tr[0][i], a01 = tr[0][i + 1], a02 = tr[0][i + 2];
                    const int a10 = tr[1][i], a12 = tr[1][i + 2];
                    const int a20 = tr[2][i], a21 = tr[2][i + 1], a22 = tr[2][i + 2];
                    dx = 3 * (a02 + a22 - a00 - a20) + 10 * (a12 - a10);
                    dy = 3 * (a20 + a22 - a00 - a02) + 10 * (a21 - a01);
                }
                dxs[i] = (short)dx; dys[i] = (short)dy;
            }
            *reinterpret_cast<uint2 *>(&Dx[y][xg]) = make_uint2((uint32_t)(uint16_t)dxs[0] | ((uint32_t)(uint16_t)dxs[1] << 16),
                                                                (uint32_t)(uint16_t)dxs[2] | ((uint32_t)(uint16_t)dxs[3] << 16));
            *reinterpret_cast<uint2 *>(&Dy[y][xg]) = make_uint2((uint32_t)(uint16_t)dys[0] | ((uint32_t)(uint16_t)dys[1] << 16),
                                                                (uint32_t)(uint16_t)dys[2] | ((uint32_t)(uint16_t)dys[3] << 16));
        }
        __syncthreads();
        int Iv[4], Ix[4], Iy[4];
        int pA11 = 0, pA12 = 0, pA22 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { Iv[q] = 0; Ix[q] = 0; Iy[q] = 0; }
        if (ry < KW) {
            // image taps: rows ry+1, ry+2 of the tile, columns g4+1 .. g4+5
            int ia[5], ib[5];
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const uint32_t lo = *reinterpret_cast<const uint32_t *>(&It8[ry + 1 + rr][g4]);
                const uint32_t hi = *reinterpret_cast<const uint32_t *>(&It8[ry + 1 + rr][g4 + 4]);
                int *dst = rr ? ib : ia;
                dst[0] = (lo >> 8) & 255; dst[1] = (lo >> 16) & 255; dst[2] = lo >> 24; dst[3] = hi & 255; dst[4] = (hi >> 8) & 255;
            }
            // derivative taps: rows ry, ry+1, columns g4 .. g4+4 (4 shorts + 1)
            int xa[5], xb[5], ya[5], yb[5];
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const uint2 vx = *reinterpret_cast<const uint2 *>(&Dx[ry + rr][g4]);
                const uint2 vy = *reinterpret_cast<const uint2 *>(&Dy[ry + rr][g4]);
                const int x4 = Dx[ry + rr][g4 + 4], y4 = Dy[ry + rr][g4 + 4];
                int *dx_ = rr ? xb : xa, *dy_ = rr ? yb : ya;
                dx_[0] = (short)(vx.x & 0xffff); dx_[1] = (short)(vx.x >> 16); dx_[2] = (short)(vx.y & 0xffff); dx_[3] = (short)(vx.y >> 16); dx_[4] = x4;
                dy_[0] = (short)(vy.x & 0xffff); dy_[1] = (short)(vy.x >> 16); dy_[2] = (short)(vy.y & 0xffff); dy_[3] = (short)(vy.y >> 16); dy_[4] = y4;
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (pv_[q]) {
                    Iv[q] = DESCALE(__mul24(ia[q], iw00) + __mul24(ia[q + 1], iw01) + __mul24(ib[q], iw10) + __mul24(ib[q + 1], iw11), W_BITS - 5);
                    Ix[q] = DESCALE(__mul24(xa[q], iw00) + __mul24(xa[q + 1], iw01) + __mul24(xb[q], iw10) + __mul24(xb[q + 1], iw11), W_BITS);
                    Iy[q] = DESCALE(__mul24(ya[q], iw00) + __mul24(ya[q + 1], iw01) + __mul24(yb[q], iw10) + __mul24(yb[q + 1], iw11), W_BITS);
                    pA11 += __mul24(Ix[q], Ix[q]);
                    pA12 += __mul24(Ix[q], Iy[q]);
                    pA22 += __mul24(Iy[q], Iy[q]);
                }
        }
        const long long sA11 = wave_sum_rows_i32(pA11), sA12 = wave_sum_rows_i32(pA12), sA22 = wave_sum_rows_i32(pA22);
        const float A11 = __fmul_rn(__ll2float_rn(sA11), FLT_SCALE);
        const float A12 = __fmul_rn(__ll2float_rn(sA12), FLT_SCALE);
        const float A22 = __fmul_rn(__ll2float_rn(sA22), FLT_SCALE);
        float D = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
        const float dA = __fsub_rn(A11, A22);
        const float minEig = __fdiv_rn(
            __fsub_rn(__fadd_rn(A22, A11),
                      rn_sqrtf(__fadd_rn(__fmul_rn(dA, dA), __fmul_rn(__fmul_rn(4.f, A12), A12)))),
            (float)(2 * KW * KW));
        if (minEig < min_eig_thr || D < 1.1920929e-07f) {
            if (level == 0) st = 0;
            continue;
        }
        D = __fdiv_rn(1.f, D);
        nx = __fsub_rn(nx, halfWin); ny = __fsub_rn(ny, halfWin);
        float pdx = 0.f, pdy = 0.f;
        int tx0 = -(1 << 28), ty0 = -(1 << 28);            // cached tile origin (invalid)
        for (int j = 0; j < 10; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -KW || inx >= w || iny < -KW || iny >= h) {
                if (level == 0) st = 0;
                break;
            }
            bilin_weights(__fsub_rn(nx, (float)inx), __fsub_rn(ny, (float)iny), iw00, iw01, iw10, iw11);
            if (!(inx >= tx0 && inx - tx0 <= 15 && iny >= ty0 && iny - ty0 <= 15)) {
                tx0 = inx - 8; ty0 = iny - 8;
                __syncthreads();
                load_j_tile(J, w, h, tx0, ty0, lane, Jt8);
                __syncthreads();
            }
            const int jox = inx - tx0, joy = iny - ty0;
            int pb1 = 0, pb2 = 0;
            if (ry < KW) {
                int ra[5], rb[5];
                row5(joy + ry, jox + g4, ra);
                row5(joy + ry + 1, jox + g4, rb);
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (pv_[q]) {
                        int jv = DESCALE(__mul24(ra[q], iw00) + __mul24(ra[q + 1], iw01) + __mul24(rb[q], iw10) + __mul24(rb[q + 1], iw11), W_BITS - 5);
                        int diff = jv - Iv[q];
                        pb1 += __mul24(diff, Ix[q]);
                        pb2 += __mul24(diff, Iy[q]);
                    }
            }
            const long long sb1 = wave_sum_rows_i32(pb1), sb2 = wave_sum_rows_i32(pb2);
            const float b1 = __fmul_rn(__ll2float_rn(sb1), FLT_SCALE);
            const float b2 = __fmul_rn(__ll2float_rn(sb2), FLT_SCALE);
            const float ddx = __fmul_rn(__fsub_rn(__fmul_rn(A12, b2), __fmul_rn(A22, b1)), D);
            const float ddy = __fmul_rn(__fsub_rn(__fmul_rn(A12, b1), __fmul_rn(A11, b2)), D);
            nx = __fadd_rn(nx, ddx); ny = __fadd_rn(ny, ddy);
            out_x = __fadd_rn(nx, halfWin); out_y = __fadd_rn(ny, halfWin);
            if (__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)) <= eps2) break;
            if (j > 0 && fabsf(__fadd_rn(ddx, pdx)) < 0.01f && fabsf(__fadd_rn(ddy, pdy)) < 0.01f) {
                out_x = __fsub_rn(out_x, __fmul_rn(ddx, 0.5f));
                out_y = __fsub_rn(out_y, __fmul_rn(ddy, 0.5f));
                break;
            }
            pdx = ddx; pdy = ddy;
        }
        if (st && level == 0) {
            const float ex = __fsub_rn(out_x, halfWin), ey = __fsub_rn(out_y, halfWin);
            const int iex = (int)floorf(ex), iey = (int)floorf(ey);
            if (iex < -KW || iex >= w || iey < -KW || iey >= h) { st = 0; continue; }
            bilin_weights(__fsub_rn(ex, (float)iex), __fsub_rn(ey, (float)iey), iw00, iw01, iw10, iw11);
            if (!(iex >= tx0 && iex - tx0 <= 15 && iey >= ty0 && iey - ty0 <= 15)) {
                tx0 = iex - 8; ty0 = iey - 8;
                __syncthreads();
                load_j_tile(J, w, h, tx0, ty0, lane, Jt8);
                __syncthreads();
            }
            const int eox = iex - tx0, eoy = iey - ty0;
            int pe = 0;
            if (ry < KW) {
                int ra[5], rb[5];
                row5(eoy + ry, eox + g4, ra);
                row5(eoy + ry + 1, eox + g4, rb);
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (pv_[q]) {
                        int jv = DESCALE(__mul24(ra[q], iw00) + __mul24(ra[q + 1], iw01) + __mul24(rb[q], iw10) + __mul24(rb[q + 1], iw11), W_BITS - 5);
                        int diff = jv - Iv[q];
                        pe += diff < 0 ? -diff : diff;
                    }
            }
            const long long se = wave_sum_rows_i32(pe);
            er = __fmul_rn(__ll2float_rn(se), 1.f / (float)(32 * KW * KW));
        }
    }
    if (lane == 0) {
        next_out[pidx] = out_x; next_out[pidx + 1] = out_y;
        status_out[(int64_t)b * kstride + k] = (uint8_t)st;
        err_out[(int64_t)b * kstride + k] = er;
    }
}

hipError_t launch_klt(hipStream_t st, const uint8_t *prev_pyr, const uint8_t *next_pyr,
                      const PyrDesc &d, const float *pts, const int32_t *count, int K, int kstride,
                      int B, float *next, uint8_t *status, float *err)
{
    if (K <= 0 || B <= 0) return hipSuccess;
    hipLaunchKernelGGL(klt_kernel, dim3(K, B), dim3(64), 0, st, prev_pyr, next_pyr, d, pts, count, kstride,
                       next, status, err);
    return hipGetLastError();
}
