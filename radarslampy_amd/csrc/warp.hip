// polar -> Cartesian warp + u8 quantisation (a3, feeds a7).
//
// Replaces parseData.convertPolarImageToCartesian (reference parseData.py:100-135, i.e.
// cv2.warpPolar with WARP_POLAR_LINEAR|WARP_INVERSE_MAP|INTER_LINEAR|WARP_FILL_OUTLIERS)
// and the (img*255).astype(uint8) of getTransformKLT.py:356-357, fused: the f32 Cartesian
// image (16.4 MB) is only materialised when the caller asks for it.
//
// Arithmetic follows the published OpenCV pipeline operation by operation (see
// oracle/c/warp_klt.c for the statement of what is and is not pinned): float maps from
// sqrt + the degree-7 fastAtan polynomial, double-precision scale, 1/32-pixel coordinate
// rounding (round-half-even), table-equivalent bilinear weights, zero fill in range, wrap
// in azimuth.  Everything is written with IEEE round-to-nearest intrinsics so that the
// compiler cannot contract a*b+c into an FMA: results are bit-identical to the oracle.
//
// Layout: one thread produces 4 consecutive output pixels of one row and stores them as a
// single 32-bit word (u8) / 128-bit vector (f32): 256-thread blocks, grid = (ceil(W/4/256),
// W, lanes).  The polar source (0.8 MB u8 per scan) is gathered through L2.
#include "roam_internal.h"

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float R2D = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * R2D;
    const float p3 = -0.3258083974640975f * R2D;
    const float p5 = 0.1555786518463281f * R2D;
    const float p7 = -0.04432655554792128f * R2D;
    const float DE = (float)2.220446049250313e-16;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, DE));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, DE));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

template <bool U8>
__device__ __forceinline__ float polar_tap(const void *base, int64_t row_stride, int payload_off,
                                           int rows, int cols, int py, int px)
{
    if (px < 0 || px >= cols || py < 0 || py >= rows + 2) return 0.f;
    int r = py - 1;
    if (r < 0) r += rows;
    else if (r >= rows) r -= rows;
    if (U8) {
        const uint8_t *p = reinterpret_cast<const uint8_t *>(base) + (int64_t)r * row_stride + payload_off;
        return __fdiv_rn((float)p[px], 255.f);
    } else {
        const float *p = reinterpret_cast<const float *>(base) + (int64_t)r * row_stride;
        return p[px];
    }
}

template <bool U8>
__device__ __forceinline__ float warp_pixel(const void *base, int64_t row_stride, int payload_off,
                                            int rows, int cols, int R, double Kangle, double Kmag,
                                            int x, int y)
{
    const float fx = __fsub_rn((float)x, (float)R);
    const float fy = __fsub_rn((float)y, (float)R);
    const float mag = rn_sqrtf(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)));
    const float ang = __fmul_rn(fast_atan2_deg(fy, fx), (float)(3.14159265358979323846 / 180.0));
    const double rho = __ddiv_rn((double)mag, Kmag);
    const double phi = __ddiv_rn((double)ang, Kangle);
    const float mx = (float)rho;
    const float my = __fadd_rn((float)phi, 1.f);
    const int sx = __float2int_rn(__fmul_rn(mx, 32.f));
    const int sy = __float2int_rn(__fmul_rn(my, 32.f));
    const int ix = sx >> 5, iy = sy >> 5;
    const float wx1 = __fmul_rn((float)(sx & 31), 1.f / 32.f), wx0 = __fsub_rn(1.f, wx1);
    const float wy1 = __fmul_rn((float)(sy & 31), 1.f / 32.f), wy0 = __fsub_rn(1.f, wy1);
    const float w00 = __fmul_rn(wy0, wx0), w01 = __fmul_rn(wy0, wx1);
    const float w10 = __fmul_rn(wy1, wx0), w11 = __fmul_rn(wy1, wx1);
    const float s00 = polar_tap<U8>(base, row_stride, payload_off, rows, cols, iy, ix);
    const float s01 = polar_tap<U8>(base, row_stride, payload_off, rows, cols, iy, ix + 1);
    const float s10 = polar_tap<U8>(base, row_stride, payload_off, rows, cols, iy + 1, ix);
    const float s11 = polar_tap<U8>(base, row_stride, payload_off, rows, cols, iy + 1, ix + 1);
    float v = __fmul_rn(s00, w00);
    v = __fadd_rn(v, __fmul_rn(s01, w01));
    v = __fadd_rn(v, __fmul_rn(s10, w10));
    v = __fadd_rn(v, __fmul_rn(s11, w11));
    return v;
}

__device__ __forceinline__ uint32_t quant_u8(float v)
{
    return (uint32_t)(int)__fmul_rn(v, 255.f) & 0xffu;       // (img*255).astype(uint8): truncation
}

// the same for a bilinear sample of u8 codes: the weights are multiples of 2^-10 that sum to exactly 1 and every sample is at most 1, so
// every rounded product is at most its weight and every rounded partial sum at most the (representable) sum of the weights: 0 <= v <= 1,
// v * 255 <= 255 - the truncated value needs no mask
__device__ __forceinline__ uint32_t quant_u8_unit(float v)
{
    return (uint32_t)(int)__fmul_rn(v, 255.f);
}

template <bool U8>
__global__ __launch_bounds__(256) void polar_to_cart_kernel(WarpSrc src, int rows, int cols, int R,
                                                            uint8_t *__restrict__ cart_u8, int64_t u8_lane_stride,
                                                            float *__restrict__ cart_f32, int64_t f32_lane_stride)
{
    const int W = 2 * R;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int y = blockIdx.y, b = blockIdx.z;
    if (x0 >= W) return;
    const int64_t lane_sel = src.lane_index ? (int64_t)src.lane_index[b] : (int64_t)b;
    const void *base = U8 ? (const void *)(reinterpret_cast<const uint8_t *>(src.base) + lane_sel * src.lane_stride)
                          : (const void *)(reinterpret_cast<const float *>(src.base) + lane_sel * src.lane_stride);
    const double Kangle = 6.283185307179586476925286766559 / (double)rows;
    const double Kmag = (double)R / (double)cols;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        v[k] = (x0 + k < W) ? warp_pixel<U8>(base, src.row_stride, src.payload_off, rows, cols, R, Kangle, Kmag, x0 + k, y) : 0.f;
    if (cart_f32) {
        float *o = cart_f32 + (int64_t)b * f32_lane_stride + (int64_t)y * W + x0;
        if (x0 + 3 < W && ((W & 3) == 0)) *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
        else for (int k = 0; k < 4 && x0 + k < W; k++) o[k] = v[k];
    }
    if (cart_u8) {
        uint8_t *o = cart_u8 + (int64_t)b * u8_lane_stride + (int64_t)y * W + x0;
        if (x0 + 3 < W && ((W & 3) == 0) && ((u8_lane_stride & 3) == 0)) {
            uint32_t pk = quant_u8(v[0]) | (quant_u8(v[1]) << 8) | (quant_u8(v[2]) << 16) | (quant_u8(v[3]) << 24);
            *reinterpret_cast<uint32_t *>(o) = pk;
        } else
            for (int k = 0; k < 4 && x0 + k < W; k++) o[k] = (uint8_t)quant_u8(v[k]);
    }
}

// ------------------------------------------------------------------------------ map path
// The sampling map (sx, sy in 1/32 px) depends only on the geometry (rows, cols), not on
// the scan: the engine computes it ONCE with the exact arithmetic of warp_pixel and every
// later warp is a pure gather: 4 B of map per pixel shared by LB lanes of the batch, four
// u8 taps through L1/L2, 7 float ops, one packed 32-bit store per 4 pixels.
// u8 -> float32 decode: (float)k / 255.f for all 256 codes (verified exhaustively in tests/test_abi_cpu.py) without a divide and
// without float64: 1 / 255 split into a float32 head and tail, fma(k, head, k * tail) is the correctly rounded quotient for every
// code (round 5; rounds 1-4 went through (float)((double)k * (1.0 / 255.0)): a v_cvt_f64_u32 + a half-rate v_mul_f64 + a
// v_cvt_f32_f64 per sample, and the staging of the polar boxes - one decode per sample and scan - was a third of this kernel's vector
// instructions).  (float)(byte of a word) is ONE instruction (v_cvt_f32_ubyteN).
#ifndef WG_DECODE_F64
#define WG_DECODE_F64 0
#endif
__device__ __forceinline__ float code_to_f32(uint32_t k)
{
#if WG_DECODE_F64
    return (float)__dmul_rn((double)k, 1.0 / 255.0);
#else
    const float kf = (float)k;
    return __fmaf_rn(kf, 0x1.0101020000000p-8f, __fmul_rn(kf, -0x1.fdfdfe0000000p-33f));            // head + tail = 1 / 255 to 2^-57
#endif
}

// pack: ix [0,12) | iy [12,22) | fx [22,27) | fy [27,32)
__global__ __launch_bounds__(256) void warp_map_kernel(int rows, int cols, int R, uint32_t *__restrict__ map)
{
    const int W = 2 * R;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const double Kangle = 6.283185307179586476925286766559 / (double)rows;
    const double Kmag = (double)R / (double)cols;
    const float fx = __fsub_rn((float)x, (float)R);
    const float fy = __fsub_rn((float)y, (float)R);
    const float mag = rn_sqrtf(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)));
    const float ang = __fmul_rn(fast_atan2_deg(fy, fx), (float)(3.14159265358979323846 / 180.0));
    const double rho = __ddiv_rn((double)mag, Kmag);
    const double phi = __ddiv_rn((double)ang, Kangle);
    const float mx = (float)rho;
    const float my = __fadd_rn((float)phi, 1.f);
    const int sx = __float2int_rn(__fmul_rn(mx, 32.f));
    const int sy = __float2int_rn(__fmul_rn(my, 32.f));
    int ix = sx >> 5, iy = sy >> 5;
    if (ix > 4095) ix = 4095;                   // >= cols: both taps read zero anyway
    if (ix < 0) ix = 4095;
    if (iy < 0) iy = 0;
    if (iy > 1022) iy = 1022;
    map[(int64_t)y * W + x] = (uint32_t)ix | ((uint32_t)iy << 12) | ((uint32_t)(sx & 31) << 22) | ((uint32_t)(sy & 31) << 27);
}

hipError_t launch_warp_map(hipStream_t st, int rows, int cols, uint32_t *map)
{
    const int R = cols / 2, W = 2 * R;
    hipLaunchKernelGGL(warp_map_kernel, dim3((W + 255) / 256, W), dim3(256), 0, st, rows, cols, R, map);
    return hipGetLastError();
}

#ifndef WG_LB
#define WG_LB 32         // scans of the batch per workgroup (amortises the map read and the weights; 16 / 24 / 48 / 64 / 128 measured slower)
#endif
#define WG_TW 64          // tile width  (one wavefront = 64 consecutive pixels of a row)
#define WG_TH 16          // tile height
#define WG_BOX_ELEMS 4096     // polar samples staged per pass, already decoded to float32 (16 KB; WG_CELL: twice that)
#ifndef WG_CELL
#define WG_CELL 0             // (round 6 experiment) the box as CELLS {s[k][c], s[k+1][c]}: the four taps of a pixel are two neighbouring cells - ONE ds_read2_b64
#endif
#ifndef WG_FILL_U
#define WG_FILL_U 4       // scans whose box rows are loaded before the first load is consumed
#endif
// 256-thread block = 64 x 16 pixel tile; wave w owns rows 4w..4w+3, lane = x offset.
// The polar footprint of a tile is a small box (range span x azimuth span; median 310 samples,
// 52 x 7): it is staged in LDS with coalesced row loads (16 lanes per polar row, one misaligned
// dword per lane, four rows per load instruction) and the 4 bilinear taps per pixel become LDS reads - the PMC
// profile of the direct-gather version showed 28 L1 accesses per wave-level load and the texture
// addresser 63 % busy.  Because the box is small, the boxes of SEVERAL scans of the batch
// (WG_BOX_ELEMS / box size, up to WG_LB) are staged in one pass: the fill issues the loads of
// WG_FILL_U scans per wavefront before the first one is consumed, and a pass costs two
// barriers whatever the number of scans it covers.  Tiles whose footprint does not fit (next to
// the image centre, or straddling the 0/2pi seam) fall back to direct L1/L2 gathers; tiles
// beyond the maximum range write zeros.  A thread's 4 results (4 rows of one column) are
// transposed against its 3 quad neighbours with DPP quad broadcasts + v_perm_b32, so that every
// thread stores one whole dword of one row without going through LDS.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int M> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, M * 0x55, 0xf, 0xf, true);
}

// stage the polar box of U consecutive scans (l, l+1, ...).  A wavefront covers FOUR box rows per load
// instruction: 16 lanes per row, one misaligned dword (4 range bins) per lane, decoded to four floats and
// written with one 16-byte LDS store (the LDS row pitch bp is a multiple of 4 floats).  wave wvs takes the row
// groups wvs, wvs+4, ...; the U scans' loads are issued before the first is consumed.
typedef uint32_t u32_a1 __attribute__((aligned(1)));
template <int U, bool CHK, bool RAWK = false, bool CELL = false>
__device__ __forceinline__ void box_fill(const uint8_t *__restrict__ sp, int64_t lane_stride,
                                         const int32_t *__restrict__ lane_index, int l, int64_t row_stride, int rows,
                                         int cols, int mnx, int mny, int bw, int bp, int bh, int elems, int wvs, int lane,
                                         float *__restrict__ bq, int pp = 0)
{
    // pp != 0: the swizzled layout of the kernel below (WG_SWZ) - sample c of a row at float c + (c >> 5), pitch pp
    int64_t so[U];
#pragma unroll
    for (int u = 0; u < U; u++) so[u] = (lane_index ? (int64_t)lane_index[l + u] : (int64_t)(l + u)) * lane_stride;
    const int sub = lane >> 4, c4 = (lane & 15) * 4;
    for (int kg = 4 * wvs; kg < bh; kg += 16) {
        const int k = min(kg + sub, bh - 1);              // clamped lanes rewrite the last row
        int r = mny + k - 1;
        if (r < 0) r += rows; else if (r >= rows) r -= rows;
        for (int cb = 0; cb < bw; cb += 64) {
            const int c = min(cb + c4, bp - 4);           // clamped lanes rewrite the last column group
            const int x0 = mnx + c;                       // first range bin of this lane's dword
            const uint8_t *srow = sp + (__mul24(r, (int)row_stride) + x0);       // rows * stride < 2^31 (launcher requirement)
            float *drow = pp ? bq + (k * pp + c + (c >> 5)) : bq + (k * bp + c);
            uint32_t raw[U];
#pragma unroll
            for (int u = 0; u < U; u++) raw[u] = *reinterpret_cast<const u32_a1 *>(srow + so[u]);
#pragma unroll
            for (int u = 0; u < U; u++) {
                float4 v;
                // (CHK: the box reaches past the scan's last range bin - those samples are zero; most boxes do not, and are spared the
                // four compares and selects per dword)
                // RAWK: the codes as they are (integer blend, below): one v_cvt_f32_ubyteN per sample
                v.x = (!CHK || x0 < cols) ? (RAWK ? (float)(raw[u] & 255u) : code_to_f32(raw[u] & 255u)) : 0.f;
                v.y = (!CHK || x0 + 1 < cols) ? (RAWK ? (float)((raw[u] >> 8) & 255u) : code_to_f32((raw[u] >> 8) & 255u)) : 0.f;
                v.z = (!CHK || x0 + 2 < cols) ? (RAWK ? (float)((raw[u] >> 16) & 255u) : code_to_f32((raw[u] >> 16) & 255u)) : 0.f;
                v.w = (!CHK || x0 + 3 < cols) ? (RAWK ? (float)(raw[u] >> 24) : code_to_f32(raw[u] >> 24)) : 0.f;
                if (CELL) {
                    // sample (k, c) is the upper half of cell (k, c) and the lower half of cell (k - 1, c): two ds_write2_b32 each
                    float *d = bq + 2 * (k * bp + c) + u * 2 * elems;
                    d[0] = v.x; d[2] = v.y; d[4] = v.z; d[6] = v.w;
                    if (k > 0) { float *e = d - 2 * bp + 1; e[0] = v.x; e[2] = v.y; e[4] = v.z; e[6] = v.w; }
                } else
                if (pp) {                                 // (4-byte aligned only: four dword stores; the slot before a 32-sample block repeats its first sample)
                    float *d = drow + u * elems;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                    if ((c & 31) == 0 && c > 0) d[-1] = v.x;
                } else
                    *reinterpret_cast<float4 *>(drow + u * elems) = v;
            }
        }
    }
}

// ---- exact blend (round 6).  The KLT input is trunc(v * 255) with v = ((s00 w00 + s01 w01) + s10 w10) + s11 w11 in float32, s = code / 255
// and the weights multiples of 2^-10 that sum to 1.  With integer weights W = 1024 w the exact value of v * 255 is E = sum(W k) / 1024, a
// multiple of 2^-10, and the float32 chain (one rounding for the decode, one per product, three for the sums, one for the scaling, all
// operands non-negative) stays within 6.001 * 2^-24 * E < 9.2e-5 of it - a tenth of 2^-10.  So whenever E is NOT an integer the truncated
// float equals floor(E) = sum(W k) >> 10, whatever the roundings did; only a sum that is an exact non-zero multiple of 1024 (the float may
// land just below the integer: one pixel in ~1024) needs the float chain, and a sum of zero is zero either way.
// sum(W k) is formed in FLOAT32 all the same - 64 W (at most 2^16) and k (at most 255) are integers, every product and partial sum is an
// integer below 2^24, so three v_fma_f32 after one v_mul_f32 are exact in any order - because float32 multiply-adds are the cheapest
// vector instructions on this chip (SIMD-32: half the cycles of an integer instruction; a first version with v_dot4_u32_u8 on packed codes
// and 24-bit integer multiply-adds was bit-identical and took 14.3 ms against 10.5).  The box holds the codes as floats (one
// v_cvt_f32_ubyteN per sample instead of the three-instruction exact quotient by 255), v_cvt_u32_f32 turns the sum into the integer
// 64 sum(W k): byte 2 is the pixel and "multiple of 1024" reads "low half zero" (v_min3_u16 over a thread's four pixels).  Flagged pixels
// go to a per-wave list and are recomputed by the float32 chain from global memory after the dword stores they correct (same wave, after
// s_waitcnt vmcnt(0)): at the end of the workgroup, or earlier when the list runs full.
#ifndef WG_INT_BLEND
#define WG_INT_BLEND 0                      // NOT the default: bit-identical on every test, and slower (profiles/r06_warp_exact_blend.txt)
#endif
#ifndef WG_DIAG
#define WG_DIAG 0                           // timing experiments (wrong bytes): 1 = no fill, 2 = no blend loop
#endif
#define WG_FIX_CAP 320                      // entries per wave; flushed at 64 before a pass of at most 256 new ones

__global__ __launch_bounds__(256) void warp_gather_kernel(const uint32_t *__restrict__ map, const uint8_t *__restrict__ pool,
                                                          int64_t lane_stride, int64_t row_stride, int payload_off,
                                                          const int32_t *__restrict__ lane_index, int B, int rows,
                                                          int cols, int W, uint8_t *__restrict__ cart_u8,
                                                          int64_t u8_lane_stride, int gx, int gy, int total, int dark_stays_zero)
{
    __shared__ __align__(16) float box[WG_BOX_ELEMS * (WG_CELL ? 2 : 1)];
    __shared__ int red[4][4];
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own
    // L2), so XCD x is given the x-th contiguous eighth of the (scan group, tile row, tile) list:
    // the polar rows one scan group needs are then fetched into ONE L2 instead of all eight.
    const int per_xcd = (total + 7) >> 3;
    const int vid = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (vid >= total) return;
    const int bz = vid / (gx * gy), brem = vid - bz * (gx * gy);
    const int by = brem / gx, bx_ = brem - by * gx;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = bx_ * WG_TW + lane;
    const int y0 = by * WG_TH;
    const bool xin = x < W;
    const int l0 = bz * WG_LB, l1 = min(B, l0 + WG_LB);
    // scan -> byte offset of its payload; the argument is wave-uniform, so this is scalar code
    auto src_off = [&](int l) -> int64_t {
        const int64_t sel = lane_index ? (int64_t)lane_index[l] : (int64_t)l;
        return sel * lane_stride + payload_off;
    };
    int ixv[4], iyv[4];
    float w00[4], w01[4], w10[4], w11[4];
    uint32_t mw[4];
    bool in0[4];
    int mnx = 0x7fffffff, mxx = -1, mny = 0x7fffffff, mxy = -1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int y = y0 + wv * 4 + j;
        const bool ok = xin && y < W;
        const uint32_t m = ok ? map[(int64_t)y * W + x] : 4095u;
        const int ix = m & 4095, iy = (m >> 12) & 1023;
        const float wx1 = __fmul_rn((float)((m >> 22) & 31), 1.f / 32.f), wx0 = __fsub_rn(1.f, wx1);
        const float wy1 = __fmul_rn((float)(m >> 27), 1.f / 32.f), wy0 = __fsub_rn(1.f, wy1);
        w00[j] = __fmul_rn(wy0, wx0); w01[j] = __fmul_rn(wy0, wx1);
        w10[j] = __fmul_rn(wy1, wx0); w11[j] = __fmul_rn(wy1, wx1);
        ixv[j] = ix; iyv[j] = iy; mw[j] = m;
        in0[j] = ok && ix < cols;
        if (in0[j]) { mnx = min(mnx, ix); mxx = max(mxx, ix); mny = min(mny, iy); mxy = max(mxy, iy); }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        mnx = min(mnx, __shfl_xor(mnx, m)); mxx = max(mxx, __shfl_xor(mxx, m));
        mny = min(mny, __shfl_xor(mny, m)); mxy = max(mxy, __shfl_xor(mxy, m));
    }
    if (lane == 0) { red[wv][0] = mnx; red[wv][1] = mxx; red[wv][2] = mny; red[wv][3] = mxy; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        mnx = min(mnx, red[k][0]); mxx = max(mxx, red[k][1]); mny = min(mny, red[k][2]); mxy = max(mxy, red[k][3]);
    }
    // the box is the same for the whole workgroup: keep it in scalar registers
    mnx = __builtin_amdgcn_readfirstlane(mnx); mxx = __builtin_amdgcn_readfirstlane(mxx);
    mny = __builtin_amdgcn_readfirstlane(mny); mxy = __builtin_amdgcn_readfirstlane(mxy);
    const bool any = mxx >= 0;
    const int bw = mxx - mnx + 2, bh = mxy - mny + 2;
#ifndef WG_BP_MODE
#define WG_BP_MODE 0
#endif
    int bp_ = (bw + 3) & ~3;                          // LDS pitch of a box row: whole 16-byte stores
    if (WG_BP_MODE == 1 && (bp_ & 4) == 0) bp_ += 4;  // (experiment: an odd multiple of four floats - rows k, k + 1, .. start in different banks)
    if (WG_BP_MODE == 2) { while ((bp_ & 31) != 4) bp_ += 4; }
    const int bp = bp_;
    // WG_SWZ (experiment, round 6; NOT the default): adjacent pixels of a tile whose rows run along the range axis sit two samples apart, so
    // the 32 lanes of an LDS access group meet 16 banks.  With one spare float after every 32 samples of a box row the second half of
    // such a group lands on the other parity; the spare slot repeats the sample that follows it, so that the pair (c, c + 1) stays two
    // neighbouring floats.  Bit-identical, and no better: over all tiles the tap reads average 2.05 LDS cycles per 32-lane dword access
    // with either layout (a model of the bank mapping over the real sampling map: profiles/r06_warp_experiments.txt) - the footprint of
    // a tile is a slanted patch, not a stride - and the PMC conflict count rose (2.42e8 -> 2.73e8 with the four dword stores of the fill).
#ifndef WG_SWZ
#define WG_SWZ 0
#endif
    const int pp = WG_SWZ ? bp + (bp >> 5) + 1 : bp;   // physical pitch of a box row
    const int elems = pp * bh;
    const bool use_box = any && (elems <= WG_BOX_ELEMS);
    int off0[4], off1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (use_box) { const int cc = ixv[j] - mnx; off0[j] = (iyv[j] - mny) * pp + cc + (WG_SWZ ? (cc >> 5) : 0); off1[j] = off0[j] + pp; }
        else {
            int r0 = iyv[j] - 1, r1 = iyv[j];
            if (r0 < 0) r0 += rows; else if (r0 >= rows) r0 -= rows;
            if (r1 >= rows) r1 -= rows;
            off0[j] = r0 * (int)row_stride + ixv[j];
            off1[j] = r1 * (int)row_stride + ixv[j];
        }
    }
    // after the quad transpose lane 4g+i holds row 4*wv+i, columns 4g..4g+3 of the tile; one
    // ds_bpermute then moves that dword to lane 16*i+g so that 16 consecutive lanes store 64
    // consecutive bytes of one row
    const int qi = lane & 3;
    const int pull = ((lane & 15) << 2) | (lane >> 4);
    const int sy = y0 + wv * 4 + (lane >> 4), sx = bx_ * WG_TW + ((lane & 15) << 2);
    const bool sok = sy < W && sx + 3 < W;
    const uint32_t psel = 0x0c0c0000u | (uint32_t)qi | ((uint32_t)(4 + qi) << 8);
    uint8_t *dst = cart_u8 + (int64_t)sy * W + sx;
    if (!any) {                                       // beyond the maximum range: zeros (dark_stays_zero: the destination was
        if (sok && !dark_stays_zero)                  // zero-filled once and only this kernel writes it - nothing to do)
            for (int l = l0; l < l1; l++) *reinterpret_cast<uint32_t *>(dst + (int64_t)l * u8_lane_stride) = 0u;
        return;
    }
    if (!use_box) {                                   // rare tiles: direct gathers, one scan at a time
        for (int l = l0; l < l1; l++) {
            const uint8_t *p = pool + src_off(l);
            uint32_t pk = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = 0.f;
                if (in0[j]) {
                    const bool i1 = ixv[j] + 1 < cols;
                    const float s00 = code_to_f32(p[off0[j]]), s01 = i1 ? code_to_f32(p[off0[j] + 1]) : 0.f;
                    const float s10 = code_to_f32(p[off1[j]]), s11 = i1 ? code_to_f32(p[off1[j] + 1]) : 0.f;
                    v = __fmul_rn(s00, w00[j]);
                    v = __fadd_rn(v, __fmul_rn(s01, w01[j]));
                    v = __fadd_rn(v, __fmul_rn(s10, w10[j]));
                    v = __fadd_rn(v, __fmul_rn(s11, w11[j]));
                }
                pk |= quant_u8(v) << (8 * j);
            }
            const uint32_t v0 = quad_bcast<0>(pk), v1 = quad_bcast<1>(pk), v2 = quad_bcast<2>(pk), v3 = quad_bcast<3>(pk);
            const uint32_t t01 = __builtin_amdgcn_perm(v1, v0, psel), t23 = __builtin_amdgcn_perm(v3, v2, psel);
            const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute(pull << 2, (int)(t01 | (t23 << 16)));
            if (sok) *reinterpret_cast<uint32_t *>(dst + (int64_t)l * u8_lane_stride) = o;
        }
        return;
    }
#if WG_INT_BLEND
    // box path, exact blend (see above); branch-free: pixels outside the scan get zero weights and offset 0
    static_assert(WG_LB <= 32, "a list entry holds the scan in five bits");
    __shared__ uint16_t fixl[4][WG_FIX_CAP];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!in0[j]) { w00[j] = w01[j] = w10[j] = w11[j] = 0.f; off0[j] = off1[j] = 0; }
        w00[j] = __fmul_rn(w00[j], 65536.f); w01[j] = __fmul_rn(w01[j], 65536.f);       // 64 W: integers up to 2^16
        w10[j] = __fmul_rn(w10[j], 65536.f); w11[j] = __fmul_rn(w11[j], 65536.f);
    }
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    const bool chk = mnx + bp > cols;                 // some staged column lies past the last range bin (wave-uniform)
    const int per = max(1, min(WG_LB, WG_BOX_ELEMS / elems));
    uint16_t *fl = fixl[wvs];
    int nfix = 0;                                     // entries in this wave's list (wave-uniform)
    const uint64_t below = (1ull << lane) - 1ull;
    // the float32 chain for the listed pixels (entry = q | j << 5 | lane << 7: scan l0 + q, row 4 wv + j of the tile, column lane)
    auto fix_pixels = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0f70);           // vmcnt(0): the dword stores these bytes correct have left
        for (int i = lane; i < nfix; i += 64) {
            const uint32_t e = fl[i];
            const int q = e & 31, j = (e >> 5) & 3, ln = e >> 7;
            const int xx = bx_ * WG_TW + ln, yy = y0 + wvs * 4 + j, l = l0 + q;
            const uint32_t m = map[(int64_t)yy * W + xx];
            const int ix = m & 4095, iy = (m >> 12) & 1023;
            const float wx1 = __fmul_rn((float)((m >> 22) & 31), 1.f / 32.f), wx0 = __fsub_rn(1.f, wx1);
            const float wy1 = __fmul_rn((float)(m >> 27), 1.f / 32.f), wy0 = __fsub_rn(1.f, wy1);
            int r0 = iy - 1, r1 = iy;
            if (r0 < 0) r0 += rows; else if (r0 >= rows) r0 -= rows;
            if (r1 >= rows) r1 -= rows;
            const uint8_t *p = pool + ((lane_index ? (int64_t)lane_index[l] : (int64_t)l) * lane_stride + payload_off);
            const uint8_t *q0 = p + r0 * (int)row_stride + ix, *q1 = p + r1 * (int)row_stride + ix;
            const bool i1 = ix + 1 < cols;
            const float s00 = code_to_f32(q0[0]), s01 = i1 ? code_to_f32(q0[1]) : 0.f;
            const float s10 = code_to_f32(q1[0]), s11 = i1 ? code_to_f32(q1[1]) : 0.f;
            float v = __fmul_rn(s00, __fmul_rn(wy0, wx0));
            v = __fadd_rn(v, __fmul_rn(s01, __fmul_rn(wy0, wx1)));
            v = __fadd_rn(v, __fmul_rn(s10, __fmul_rn(wy1, wx0)));
            v = __fadd_rn(v, __fmul_rn(s11, __fmul_rn(wy1, wx1)));
            cart_u8[(int64_t)l * u8_lane_stride + (int64_t)yy * W + xx] = (uint8_t)quant_u8_unit(v);
        }
        nfix = 0;
    };
    for (int lb = l0; lb < l1; lb += per) {
        const int nq = min(per, l1 - lb);
        for (int qb = (WG_DIAG == 1 ? nq : 0); qb < nq;) {
            const int rem = nq - qb;
            const uint8_t *sp = pool + payload_off;
            float *bq = box + qb * elems;
#define WG_FILL(U_) { if (chk) box_fill<U_, true, true>(sp, lane_stride, lane_index, lb + qb, row_stride, rows, cols, mnx, mny, bw, bp, bh, elems, wvs, lane, bq, WG_SWZ ? pp : 0); \
                      else box_fill<U_, false, true>(sp, lane_stride, lane_index, lb + qb, row_stride, rows, cols, mnx, mny, bw, bp, bh, elems, wvs, lane, bq, WG_SWZ ? pp : 0); qb += U_; }
            if (rem >= 8 && WG_FILL_U >= 8) WG_FILL(8)
            else if (rem >= 4 && WG_FILL_U >= 4) WG_FILL(4)
            else if (rem >= 2 && WG_FILL_U >= 2) WG_FILL(2)
            else WG_FILL(1)
#undef WG_FILL
        }
        __syncthreads();
        uint8_t *dq = dst + (int64_t)lb * u8_lane_stride;          // this thread's dword in scan lb, advanced per scan
        for (int q = (WG_DIAG == 2 ? nq : 0); q < nq; q++) {
            const float *bx = box + q * elems;
            uint32_t sm[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x2 ta = f32x2{bx[off0[j]], bx[off0[j] + 1]}, tb = f32x2{bx[off1[j]], bx[off1[j] + 1]};   // one ds_read2_b32 each
                float e = __fmul_rn(ta.x, w00[j]);                                 // exact: integers below 2^24
                e = __fmaf_rn(ta.y, w01[j], e);
                e = __fmaf_rn(tb.x, w10[j], e);
                e = __fmaf_rn(tb.y, w11[j], e);
                sm[j] = (uint32_t)e;                                               // 64 * sum(W k)
            }
            uint32_t mn;
            asm("v_min3_u16 %0, %1, %2, %3" : "=v"(mn) : "v"(sm[0]), "v"(sm[1]), "v"(sm[2]));
            asm("v_min_u16 %0, %1, %2" : "=v"(mn) : "v"(mn), "v"(sm[3]));
            if (WG_DIAG != 3 && WG_DIAG != 4 && __builtin_expect(__ballot((mn & 0xffffu) == 0u) != 0ull, 0)) {
                // (rare) some pixel's sum has a zero low half: list the non-zero ones for the float chain
                if (nfix > WG_FIX_CAP - 256) fix_pixels();
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool f = (sm[j] & 0xffffu) == 0u && sm[j] != 0u;
                    const uint64_t bal = __ballot(f);
                    if (f) fl[nfix + __popcll(bal & below)] = (uint16_t)((lb - l0 + q) | (j << 5) | (lane << 7));
                    nfix += __popcll(bal);
                }
            }
            // byte 2 of each sum is the pixel
            const uint32_t pk = __builtin_amdgcn_perm(sm[1], sm[0], 0x0c0c0602u) | __builtin_amdgcn_perm(sm[3], sm[2], 0x06020c0cu);
            const uint32_t v0 = quad_bcast<0>(pk), v1 = quad_bcast<1>(pk), v2 = quad_bcast<2>(pk), v3 = quad_bcast<3>(pk);
            const uint32_t t01 = __builtin_amdgcn_perm(v1, v0, psel), t23 = __builtin_amdgcn_perm(v3, v2, psel);
            const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute(pull << 2, (int)(t01 | (t23 << 16)));
            if (sok) *reinterpret_cast<uint32_t *>(dq) = o;
            dq += u8_lane_stride;
        }
        __syncthreads();
    }
    if (WG_DIAG != 4 && nfix > 0) fix_pixels();
}
#else
    // box path, written branch-free: pixels outside the scan get zero weights and offset 0, loads
    // past the end of a row / past the last scan of the pass are clamped onto a valid duplicate
    f32x2 wa[4], wb[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!in0[j]) { w00[j] = w01[j] = w10[j] = w11[j] = 0.f; off0[j] = off1[j] = 0; }
        wa[j] = f32x2{w00[j], w01[j]};
        wb[j] = f32x2{w10[j], w11[j]};
    }
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    const bool chk = mnx + bp > cols;                 // some staged column lies past the last range bin (wave-uniform)
    const int per = max(1, min(WG_LB, WG_BOX_ELEMS / elems));
    for (int lb = l0; lb < l1; lb += per) {
        const int nq = min(per, l1 - lb);
        // wave w stages polar rows w, w+4, ... of the box for up to WG_FILL_U scans at a time:
        // their loads (same row, same columns) are issued before the first is consumed
        for (int qb = 0; qb < nq;) {
            const int rem = nq - qb;
            const uint8_t *sp = pool + payload_off;
            float *bq = box + qb * elems * (WG_CELL ? 2 : 1);
#define WG_FILL(U_) { if (chk) box_fill<U_, true, false, WG_CELL != 0>(sp, lane_stride, lane_index, lb + qb, row_stride, rows, cols, mnx, mny, bw, bp, bh, elems, wvs, lane, bq, WG_SWZ ? pp : 0); \
                      else box_fill<U_, false, false, WG_CELL != 0>(sp, lane_stride, lane_index, lb + qb, row_stride, rows, cols, mnx, mny, bw, bp, bh, elems, wvs, lane, bq, WG_SWZ ? pp : 0); qb += U_; }
            if (rem >= 8 && WG_FILL_U >= 8) WG_FILL(8)
            else if (rem >= 4 && WG_FILL_U >= 4) WG_FILL(4)
            else if (rem >= 2 && WG_FILL_U >= 2) WG_FILL(2)
            else WG_FILL(1)
#undef WG_FILL
        }
        __syncthreads();
        uint8_t *dq = dst + (int64_t)lb * u8_lane_stride;          // this thread's dword in scan lb, advanced per scan
        for (int q = 0; q < nq; q++) {
            float vq[4];
#if WG_CELL
            // cells: (s00, s10) and (s01, s11) are two neighbouring 8-byte cells - one ds_read2_b64 per pixel
            const f32x2 *cb = reinterpret_cast<const f32x2 *>(box) + q * elems;
            f32x2 ta[4], tb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { ta[j] = cb[off0[j]]; tb[j] = cb[off0[j] + 1]; }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x2 pa = ta[j] * f32x2{w00[j], w10[j]}, pb = tb[j] * f32x2{w01[j], w11[j]};      // (s00 w00, s10 w10), (s01 w01, s11 w11)
                vq[j] = __fadd_rn(__fadd_rn(__fadd_rn(pa.x, pb.x), pa.y), pb.y);
            }
#else
            const float *bx = box + q * elems;
            f32x2 ta[4], tb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                ta[j] = f32x2{bx[off0[j]], bx[off0[j] + 1]};      // (s00, s01): one ds_read2_b32 (4-byte aligned)
                tb[j] = f32x2{bx[off1[j]], bx[off1[j] + 1]};      // (s10, s11)
            }
            // the products are formed two at a time (v_pk_mul_f32: IEEE multiplies, no fusion); the
            // sums keep the reference order ((s00 w00 + s01 w01) + s10 w10) + s11 w11
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x2 pa = ta[j] * wa[j], pb = tb[j] * wb[j];
                vq[j] = __fadd_rn(__fadd_rn(__fadd_rn(pa.x, pa.y), pb.x), pb.y);
            }
#endif
            // (scalar multiplies here: a packed one makes the compiler pack the three adds above as well, at the
            // price of a dozen register moves)
            // (v_cvt_pk_u8_f32 would convert AND place the byte, but it ROUNDS to nearest where the reference's cast truncates - 11.0 ->
            // 10.6 ms with wrong bytes; fed the floor it is exact and SLOWER, 11.5 ms: round 4, measured and dropped)
            const uint32_t pk = quant_u8_unit(vq[0]) | (quant_u8_unit(vq[1]) << 8) | (quant_u8_unit(vq[2]) << 16) | (quant_u8_unit(vq[3]) << 24);
            const uint32_t v0 = quad_bcast<0>(pk), v1 = quad_bcast<1>(pk), v2 = quad_bcast<2>(pk), v3 = quad_bcast<3>(pk);
            const uint32_t t01 = __builtin_amdgcn_perm(v1, v0, psel), t23 = __builtin_amdgcn_perm(v3, v2, psel);
            const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute(pull << 2, (int)(t01 | (t23 << 16)));
            if (sok) *reinterpret_cast<uint32_t *>(dq) = o;
            dq += u8_lane_stride;
        }
        __syncthreads();
    }
}

#endif

// requires W % 4 == 0, u8_lane_stride % 4 == 0, rows * row_stride + cols < 2^31
hipError_t launch_warp_gather(hipStream_t st, const uint32_t *map, WarpSrc src, int B, int rows, int cols,
                              uint8_t *cart_u8, int64_t u8_lane_stride, bool dark_stays_zero)
{
    const int R = cols / 2, W = 2 * R;
    const int gx = (W + WG_TW - 1) / WG_TW, gy = (W + WG_TH - 1) / WG_TH, gz = (B + WG_LB - 1) / WG_LB;
    const int64_t total = (int64_t)gx * gy * gz;
    if (total > 0x7ffffff0) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)(((total + 7) >> 3) << 3);
    hipLaunchKernelGGL(warp_gather_kernel, dim3(blocks), dim3(256), 0, st, map, reinterpret_cast<const uint8_t *>(src.base),
                       src.lane_stride, src.row_stride, src.payload_off, src.lane_index, B, rows, cols, W, cart_u8,
                       u8_lane_stride, gx, gy, (int)total, dark_stays_zero ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_polar_to_cart(hipStream_t st, WarpSrc src, int B, int rows, int cols,
                                uint8_t *cart_u8, int64_t u8_lane_stride, float *cart_f32,
                                int64_t f32_lane_stride)
{
    const int R = cols / 2, W = 2 * R;
    dim3 grid((W / 4 + 255) / 256 + ((W % 4) ? 1 : 0), W, B), block(256);
    if (src.is_u8)
        hipLaunchKernelGGL(polar_to_cart_kernel<true>, grid, block, 0, st, src, rows, cols, R, cart_u8, u8_lane_stride, cart_f32, f32_lane_stride);
    else
        hipLaunchKernelGGL(polar_to_cart_kernel<false>, grid, block, 0, st, src, rows, cols, R, cart_u8, u8_lane_stride, cart_f32, f32_lane_stride);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void quantize_u8_kernel(const float *__restrict__ img, int64_t n, uint8_t *__restrict__ out)
{
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n && ((reinterpret_cast<uintptr_t>(img) & 15) == 0) && ((reinterpret_cast<uintptr_t>(out) & 3) == 0)) {
        float4 v = *reinterpret_cast<const float4 *>(img + i);
        uint32_t pk = quant_u8(v.x) | (quant_u8(v.y) << 8) | (quant_u8(v.z) << 16) | (quant_u8(v.w) << 24);
        *reinterpret_cast<uint32_t *>(out + i) = pk;
    } else
        for (int k = 0; k < 4 && i + k < n; k++) out[i + k] = (uint8_t)quant_u8(img[i + k]);
}

hipError_t launch_quantize_u8(hipStream_t st, const float *img, int64_t n, uint8_t *out)
{
    int64_t blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(quantize_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, st, img, n, out);
    return hipGetLastError();
}
