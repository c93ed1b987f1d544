// doh_common.h - device functions shared by the stage-level DoH kernels (doh.hip) and the engine's retrack (retrack.hip):
// the box-filter Hessian determinant of skimage's _hessian_matrix_det, operation by operation (oracle/c/doh.c).
// The integral image is read through an accessor (indices clipped at the image border like skimage's _integ).  hessian_box /
// hessian_det_pruned below are the box layouts and the pruning rule that retrack.hip's strip-march kernel (rt_det_strip_kernel)
// implements with its own LDS addressing: its comments refer to them.
#pragma once
#include "roam_internal.h"

__device__ __forceinline__ int clipi(int x, int lo, int hi) { return x > hi ? hi : (x < lo ? lo : x); }

struct DohGlobalAcc {
    const double *__restrict__ S; int H, W;
    __device__ __forceinline__ int cr(int r) const { return clipi(r, 0, H - 1); }
    __device__ __forceinline__ int cc(int c) const { return clipi(c, 0, W - 1); }
    __device__ __forceinline__ double at(int r, int c) const { return S[(int64_t)r * W + c]; }
};

template <typename ACC>
__device__ __forceinline__ double integ_acc(const ACC &a, int r, int c, int rl, int cl)
{
    r = a.cr(r);
    c = a.cc(c);
    const int r2 = a.cr(r + rl), c2 = a.cc(c + cl);
    const double ans = __dsub_rn(__dsub_rn(__dadd_rn(a.at(r, c), a.at(r2, c2)), a.at(r, c2)), a.at(r2, c));
    return fmax(ans, 0.0);       // skimage: max(0, ans); one v_max_f64 (a -0.0 result may come out as +0.0: no comparison or sum downstream can tell)
}

// determinant of the approximated Hessian at (r, c) for box size `size` = int(3 * sigma)
template <typename ACC>
__device__ __forceinline__ double hessian_det_acc(const ACC &a, int size, int r, int c)
{
    const int s2 = (size - 1) / 2, s3 = size / 3, w = size;
    const double w_i = __ddiv_rn(__ddiv_rn(1.0, (double)size), (double)size);
    const double tl = integ_acc(a, r - s3, c - s3, s3, s3);
    const double br = integ_acc(a, r + 1, c + 1, s3, s3);
    const double bl = integ_acc(a, r - s3, c + 1, s3, s3);
    const double tr = integ_acc(a, r + 1, c - s3, s3, s3);
    double dxy = __dsub_rn(__dsub_rn(__dadd_rn(bl, tr), tl), br);
    dxy = __dmul_rn(-dxy, w_i);
    double mid = integ_acc(a, r - s3 + 1, c - s2, 2 * s3 - 1, w);
    double side = integ_acc(a, r - s3 + 1, c - s3 / 2, 2 * s3 - 1, s3);
    double dxx = __dsub_rn(mid, __dmul_rn(3.0, side));
    dxx = __dmul_rn(-dxx, w_i);
    mid = integ_acc(a, r - s2, c - s3 + 1, w, 2 * s3 - 1);
    side = integ_acc(a, r - s3 / 2, c - s3 + 1, s3, 2 * s3 - 1);
    double dyy = __dsub_rn(mid, __dmul_rn(3.0, side));
    dyy = __dmul_rn(-dyy, w_i);
    return __dsub_rn(__dmul_rn(dxx, dyy), __dmul_rn(0.81, __dmul_rn(dxy, dxy)));
}

// One box of the determinant (k = 0..7: tl, br, bl, tr of dxy, then xx-mid, xx-side, yy-mid, yy-side): skimage's _integ, clipping included.
template <int SIZE, int K, typename ACC>
__device__ __forceinline__ double hessian_box(const ACC &a, int r, int c)
{
    constexpr int s2 = (SIZE - 1) / 2, s3 = SIZE / 3, w = SIZE;
    constexpr int B[8][4] = {{-s3, -s3, s3, s3}, {1, 1, s3, s3}, {-s3, 1, s3, s3}, {1, -s3, s3, s3},
                             {-s3 + 1, -s2, 2 * s3 - 1, w}, {-s3 + 1, -(s3 / 2), 2 * s3 - 1, s3},
                             {-s2, -s3 + 1, w, 2 * s3 - 1}, {-(s3 / 2), -s3 + 1, s3, 2 * s3 - 1}};
    const int r0 = a.cr(r + B[K][0]), c0 = a.cc(c + B[K][1]), r1 = a.cr(r0 + B[K][2]), c1 = a.cc(c0 + B[K][3]);
    return fmax(__dsub_rn(__dsub_rn(__dadd_rn(a.at(r0, c0), a.at(r1, c1)), a.at(r0, c1)), a.at(r1, c0)), 0.0);
}

// The determinant for a consumer that only asks "which pixels exceed `thr` and are local maxima": det = dxx*dyy - 0.81*dxy^2 in
// round-to-nearest is never above P = dxx*dyy (the subtrahend is >= 0 and rounding is monotone), so where P <= thr the pixel can
// neither pass the threshold nor exceed a neighbour that does, and the four dxy boxes (16 of the 32 corner reads) are skipped; P
// itself is returned there.  Where P > thr the result is hessian_det_acc's, operation by operation.  On radar frames (real and
// synthetic) more than 99 % of the pixels stop at P.
template <int SIZE, typename ACC>
__device__ __forceinline__ double hessian_det_pruned(const ACC &a, int r, int c, double thr)
{
    const double w_i = __ddiv_rn(__ddiv_rn(1.0, (double)SIZE), (double)SIZE);
    double dxx = __dsub_rn(hessian_box<SIZE, 4>(a, r, c), __dmul_rn(3.0, hessian_box<SIZE, 5>(a, r, c)));
    dxx = __dmul_rn(-dxx, w_i);
    double dyy = __dsub_rn(hessian_box<SIZE, 6>(a, r, c), __dmul_rn(3.0, hessian_box<SIZE, 7>(a, r, c)));
    dyy = __dmul_rn(-dyy, w_i);
    double det = __dmul_rn(dxx, dyy);
    if (det > thr) {
        const double tl = hessian_box<SIZE, 0>(a, r, c), br = hessian_box<SIZE, 1>(a, r, c);
        const double bl = hessian_box<SIZE, 2>(a, r, c), tr = hessian_box<SIZE, 3>(a, r, c);
        double dxy = __dsub_rn(__dsub_rn(__dadd_rn(bl, tr), tl), br);
        dxy = __dmul_rn(-dxy, w_i);
        det = __dsub_rn(det, __dmul_rn(0.81, __dmul_rn(dxy, dxy)));
    }
    return det;
}

__device__ __forceinline__ double hessian_det_at(const double *__restrict__ S, int H, int W, int size, int r, int c)
{
    const DohGlobalAcc a = {S, H, W};
    return hessian_det_acc(a, size, r, c);
}
