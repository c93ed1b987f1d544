// doh_common.h - device functions shared by the stage-level DoH kernels (doh.hip) and the engine's retrack (retrack.hip):
// the box-filter Hessian determinant of skimage's _hessian_matrix_det, operation by operation (oracle/c/doh.c).
#pragma once
#include "roam_internal.h"

__device__ __forceinline__ int clipi(int x, int lo, int hi) { return x > hi ? hi : (x < lo ? lo : x); }
__device__ __forceinline__ double integ(const double *__restrict__ S, int H, int W, int r, int c, int rl, int cl)
{
    r = clipi(r, 0, H - 1);
    c = clipi(c, 0, W - 1);
    const int r2 = clipi(r + rl, 0, H - 1), c2 = clipi(c + cl, 0, W - 1);
    const double ans = __dsub_rn(__dsub_rn(__dadd_rn(S[(int64_t)r * W + c], S[(int64_t)r2 * W + c2]), S[(int64_t)r * W + c2]),
                                 S[(int64_t)r2 * W + c]);
    return ans < 0 ? 0 : ans;
}

// determinant of the approximated Hessian at (r, c) for box size `size` = int(3 * sigma)
__device__ __forceinline__ double hessian_det_at(const double *__restrict__ S, int H, int W, int size, int r, int c)
{
    const int s2 = (size - 1) / 2, s3 = size / 3, w = size;
    const double w_i = __ddiv_rn(__ddiv_rn(1.0, (double)size), (double)size);
    const double tl = integ(S, H, W, r - s3, c - s3, s3, s3);
    const double br = integ(S, H, W, r + 1, c + 1, s3, s3);
    const double bl = integ(S, H, W, r - s3, c + 1, s3, s3);
    const double tr = integ(S, H, W, r + 1, c - s3, s3, s3);
    double dxy = __dsub_rn(__dsub_rn(__dadd_rn(bl, tr), tl), br);
    dxy = __dmul_rn(-dxy, w_i);
    double mid = integ(S, H, W, r - s3 + 1, c - s2, 2 * s3 - 1, w);
    double side = integ(S, H, W, r - s3 + 1, c - s3 / 2, 2 * s3 - 1, s3);
    double dxx = __dsub_rn(mid, __dmul_rn(3.0, side));
    dxx = __dmul_rn(-dxx, w_i);
    mid = integ(S, H, W, r - s2, c - s3 + 1, w, 2 * s3 - 1);
    side = integ(S, H, W, r - s3 / 2, c - s3 + 1, s3, 2 * s3 - 1);
    double dyy = __dsub_rn(mid, __dmul_rn(3.0, side));
    dyy = __dmul_rn(-dyy, w_i);
    return __dsub_rn(__dmul_rn(dxx, dyy), __dmul_rn(0.81, __dmul_rn(dxy, dxy)));
}
