// Adaptive non-maximal suppression, "suppression via square covering" (a5).
//
// Replaces ANMS.ssc (reference ANMS.py:5-102): binary search over the square width; per
// width a greedy pass over the keypoints in priority order.  One wavefront per call: the
// 64 lanes test 64 consecutive keypoints against the covered-cell bitmap (LDS) with one
// load each, conflicts inside the chunk are resolved in priority order with ballots and
// lane broadcasts (wave-uniform loop over the surviving lanes), accepted lanes cover their
// 5x5 cell block with LDS atomic ORs.  When the grid would not fit the LDS bitmap (very
// small widths, i.e. B < ~400) the equivalent pairwise test against the accepted list is
// used (see oracle/c/ssc.c for the equivalence argument).  Binary-search arithmetic is
// float64 and identical to the reference (Python round == rint, half-to-even).
#include "roam_internal.h"

#include "ssc_body.inc"

__global__ __launch_bounds__(64) void ssc_kernel(const double *__restrict__ kp, int B, int num_ret, double tol,
                                                 int cols, int rows, int32_t *__restrict__ work,
                                                 int32_t *__restrict__ sel, int32_t *__restrict__ n_sel)
{
    extern __shared__ uint32_t bitmap[];
    ssc_body(kp, B, num_ret, tol, cols, rows, work, sel, n_sel, bitmap, SSC_BITMAP_BYTES);
}

// batched: problem p = blockIdx.x (skipped when first + p >= *n_active): kp + p * kp_stride (rows of 3 doubles), count[p]
// keypoints, work + p * 4 * kp_cap, sel + p * kp_cap
__global__ __launch_bounds__(64) void ssc_batch_kernel(const double *__restrict__ kp, int64_t kp_stride, const int32_t *__restrict__ count,
                                                       int kp_cap, int num_ret, double tol, int cols, int rows,
                                                       int32_t *__restrict__ work, int32_t *__restrict__ sel,
                                                       int32_t *__restrict__ n_sel, const int32_t *__restrict__ n_active, int first)
{
    extern __shared__ uint32_t bitmap[];
    const int p = blockIdx.x;
    if (first + p >= *n_active) return;
    const int B = min(count[p], kp_cap);
    if (B <= 0) { if (threadIdx.x == 0) n_sel[p] = 0; return; }
    ssc_body(kp + (int64_t)p * kp_stride, B, num_ret, tol, cols, rows, work + (int64_t)p * 4 * kp_cap, sel + (int64_t)p * kp_cap,
             n_sel + p, bitmap, SSC_BATCH_BITMAP_BYTES);
}

hipError_t launch_ssc_batch(hipStream_t st, const double *kp, int64_t kp_stride, const int32_t *count, int kp_cap, int P,
                            int num_ret, double tol, int cols, int rows, int32_t *work, int32_t *sel, int32_t *n_sel,
                            const int32_t *n_active, int first)
{
    hipLaunchKernelGGL(ssc_batch_kernel, dim3(P), dim3(64), SSC_BATCH_BITMAP_BYTES, st, kp, kp_stride, count, kp_cap, num_ret, tol, cols,
                       rows, work, sel, n_sel, n_active, first);
    return hipGetLastError();
}

hipError_t launch_ssc(hipStream_t st, const double *kp, int B, int num_ret, double tol, int cols,
                      int rows, int32_t *work, int32_t *sel, int32_t *n_sel)
{
    hipLaunchKernelGGL(ssc_kernel, dim3(1), dim3(64), SSC_BITMAP_BYTES, st, kp, B, num_ret, tol, cols, rows, work, sel, n_sel);
    return hipGetLastError();
}

extern "C" int32_t roam_ssc(roam_ctx *ctx, const double *kp, int32_t B, int32_t num_ret, double tol,
                            int32_t cols, int32_t rows, int32_t *sel_out, int32_t *n_sel)
{
    if (!ctx) return ROAM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ARG_CHECK(ctx, B >= 0 && num_ret >= 2 && cols > 0 && rows > 0 && sel_out && n_sel && (B == 0 || kp));
    if (B == 0) { *n_sel = 0; return ROAM_OK; }
    double *dkp = (double *)roam_scratch(ctx, S_IN0, sizeof(double) * 3 * (size_t)B);
    int32_t *dwork = (int32_t *)roam_scratch(ctx, S_TMP0, sizeof(int32_t) * 4 * (size_t)B);
    int32_t *dsel = (int32_t *)roam_scratch(ctx, S_OUT0, sizeof(int32_t) * ((size_t)B + 1));
    if (!dkp || !dwork || !dsel) return ROAM_E_HIP;
    HIP_TRY(ctx, hipMemcpyAsync(dkp, kp, sizeof(double) * 3 * (size_t)B, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, launch_ssc(ctx->stream, dkp, B, num_ret, tol, cols, rows, dwork, dsel + 1, dsel));
    int32_t n = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n, dsel, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (n > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(sel_out, dsel + 1, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    *n_sel = n;
    return ROAM_OK;
}
