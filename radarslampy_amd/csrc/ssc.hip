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

#define SSC_BITMAP_BYTES 65536
// the batched (engine) launch asks for a quarter of that: four square widths out of five of a 2024 x 2024 image need less than 1 KB of
// cell bitmap, and with 64 KB only two problems fit a CU (the pairwise form takes over below a width of ~11 px - same result)
#define SSC_BATCH_BITMAP_BYTES 16384

__device__ __forceinline__ void ssc_body(const double *__restrict__ kp, int B, int num_ret, double tol, int cols, int rows,
                                         int32_t *__restrict__ work, int32_t *__restrict__ sel, int32_t *__restrict__ n_sel,
                                         uint32_t *bitmap, int bitmap_bytes)
{
    const int lane = threadIdx.x;
    int32_t *resA = work, *resB = work + B, *accr = work + 2 * (size_t)B, *accq = work + 3 * (size_t)B;
    const double exp1 = (double)rows + cols + 2 * num_ret;
    const double exp2 = 4.0 * cols + 4.0 * num_ret + 4.0 * rows * num_ret + (double)rows * rows +
                        (double)cols * cols - 2.0 * rows * cols + 4.0 * rows * (double)cols * num_ret;
    const double exp3 = sqrt(exp2);
    const double exp4 = num_ret - 1;
    const double sol1 = -rint((exp1 + exp3) / exp4);
    const double sol2 = -rint((exp1 - exp3) / exp4);
    double high = sol1 > sol2 ? sol1 : sol2;
    double low = floor(sqrt((double)B / num_ret));
    double prev_width = -1;
    const int k_min = (int)rint(num_ret - num_ret * tol);
    const int k_max = (int)rint(num_ret + num_ret * tol);
    int32_t *cur = resA, *prev = resB;
    int nprev = 0;
    const int32_t *final_list = prev;
    int nfinal = 0;
    for (;;) {
        const double width = low + (high - low) / 2;
        if (width == prev_width || low > high || width == 0) { final_list = prev; nfinal = nprev; break; }
        const double c = width / 2;
        const int w = (int)floor(width / c);
        const double ncd = floor(cols / c), nrd = floor(rows / c);
        const bool grid_mode = (ncd + 1) * (nrd + 1) <= (double)bitmap_bytes * 8;
        const int ncols = (int)ncd + 1, nrows = (int)nrd + 1;
        if (grid_mode) {
            const int nwords = (ncols * nrows + 31) / 32;
            for (int i = lane; i < nwords; i += 64) bitmap[i] = 0;
        }
        __syncthreads();
        int nres = 0;
        for (int base = 0; base < B; base += 64) {
            const int i = base + lane;
            int r = 0, q = 0;
            bool cand = false;
            if (i < B) {
                r = (int)floor(kp[3 * (size_t)i] / c);
                q = (int)floor(kp[3 * (size_t)i + 1] / c);
                if (grid_mode) {
                    r = min(max(r, 0), nrows - 1); q = min(max(q, 0), ncols - 1);
                    const int idx = r * ncols + q;
                    cand = !((bitmap[idx >> 5] >> (idx & 31)) & 1u);
                } else {
                    cand = true;
                    for (int a = 0; a < nres; a++) {
                        int dr = accr[a] - r, dq = accq[a] - q;
                        dr = dr < 0 ? -dr : dr; dq = dq < 0 ? -dq : dq;
                        if (dr <= w && dq <= w) { cand = false; break; }
                    }
                }
            }
            uint64_t alive = __ballot(cand);
            uint64_t accepted = 0;
            while (alive) {
                const int j = __ffsll((long long)alive) - 1;
                accepted |= 1ull << j;
                alive &= ~(1ull << j);
                const int rj = __builtin_amdgcn_readlane(r, j), qj = __builtin_amdgcn_readlane(q, j);   // (j is wave-uniform: v_readlane, not ds_bpermute)
                int dr = r - rj, dq = q - qj;
                dr = dr < 0 ? -dr : dr; dq = dq < 0 ? -dq : dq;
                const uint64_t kill = __ballot(lane > j && dr <= w && dq <= w);
                alive &= ~kill;
            }
            const bool mine = (accepted >> lane) & 1ull;
            const int pos = nres + __popcll(accepted & ((1ull << lane) - 1ull));
            if (mine) {
                cur[pos] = i;
                if (grid_mode) {
                    const int r0 = max(r - w, 0), r1 = min(r + w, nrows - 1);
                    const int q0 = max(q - w, 0), q1 = min(q + w, ncols - 1);
                    for (int rr = r0; rr <= r1; rr++)
                        for (int qq = q0; qq <= q1; qq++) {
                            const int idx = rr * ncols + qq;
                            atomicOr(&bitmap[idx >> 5], 1u << (idx & 31));
                        }
                } else { accr[pos] = r; accq[pos] = q; }
            }
            nres += __popcll(accepted);
            __syncthreads();
        }
        if (nres >= k_min && nres <= k_max) { final_list = cur; nfinal = nres; break; }
        else if (nres < k_min) high = width - 1;
        else low = width + 1;
        prev_width = width;
        int32_t *tmp = cur; cur = prev; prev = tmp;
        nprev = nres;
    }
    __syncthreads();
    for (int i = lane; i < nfinal; i += 64) sel[i] = final_list[i];
    if (lane == 0) *n_sel = nfinal;
}

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
