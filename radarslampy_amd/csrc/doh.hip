// Determinant-of-Hessian blob candidates (a4).
//
// Replaces the image-scale part of skimage.feature.blob_doh as called by
// getFeatures.getBlobsFromCart (reference getFeatures.py:22-53, parameters :13-18).  Algorithm
// statement and its (un)pinned status: oracle/c/doh.c.  Stages, all float64 like the reference:
//   integ_cols / integ_rows : integral image = cumsum over rows then columns.  Both passes keep
//       NumPy's SEQUENTIAL summation order (one thread per column, then one lane per row with a
//       64x64 LDS transpose tile so that global traffic stays coalesced) => bit-identical sums.
//   hessian_det : one thread per pixel, 8 clipped box sums (32 L2-resident f64 taps), per sigma.
//   maxima (count + write) : 3x3x3 strict-threshold local maxima, compacted in C (row, col,
//       sigma) order with ordered block scans, exactly the order np.nonzero would produce.
// Ordering by response, sigma lookup and the overlap pruning are host-side bookkeeping
// (radarslampy_amd/getFeatures.py), as in scikit-image itself.
#include "roam_internal.h"
#include "doh_common.h"

#define DOH_MAX_LAYERS 8
struct DohLayers { const double *p[DOH_MAX_LAYERS]; int nl; };

__global__ __launch_bounds__(256) void integ_cols_kernel(const float *__restrict__ img, int H, int W, double *__restrict__ S)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= W) return;
    double acc = 0;
    for (int r = 0; r < H; r++) { acc = __dadd_rn(acc, (double)img[(int64_t)r * W + c]); S[(int64_t)r * W + c] = acc; }
}

__global__ __launch_bounds__(64) void integ_rows_kernel(double *__restrict__ S, int H, int W)
{
    __shared__ double tile[64][65];
    const int lane = threadIdx.x, r0 = blockIdx.x * 64;
    double acc = 0;
    for (int c0 = 0; c0 < W; c0 += 64) {
        for (int k = 0; k < 64; k++) {
            const int r = r0 + k, c = c0 + lane;
            tile[k][lane] = (r < H && c < W) ? S[(int64_t)r * W + c] : 0.0;
        }
        __syncthreads();
        const int nc = min(64, W - c0);
        for (int j = 0; j < nc; j++) { acc = __dadd_rn(acc, tile[lane][j]); tile[lane][j] = acc; }
        __syncthreads();
        for (int k = 0; k < 64; k++) {
            const int r = r0 + k, c = c0 + lane;
            if (r < H && c < W) S[(int64_t)r * W + c] = tile[k][lane];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void hessian_det_kernel(const double *__restrict__ S, int H, int W, int size,
                                                          double *__restrict__ out)
{
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    out[(int64_t)r * W + c] = hessian_det_at(S, H, W, size, r, c);
}

__device__ __forceinline__ bool is_max(const DohLayers &L, int H, int W, int r, int c, int s, double thr, double *val)
{
    if (!L.p[s]) return false;
    const double v = L.p[s][(int64_t)r * W + c];
    if (!(v > thr)) return false;
    for (int ds = -1; ds <= 1; ds++) {
        const int ss = s + ds;
        if (ss < 0 || ss >= L.nl) continue;            // outside the cube: 0 < thr < v
        if (!L.p[ss]) continue;                        // NaN layer: ignored
        for (int dr = -1; dr <= 1; dr++) {
            const int rr = r + dr;
            if (rr < 0 || rr >= H) continue;
            for (int dc = -1; dc <= 1; dc++) {
                const int cc = c + dc;
                if (cc < 0 || cc >= W) continue;
                if (L.p[ss][(int64_t)rr * W + cc] > v) return false;
            }
        }
    }
    *val = v;
    return true;
}

__device__ __forceinline__ int doh_blk_scan(int v, int *sh, int *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int n = __shfl_up(inc, d);
        if (lane >= d) inc += n;
    }
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) { int s = sh[i]; if (i < w) base += s; tot += s; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// WRITE=false: row_count[r] = number of maxima in row r.  WRITE=true: emit at row_off[r] + position.
template <bool WRITE>
__global__ __launch_bounds__(256) void doh_maxima_kernel(DohLayers L, int H, int W, double thr, int32_t *__restrict__ row_count,
                                                         const int32_t *__restrict__ row_off, int32_t *__restrict__ out_rcs,
                                                         double *__restrict__ out_val, int cap)
{
    __shared__ int sh[8];
    const int r = blockIdx.x, t = threadIdx.x;
    const int items = (W + 255) / 256;
    const int lo = t * items, hi = min(lo + items, W);
    int cnt = 0;
    double v;
    for (int c = lo; c < hi; c++)
        for (int s = 0; s < L.nl; s++) cnt += is_max(L, H, W, r, c, s, thr, &v) ? 1 : 0;
    int total;
    int pos = doh_blk_scan(cnt, sh, &total);
    if (!WRITE) { if (t == 0) row_count[r] = total; return; }
    pos += row_off[r];
    for (int c = lo; c < hi; c++)
        for (int s = 0; s < L.nl; s++)
            if (is_max(L, H, W, r, c, s, thr, &v)) {
                if (pos < cap) { out_rcs[3 * (int64_t)pos] = r; out_rcs[3 * (int64_t)pos + 1] = c; out_rcs[3 * (int64_t)pos + 2] = s; out_val[pos] = v; }
                pos++;
            }
}

__global__ __launch_bounds__(256) void doh_row_scan_kernel(const int32_t *__restrict__ row_count, int H, int32_t *__restrict__ row_off,
                                                           int32_t *__restrict__ total_out)
{
    __shared__ int sh[8];
    const int t = threadIdx.x;
    const int items = (H + 255) / 256;
    const int lo = t * items, hi = min(lo + items, H);
    int c = 0;
    for (int r = lo; r < hi; r++) c += row_count[r];
    int total;
    int pos = doh_blk_scan(c, sh, &total);
    for (int r = lo; r < hi; r++) { row_off[r] = pos; pos += row_count[r]; }
    if (t == 0) *total_out = total;
}

// device-side core: dimg (w x h f32, device) -> maxima copied to the host arrays
static int32_t doh_maxima_device(roam_ctx *ctx, const float *dimg, int32_t w, int32_t h, const double *sigmas,
                                 int32_t num_sigma, double threshold, int32_t *out_rcs, double *out_val,
                                 int32_t cap, int32_t *n_out)
{
    hipStream_t st = ctx->stream;
    const size_t npx = (size_t)w * h;
    double *S = (double *)roam_scratch(ctx, S_TMP0, sizeof(double) * npx);
    int32_t *rowc = (int32_t *)roam_scratch(ctx, S_TMP1, sizeof(int32_t) * (2 * (size_t)h + 1));
    int32_t *drcs = (int32_t *)roam_scratch(ctx, S_OUT0, sizeof(int32_t) * 3 * (size_t)(cap > 0 ? cap : 1));
    double *dval = (double *)roam_scratch(ctx, S_OUT1, sizeof(double) * (size_t)(cap > 0 ? cap : 1));
    if (!S || !rowc || !drcs || !dval) return ROAM_E_HIP;
    static const int slots[DOH_MAX_LAYERS] = {S_TMP2, S_TMP3, S_TMP4, S_TMP5, S_TMP6, S_TMP7, S_IN2, S_IN3};
    DohLayers L;
    L.nl = num_sigma;
    hipLaunchKernelGGL(integ_cols_kernel, dim3((w + 255) / 256), dim3(256), 0, st, dimg, h, w, S);
    hipLaunchKernelGGL(integ_rows_kernel, dim3((h + 63) / 64), dim3(64), 0, st, S, h, w);
    HIP_TRY(ctx, hipGetLastError());
    for (int s = 0; s < num_sigma; s++) {
        const int size = (int)(3 * sigmas[s]);
        L.p[s] = nullptr;
        if (size <= 0) continue;                       // degenerate layer (all NaN in the reference): ignored
        double *lay = (double *)roam_scratch(ctx, slots[s], sizeof(double) * npx);
        if (!lay) return ROAM_E_HIP;
        hipLaunchKernelGGL(hessian_det_kernel, dim3((w + 255) / 256, h), dim3(256), 0, st, S, h, w, size, lay);
        HIP_TRY(ctx, hipGetLastError());
        L.p[s] = lay;
    }
    int32_t *rowo = rowc + h, *dtotal = rowc + 2 * h;
    hipLaunchKernelGGL(doh_maxima_kernel<false>, dim3(h), dim3(256), 0, st, L, h, w, threshold, rowc, rowo, drcs, dval, cap);
    hipLaunchKernelGGL(doh_row_scan_kernel, dim3(1), dim3(256), 0, st, rowc, h, rowo, dtotal);
    hipLaunchKernelGGL(doh_maxima_kernel<true>, dim3(h), dim3(256), 0, st, L, h, w, threshold, rowc, rowo, drcs, dval, cap);
    HIP_TRY(ctx, hipGetLastError());
    int32_t n = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n, dtotal, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    *n_out = n;
    const int m = n < cap ? n : cap;
    if (m > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(out_rcs, drcs, sizeof(int32_t) * 3 * (size_t)m, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(out_val, dval, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
    }
    if (n > cap) { ROAM_SET_ERR(ctx, "doh: %d maxima, capacity %d", n, cap); return ROAM_E_CAPACITY; }
    return ROAM_OK;
}

extern "C" int32_t roam_doh_maxima(roam_ctx *ctx, const float *img, int32_t w, int32_t h, const double *sigmas,
                                   int32_t num_sigma, double threshold, int32_t *out_rcs, double *out_val,
                                   int32_t cap, int32_t *n_out)
{
    if (!ctx) return ROAM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ARG_CHECK(ctx, img && sigmas && out_rcs && out_val && n_out && w >= 3 && h >= 3 && num_sigma >= 1 &&
                       num_sigma <= DOH_MAX_LAYERS && cap >= 0 && threshold >= 0);
    const size_t npx = (size_t)w * h;
    float *dimg = (float *)roam_scratch(ctx, S_IN0, sizeof(float) * npx);
    if (!dimg) return ROAM_E_HIP;
    HIP_TRY(ctx, hipMemcpyAsync(dimg, img, sizeof(float) * npx, hipMemcpyHostToDevice, ctx->stream));
    return doh_maxima_device(ctx, dimg, w, h, sigmas, num_sigma, threshold, out_rcs, out_val, cap, n_out);
}

// engine variant: the image is the float32 Cartesian warp of a raw record that is already resident
// in HBM (`rec` = device pointer to rows x stride u8), i.e. appendNewFeatures(currImgCart, ...) of
// RawROAMSystem.py:264 without any host round trip of image data.
int32_t roam_doh_maxima_record_device(roam_ctx *ctx, const uint8_t *rec, int rows, int64_t stride, int payload_off,
                                      int clip, const double *sigmas, int32_t num_sigma, double threshold,
                                      int32_t *out_rcs, double *out_val, int32_t cap, int32_t *n_out)
{
    ARG_CHECK(ctx, rec && sigmas && out_rcs && out_val && n_out && num_sigma >= 1 && num_sigma <= DOH_MAX_LAYERS && cap >= 0);
    const int W = 2 * (clip / 2);
    float *dimg = (float *)roam_scratch(ctx, S_IN0, sizeof(float) * (size_t)W * W);
    if (!dimg) return ROAM_E_HIP;
    WarpSrc ws = {rec, 0, stride, payload_off, 1, nullptr};
    HIP_TRY(ctx, launch_polar_to_cart(ctx->stream, ws, 1, rows, clip, nullptr, 0, dimg, (int64_t)W * W));
    return doh_maxima_device(ctx, dimg, W, W, sigmas, num_sigma, threshold, out_rcs, out_val, cap, n_out);
}
