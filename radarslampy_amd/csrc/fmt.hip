// Fourier-Mellin rotation prior: FMT.getRotationUsingFMT (reference FMT.py:36-90), the estimate
// Tracker.track computes first (Tracker.py:62-63) and returns in slot 3.
//   clip the polar image to int(87.5 / 0.0864) = 1012 range bins, cv2.resize to W // 10 columns (INTER_LINEAR),
//   convertPolarImgToLogPolar (parseData.py:138-160): inverse linear warpPolar to a 2R x 2R Cartesian image (R = the
//   downsampled width), forward semilog warpPolar to OpenCV's default size (round(R) x round(pi R)),
//   cv2.phaseCorrelate with a Hanning window: DFT of both images at the optimal size (320 x 108), normalised cross-power
//   spectrum, inverse DFT, fftshift, peak, 5 x 5 weighted centroid -> (d rho, d phi), response;
//   angle = wrap(-d phi * 2 pi / max(H, W)), scale = log_base ** d rho.
// The images are tiny (400 x 101 -> 202 x 202 -> 317 x 101 -> 320 x 108), so the transforms are direct DFTs with float64
// accumulation (14 M complex multiply-adds per image, tens of microseconds) instead of an FFT library; twiddles come from
// sincospi on (k mod n) / n.  OpenCV is absent and the reference keeps no output of this path: parity is against the
// oracle's numpy restatement (tolerance 1e-5 rad), PARITY UNPINNED like the oracle itself.
#include "roam_internal.h"

#define FMT_PI 3.14159265358979323846

__device__ __forceinline__ float fmt_fast_atan2_deg(float y, float x)
{
    const float sc = (float)(180 / FMT_PI);
    const float p1 = __fmul_rn(0.9997878412794807f, sc), p3 = __fmul_rn(-0.3258083974640975f, sc);
    const float p5 = __fmul_rn(0.1555786518463281f, sc), p7 = __fmul_rn(-0.04432655554792128f, sc);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, (float)2.220446049250313e-16)); c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, (float)2.220446049250313e-16)); c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// cv2.resize(img[:, :clip], (nw, rows)), INTER_LINEAR, float32: two taps per output column, rows untouched
__global__ void fmt_resize_kernel(const float *__restrict__ polar, int rows, int64_t stride, int clip, int nw, float *__restrict__ out)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (dx >= nw) return;
    const double scale = 1.0 / ((double)nw / (double)clip);
    float fx = (float)(((double)dx + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx = __fsub_rn(fx, (float)sx);
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= clip - 1) { fx = 0.f; sx = clip - 1; }
    const float *p = polar + (int64_t)r * stride;
    const float s0 = p[sx], s1 = p[min(sx + 1, clip - 1)];
    out[(int64_t)r * nw + dx] = __fadd_rn(__fmul_rn(s0, __fsub_rn(1.f, fx)), __fmul_rn(s1, fx));
}

__device__ __forceinline__ float fmt_polar_tap(const float *p, int rows, int cols, int py, int px)
{
    if (px < 0 || px >= cols || py < 0 || py >= rows + 2) return 0.f;
    int r = py - 1;
    if (r < 0) r += rows; else if (r >= rows) r -= rows;
    return p[(int64_t)r * cols + px];
}

// inverse linear warpPolar, maxRadius = cols, centre (cols, cols): (rows x cols) -> (2 cols x 2 cols)
__global__ void fmt_cart_kernel(const float *__restrict__ polar, int rows, int cols, float *__restrict__ cart)
{
    const int W = 2 * cols;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const double Kangle = 2 * FMT_PI / (double)rows, Kmag = (double)cols / (double)cols;
    const float fx = __fsub_rn((float)x, (float)cols), fy = __fsub_rn((float)y, (float)cols);
    const float mag = rn_sqrtf(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)));
    const float ang = __fmul_rn(fmt_fast_atan2_deg(fy, fx), (float)(FMT_PI / 180.0));
    const float mx = (float)__ddiv_rn((double)mag, Kmag), my = __fadd_rn((float)__ddiv_rn((double)ang, Kangle), 1.f);
    const int sx = __float2int_rn(__fmul_rn(mx, 32.f)), sy = __float2int_rn(__fmul_rn(my, 32.f));
    const int ix = sx >> 5, iy = sy >> 5;
    const float wx1 = __fmul_rn((float)(sx & 31), 1.f / 32.f), wx0 = __fsub_rn(1.f, wx1);
    const float wy1 = __fmul_rn((float)(sy & 31), 1.f / 32.f), wy0 = __fsub_rn(1.f, wy1);
    float v = __fmul_rn(fmt_polar_tap(polar, rows, cols, iy, ix), __fmul_rn(wy0, wx0));
    v = __fadd_rn(v, __fmul_rn(fmt_polar_tap(polar, rows, cols, iy, ix + 1), __fmul_rn(wy0, wx1)));
    v = __fadd_rn(v, __fmul_rn(fmt_polar_tap(polar, rows, cols, iy + 1, ix), __fmul_rn(wy1, wx0)));
    v = __fadd_rn(v, __fmul_rn(fmt_polar_tap(polar, rows, cols, iy + 1, ix + 1), __fmul_rn(wy1, wx1)));
    cart[(int64_t)y * W + x] = v;
}

__device__ __forceinline__ float fmt_cart_tap(const float *c, int W, int y, int x) { return (x < 0 || x >= W || y < 0 || y >= W) ? 0.f : c[(int64_t)y * W + x]; }

// forward semilog warpPolar of the W x W Cartesian image (centre, maxRadius = W / 2) to dw x dh, multiplied by the Hanning
// window sqrt(wr * wc) of cv2.createHanningWindow and zero-padded into the M x N DFT input (float64)
__global__ void fmt_logpolar_window_kernel(const float *__restrict__ cart, int W, int dw, int dh, int M, int N, double *__restrict__ out)
{
    const int rho = blockIdx.x * blockDim.x + threadIdx.x, phi = blockIdx.y;
    if (rho >= N) return;
    double val = 0.0;
    if (rho < dw && phi < dh) {
        const double R = (double)W / 2.0;
        const double Kangle = 2 * FMT_PI / (double)dh, Kmag = log(R) / (double)dw;
        const double KKy = Kangle * (double)phi, cp = cos(KKy), sp = sin(KKy);
        const float br = (float)(exp((double)rho * Kmag) - 1.0);
        const float mx = (float)((double)br * cp + R), my = (float)((double)br * sp + R);
        const int sx = __float2int_rn(__fmul_rn(mx, 32.f)), sy = __float2int_rn(__fmul_rn(my, 32.f));
        const int ix = sx >> 5, iy = sy >> 5;
        const float wx1 = __fmul_rn((float)(sx & 31), 1.f / 32.f), wx0 = __fsub_rn(1.f, wx1);
        const float wy1 = __fmul_rn((float)(sy & 31), 1.f / 32.f), wy0 = __fsub_rn(1.f, wy1);
        float v = __fmul_rn(fmt_cart_tap(cart, W, iy, ix), __fmul_rn(wy0, wx0));
        v = __fadd_rn(v, __fmul_rn(fmt_cart_tap(cart, W, iy, ix + 1), __fmul_rn(wy0, wx1)));
        v = __fadd_rn(v, __fmul_rn(fmt_cart_tap(cart, W, iy + 1, ix), __fmul_rn(wy1, wx0)));
        v = __fadd_rn(v, __fmul_rn(fmt_cart_tap(cart, W, iy + 1, ix + 1), __fmul_rn(wy1, wx1)));
        const double wc = 0.5 * (1.0 - cos(2.0 * FMT_PI / (double)(dw - 1) * (double)rho));
        const double wr = 0.5 * (1.0 - cos(2.0 * FMT_PI / (double)(dh - 1) * (double)phi));
        const float win = rn_sqrtf((float)(wr * wc));
        val = (double)__fmul_rn(win, v);
    }
    out[(int64_t)phi * N + rho] = val;
}

// direct DFT along x: out[y][v] = sum_x in[y][x] exp(sign 2 pi i v x / N); in real (cin == null) or complex
__global__ void fmt_dft_x_kernel(const double *__restrict__ re_in, const double *__restrict__ im_in, int M, int N, double sign,
                                 double *__restrict__ re_out, double *__restrict__ im_out)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (v >= N) return;
    double ar = 0, ai = 0;
    for (int x = 0; x < N; x++) {
        double s, c;
        sincospi(sign * 2.0 * (double)((v * x) % N) / (double)N, &s, &c);
        const double xr = re_in[(int64_t)y * N + x], xi = im_in ? im_in[(int64_t)y * N + x] : 0.0;
        ar += xr * c - xi * s; ai += xr * s + xi * c;
    }
    re_out[(int64_t)y * N + v] = ar; im_out[(int64_t)y * N + v] = ai;
}

// direct DFT along y: out[u][v] = sum_y in[y][v] exp(sign 2 pi i u y / M)
__global__ void fmt_dft_y_kernel(const double *__restrict__ re_in, const double *__restrict__ im_in, int M, int N, double sign,
                                 double *__restrict__ re_out, double *__restrict__ im_out)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x, u = blockIdx.y;
    if (v >= N) return;
    double ar = 0, ai = 0;
    for (int y = 0; y < M; y++) {
        double s, c;
        sincospi(sign * 2.0 * (double)((u * y) % M) / (double)M, &s, &c);
        const double xr = re_in[(int64_t)y * N + v], xi = im_in[(int64_t)y * N + v];
        ar += xr * c - xi * s; ai += xr * s + xi * c;
    }
    re_out[(int64_t)u * N + v] = ar;
    if (im_out) im_out[(int64_t)u * N + v] = ai;
}

// normalised cross-power spectrum: mulSpectrums(F1, F2, conjB) then divSpectrums by its magnitude: P |P| / (|P|^2 + FLT_EPSILON)
__global__ void fmt_cross_power_kernel(const double *__restrict__ r1, const double *__restrict__ i1, const double *__restrict__ r2,
                                       const double *__restrict__ i2, int n, double *__restrict__ cr, double *__restrict__ ci)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double pr = r1[k] * r2[k] + i1[k] * i2[k], pi = i1[k] * r2[k] - r1[k] * i2[k];
    const double mag = sqrt(pr * pr + pi * pi), den = mag * mag + 1.1920928955078125e-07;
    cr[k] = pr * mag / den; ci[k] = pi * mag / den;
}

// fftshift + first maximum (row-major) + 5 x 5 weighted centroid; one workgroup
__global__ __launch_bounds__(256) void fmt_peak_kernel(const double *__restrict__ c, int M, int N, double *__restrict__ out3)
{
    __shared__ double bv[256];
    __shared__ int bi[256];
    const int t = threadIdx.x, n = M * N;
    double best = -1e300; int besti = n;
    for (int k = t; k < n; k += 256) {                      // k indexes the SHIFTED image
        const int y = k / N, x = k - y * N;
        const double v = c[(int64_t)((y + M / 2) % M) * N + ((x + N / 2) % N)];
        if (v > best) { best = v; besti = k; }
    }
    bv[t] = best; bi[t] = besti;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (t < s && (bv[t + s] > bv[t] || (bv[t + s] == bv[t] && bi[t + s] < bi[t]))) { bv[t] = bv[t + s]; bi[t] = bi[t + s]; }
        __syncthreads();
    }
    if (t == 0) {
        const int py = bi[0] / N, px = bi[0] - py * N;
        const int r0 = max(py - 2, 0), r1 = min(py + 2, M - 1), c0 = max(px - 2, 0), c1 = min(px + 2, N - 1);
        double sx = 0, sy = 0, sum = 0;
        for (int y = r0; y <= r1; y++)
            for (int x = c0; x <= c1; x++) {
                const double v = c[(int64_t)((y + M / 2) % M) * N + ((x + N / 2) % N)];
                sx += (double)x * v; sy += (double)y * v; sum += v;
            }
        const double den = sum + 2.220446049250313e-16;
        out3[0] = (double)N / 2.0 - sx / den;               // d rho ("scale" axis)
        out3[1] = (double)M / 2.0 - sy / den;               // d phi ("angle" axis)
        out3[2] = sum / ((double)M * (double)N);            // response (the inverse DFT here is unscaled, like cv2.idft)
    }
}

static int optimal_dft_size(int n)
{
    int best = 0;
    for (long p2 = 1; p2 < 2L * n; p2 *= 2)
        for (long p3 = p2; p3 < 2L * n; p3 *= 3)
            for (long p5 = p3; p5 < 2L * n; p5 *= 5)
                if (p5 >= n && (best == 0 || p5 < best)) best = (int)p5;
    return best;
}

extern "C" int32_t roam_fmt_rotation(roam_ctx *ctx, const float *src_polar, const float *tgt_polar, int32_t rows, int32_t cols,
                                     int32_t clip_px, int32_t downsample, double *angle_rad, double *scale, double *response)
{
    if (!ctx) return ROAM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ARG_CHECK(ctx, src_polar && tgt_polar && rows >= 8 && cols >= 2 && downsample >= 1 && angle_rad);
    const int clip = (clip_px > 0 && clip_px < cols) ? clip_px : cols;
    const int nw = clip / downsample;
    ARG_CHECK(ctx, nw >= 4 && nw <= 2048);
    const int W = 2 * nw, dw = (int)rint((double)nw), dh = (int)rint((double)nw * FMT_PI);
    const int M = optimal_dft_size(dh), N = optimal_dft_size(dw);
    hipStream_t st = ctx->stream;
    const size_t npol = (size_t)rows * cols, nmn = (size_t)M * N;
    float *d_in = (float *)roam_scratch(ctx, S_IN0, sizeof(float) * npol);
    float *d_small = (float *)roam_scratch(ctx, S_TMP0, sizeof(float) * (size_t)rows * nw);
    float *d_cart = (float *)roam_scratch(ctx, S_TMP1, sizeof(float) * (size_t)W * W);
    double *d_f = (double *)roam_scratch(ctx, S_TMP2, sizeof(double) * nmn * 8);     // a | tmp re, im | F1 re, im | F2 re, im | spare
    double *d_out = (double *)roam_scratch(ctx, S_OUT0, sizeof(double) * 4);
    if (!d_in || !d_small || !d_cart || !d_f || !d_out) return ROAM_E_HIP;
    double *a = d_f, *tr = d_f + nmn, *ti = d_f + 2 * nmn, *F[2][2] = {{d_f + 3 * nmn, d_f + 4 * nmn}, {d_f + 5 * nmn, d_f + 6 * nmn}};
    const float *imgs[2] = {src_polar, tgt_polar};
    const dim3 gmn((N + 63) / 64, M);
    for (int k = 0; k < 2; k++) {
        HIP_TRY(ctx, hipMemcpyAsync(d_in, imgs[k], sizeof(float) * npol, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(fmt_resize_kernel, dim3((nw + 63) / 64, rows), dim3(64), 0, st, d_in, rows, (int64_t)cols, clip, nw, d_small);
        hipLaunchKernelGGL(fmt_cart_kernel, dim3((W + 63) / 64, W), dim3(64), 0, st, d_small, rows, nw, d_cart);
        hipLaunchKernelGGL(fmt_logpolar_window_kernel, gmn, dim3(64), 0, st, d_cart, W, dw, dh, M, N, a);
        hipLaunchKernelGGL(fmt_dft_x_kernel, gmn, dim3(64), 0, st, a, (const double *)nullptr, M, N, -1.0, tr, ti);
        hipLaunchKernelGGL(fmt_dft_y_kernel, gmn, dim3(64), 0, st, tr, ti, M, N, -1.0, F[k][0], F[k][1]);
        HIP_TRY(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(fmt_cross_power_kernel, dim3((unsigned)((nmn + 255) / 256)), dim3(256), 0, st, F[0][0], F[0][1], F[1][0], F[1][1], (int)nmn, tr, ti);
    hipLaunchKernelGGL(fmt_dft_x_kernel, gmn, dim3(64), 0, st, tr, ti, M, N, 1.0, F[0][0], F[0][1]);
    hipLaunchKernelGGL(fmt_dft_y_kernel, gmn, dim3(64), 0, st, F[0][0], F[0][1], M, N, 1.0, a, (double *)nullptr);
    hipLaunchKernelGGL(fmt_peak_kernel, dim3(1), dim3(256), 0, st, a, M, N, d_out);
    HIP_TRY(ctx, hipGetLastError());
    double o[3];
    HIP_TRY(ctx, hipMemcpyAsync(o, d_out, sizeof(o), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const int sz = dh > dw ? dh : dw;
    double ang = -o[1] * 2.0 * FMT_PI / (double)sz;
    ang = fmod(ang + FMT_PI, 2.0 * FMT_PI);                  // utils.normalize_angles: (th + pi) % (2 pi) - pi (Python modulo)
    if (ang < 0) ang += 2.0 * FMT_PI;
    *angle_rad = ang - FMT_PI;
    if (scale) *scale = pow(exp(log((double)dh / 2.0) / (double)sz), o[0]);
    if (response) *response = o[2];
    return ROAM_OK;
}
