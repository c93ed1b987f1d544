/* ORACLE — TEST INFRASTRUCTURE ONLY (see peaks.c header).
 *
 * CPU restatement of the two OpenCV-backed stages of the path.  OpenCV is an un-vendored,
 * UNPINNED third-party dependency of the reference (absent from requirements.txt, not
 * installed in this image) => PARITY UNPINNED at these two boundaries: what is restated
 * here is the published algorithm with the reference's parameters, not a verified copy
 * of one OpenCV version.
 *
 * (1) convertPolarImageToCartesian (reference parseData.py:100-135):
 *     cv2.warpPolar(polar, (2R,2R), center=(R,R), maxRadius=R,
 *                   WARP_POLAR_LINEAR|WARP_INVERSE_MAP|INTER_LINEAR|WARP_FILL_OUTLIERS), R=cols//2.
 *     OpenCV builds float maps  mx = mag/Kmag, my = angle/Kangle + 1  (mag/angle from
 *     cartToPolar = sqrt + the degree-7 "fastAtan" polynomial, one wrap row above/below),
 *     then remap(): coordinates rounded to 1/32 px, bilinear weights from a 32x32 table,
 *     taps outside the source read 0 (BORDER_CONSTANT).
 *     followed by the reference's quantisation (img*255).astype(uint8), getTransformKLT.py:356-357.
 * (2) getTrackedPointsKLT (reference getTransformKLT.py:317-381):
 *     cv2.calcOpticalFlowPyrLK(u8, u8, pts, winSize=(15,15), maxLevel=3,
 *                              criteria=(EPS|COUNT,10,0.03))  + status &= err < 10.
 *     Bouguet pyramidal LK as OpenCV ships it: pyrDown 5x5 [1 4 6 4 1]^2/256 with
 *     REFLECT_101, Scharr int16 derivatives (zero outside the image), 14-bit fixed-point
 *     bilinear patch sampling (5 fractional bits kept in the patch), 2x2 normal equations,
 *     min-eigenvalue test 1e-4, <=10 iterations, |delta|^2 <= 0.03^2 stop, oscillation stop,
 *     error = mean |dI| / 32 over the window at level 0.
 *     One deliberate, documented difference: window sums are accumulated EXACTLY in
 *     int64 and converted to float once (OpenCV accumulates in float / SIMD int lanes in an
 *     order that depends on the build), which makes the result independent of summation
 *     order and therefore bit-reproducible on a GPU.
 * All float arithmetic below is written operation by operation and the file is compiled
 * with -ffp-contract=off so that the HIP kernels can reproduce it bit for bit.
 */
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

/* round half to even, like cvRound (lrint under the default rounding mode) */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }

/* OpenCV fastAtan (degrees), scalar form */
static float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)2.220446049250313e-16);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)2.220446049250313e-16);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* source sample with the wrap border in azimuth and zero fill in range */
static inline float polar_tap(const float *polar, int rows, int cols, int64_t stride, int py, int px)
{
    /* py indexes the padded image (rows+2): 0 -> row rows-1, rows+1 -> row 0 */
    if (px < 0 || px >= cols || py < 0 || py >= rows + 2) return 0.f;
    int r = py - 1;
    if (r < 0) r += rows;
    else if (r >= rows) r -= rows;
    return polar[(int64_t)r * stride + px];
}

/* polar (rows x cols f32) -> cart (2R x 2R f32), R = cols/2.  Optionally also the u8
 * quantisation the KLT wrapper applies.  Either output may be NULL. */
static int64_t polar_to_cart_impl(const float *polar, int rows, int cols, int64_t stride,
                                  float *cart_f32, uint8_t *cart_u8, float tie_eps, int tie_every)
{
    int64_t n_flipped = 0, n_ties = 0;
    const int R = cols / 2;
    const int W = 2 * R;
    const double Kangle = 6.283185307179586476925286766559 / rows;
    const double Kmag = (double)R / cols;
    const float cx = (float)R, cy = (float)R;      /* center = cartSize/2 */
    const float deg2rad = (float)(3.14159265358979323846 / 180.0);
    for (int y = 0; y < W; y++) {
        float fy = (float)y - cy;
        for (int x = 0; x < W; x++) {
            float fx = (float)x - cx;
            float mag = sqrtf(fx * fx + fy * fy);
            float ang = fast_atan2_deg(fy, fx) * deg2rad;
            double rho = (double)mag / Kmag;
            double phi = (double)ang / Kangle;
            float mx = (float)rho;
            float my = (float)phi + 1.f;
            int sx = cv_round_f(mx * 32.f);
            int sy = cv_round_f(my * 32.f);
            if (tie_eps > 0.f) {
                /* the radial coordinate within tie_eps of a rounding tie of the 1/32-px grid: take the OTHER neighbour (what a
                 * magnitude that differs by an ulp - IPP's ippsMagnitude_32f in the reference's cv2 build - can do) */
                float t = mx * 32.f, fl = floorf(t), fr = t - fl;
                if (fabsf(fr - 0.5f) < tie_eps && (n_ties++ % tie_every) == 0) { sx = (sx == (int)fl) ? (int)fl + 1 : (int)fl; n_flipped++; }
            }
            int ix = sx >> 5, iy = sy >> 5;
            int fxq = sx & 31, fyq = sy & 31;
            float wx1 = (float)fxq * (1.f / 32.f), wx0 = 1.f - wx1;
            float wy1 = (float)fyq * (1.f / 32.f), wy0 = 1.f - wy1;
            float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
            float s00 = polar_tap(polar, rows, cols, stride, iy, ix);
            float s01 = polar_tap(polar, rows, cols, stride, iy, ix + 1);
            float s10 = polar_tap(polar, rows, cols, stride, iy + 1, ix);
            float s11 = polar_tap(polar, rows, cols, stride, iy + 1, ix + 1);
            float v = s00 * w00;
            v = v + s01 * w01;
            v = v + s10 * w10;
            v = v + s11 * w11;
            if (cart_f32) cart_f32[(int64_t)y * W + x] = v;
            if (cart_u8) {
                float q = v * 255.f;                 /* (img*255).astype(uint8): truncation */
                int qi = (int)q;
                cart_u8[(int64_t)y * W + x] = (uint8_t)qi;
            }
        }
    }
    return n_flipped;
}

void oracle_polar_to_cart(const float *polar, int rows, int cols, int64_t stride,
                          float *cart_f32, uint8_t *cart_u8)
{
    polar_to_cart_impl(polar, rows, cols, stride, cart_f32, cart_u8, 0.f, 1);
}

/* sensitivity probe (tests/test_oracle_reference_dump.py): the same warp with every tie_every-th radial sampling coordinate that
 * lies within tie_eps of a rounding tie rounded the other way; returns the number of pixels treated so */
int64_t oracle_polar_to_cart_ties(const float *polar, int rows, int cols, int64_t stride,
                                  float *cart_f32, uint8_t *cart_u8, float tie_eps, int tie_every)
{
    return polar_to_cart_impl(polar, rows, cols, stride, cart_f32, cart_u8, tie_eps, tie_every < 1 ? 1 : tie_every);
}

/* (img*255).astype(uint8) on an arbitrary f32 image in [0,1] */
void oracle_quantize_u8(const float *img, int64_t n, uint8_t *out)
{
    for (int64_t i = 0; i < n; i++) { float q = img[i] * 255.f; out[i] = (uint8_t)(int)q; }
}

/* pyrDown: dst ((w+1)/2 x (h+1)/2) */
void oracle_pyr_down_u8(const uint8_t *src, int w, int h, uint8_t *dst)
{
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    static const int k[5] = {1, 4, 6, 4, 1};
    int *rowbuf = (int *)malloc(sizeof(int) * (size_t)dw * 5);
    for (int y = 0; y < dh; y++) {
        for (int j = 0; j < 5; j++) {
            int sy = reflect101(2 * y + j - 2, h);
            const uint8_t *s = src + (int64_t)sy * w;
            for (int x = 0; x < dw; x++) {
                int acc = 0;
                for (int i = 0; i < 5; i++) acc += k[i] * s[reflect101(2 * x + i - 2, w)];
                rowbuf[j * dw + x] = acc;
            }
        }
        for (int x = 0; x < dw; x++) {
            int acc = 0;
            for (int j = 0; j < 5; j++) acc += k[j] * rowbuf[j * dw + x];
            dst[(int64_t)y * dw + x] = (uint8_t)((acc + 128) >> 8);
        }
    }
    free(rowbuf);
}

typedef struct { const uint8_t *p; int w, h; } Img;

static inline int img_at(const Img *im, int x, int y)   /* REFLECT_101 border */
{
    return im->p[(int64_t)reflect101(y, im->h) * im->w + reflect101(x, im->w)];
}

/* Scharr derivative at (x,y): zero outside the image, REFLECT_101 taps inside */
static inline void scharr_at(const Img *im, int x, int y, int *dx, int *dy)
{
    if (x < 0 || y < 0 || x >= im->w || y >= im->h) { *dx = 0; *dy = 0; return; }
    int a00 = img_at(im, x - 1, y - 1), a01 = img_at(im, x, y - 1), a02 = img_at(im, x + 1, y - 1);
    int a10 = img_at(im, x - 1, y),                                 a12 = img_at(im, x + 1, y);
    int a20 = img_at(im, x - 1, y + 1), a21 = img_at(im, x, y + 1), a22 = img_at(im, x + 1, y + 1);
    *dx = 3 * (a02 + a22 - a00 - a20) + 10 * (a12 - a10);
    *dy = 3 * (a20 + a22 - a00 - a02) + 10 * (a21 - a01);
}

#define W_BITS 14
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

/* pyramids: levels 0..3 for both images, prev_w[l], prev_h[l].
 * pts (K,2) f32 [x,y] -> next (K,2) f32, status (K) u8, err (K) f32 */
void oracle_klt_track(const uint8_t *const *prevPyr, const uint8_t *const *nextPyr,
                      const int *lw, const int *lh, int nlevels,
                      const float *pts, int K, int win, int max_iter, float eps,
                      float min_eig_thr,
                      float *next, uint8_t *status, float *err)
{
    const float halfWin = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float eps2 = eps * eps;
    int16_t *Ibuf = (int16_t *)malloc(sizeof(int16_t) * (size_t)win * win * 3);
    for (int k = 0; k < K; k++) { status[k] = 1; err[k] = 0.f; next[2 * k] = 0; next[2 * k + 1] = 0; }
    for (int level = nlevels - 1; level >= 0; level--) {
        Img I = {prevPyr[level], lw[level], lh[level]};
        Img J = {nextPyr[level], lw[level], lh[level]};
        for (int k = 0; k < K; k++) {
            float scale = (float)(1. / (1 << level));
            float px = pts[2 * k] * scale, py = pts[2 * k + 1] * scale;
            float nx, ny;
            if (level == nlevels - 1) { nx = px; ny = py; }
            else { nx = next[2 * k] * 2.f; ny = next[2 * k + 1] * 2.f; }
            next[2 * k] = nx; next[2 * k + 1] = ny;
            px -= halfWin; py -= halfWin;
            int ipx = cv_floor_f(px), ipy = cv_floor_f(py);
            if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
                if (level == 0) { status[k] = 0; err[k] = 0.f; }
                continue;
            }
            float a = px - (float)ipx, b = py - (float)ipy;
            int iw00 = cv_round_f((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
            int iw01 = cv_round_f(a * (1.f - b) * (float)(1 << W_BITS));
            int iw10 = cv_round_f((1.f - a) * b * (float)(1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int64_t sA11 = 0, sA12 = 0, sA22 = 0;
            for (int y = 0; y < win; y++)
                for (int x = 0; x < win; x++) {
                    int X = ipx + x, Y = ipy + y;
                    int ival = DESCALE(img_at(&I, X, Y) * iw00 + img_at(&I, X + 1, Y) * iw01 +
                                       img_at(&I, X, Y + 1) * iw10 + img_at(&I, X + 1, Y + 1) * iw11,
                                       W_BITS - 5);
                    int d00x, d00y, d01x, d01y, d10x, d10y, d11x, d11y;
                    scharr_at(&I, X, Y, &d00x, &d00y);
                    scharr_at(&I, X + 1, Y, &d01x, &d01y);
                    scharr_at(&I, X, Y + 1, &d10x, &d10y);
                    scharr_at(&I, X + 1, Y + 1, &d11x, &d11y);
                    int ixval = DESCALE(d00x * iw00 + d01x * iw01 + d10x * iw10 + d11x * iw11, W_BITS);
                    int iyval = DESCALE(d00y * iw00 + d01y * iw01 + d10y * iw10 + d11y * iw11, W_BITS);
                    int16_t *q = Ibuf + 3 * (y * win + x);
                    q[0] = (int16_t)ival; q[1] = (int16_t)ixval; q[2] = (int16_t)iyval;
                    sA11 += (int64_t)ixval * ixval;
                    sA12 += (int64_t)ixval * iyval;
                    sA22 += (int64_t)iyval * iyval;
                }
            float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            float dA = A11 - A22;
            float minEig = (A22 + A11 - sqrtf(dA * dA + 4.f * A12 * A12)) / (float)(2 * win * win);
            if (minEig < min_eig_thr || D < 1.1920929e-07f) {
                if (level == 0) status[k] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= halfWin; ny -= halfWin;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < max_iter; j++) {
                int inx = cv_floor_f(nx), iny = cv_floor_f(ny);
                if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                    if (level == 0) status[k] = 0;
                    break;
                }
                a = nx - (float)inx; b = ny - (float)iny;
                iw00 = cv_round_f((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
                iw01 = cv_round_f(a * (1.f - b) * (float)(1 << W_BITS));
                iw10 = cv_round_f((1.f - a) * b * (float)(1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t sb1 = 0, sb2 = 0;
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++) {
                        int X = inx + x, Y = iny + y;
                        int jv = DESCALE(img_at(&J, X, Y) * iw00 + img_at(&J, X + 1, Y) * iw01 +
                                         img_at(&J, X, Y + 1) * iw10 + img_at(&J, X + 1, Y + 1) * iw11,
                                         W_BITS - 5);
                        const int16_t *q = Ibuf + 3 * (y * win + x);
                        int diff = jv - q[0];
                        sb1 += (int64_t)diff * q[1];
                        sb2 += (int64_t)diff * q[2];
                    }
                float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
                float dx = (A12 * b2 - A22 * b1) * D;
                float dy = (A12 * b1 - A11 * b2) * D;
                nx += dx; ny += dy;
                next[2 * k] = nx + halfWin; next[2 * k + 1] = ny + halfWin;
                if (dx * dx + dy * dy <= eps2) break;
                if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                    next[2 * k] -= dx * 0.5f; next[2 * k + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
            if (status[k] && level == 0) {
                float ex = next[2 * k] - halfWin, ey = next[2 * k + 1] - halfWin;
                int iex = cv_floor_f(ex), iey = cv_floor_f(ey);
                if (iex < -win || iex >= J.w || iey < -win || iey >= J.h) { status[k] = 0; continue; }
                float aa = ex - (float)iex, bb = ey - (float)iey;
                iw00 = cv_round_f((1.f - aa) * (1.f - bb) * (float)(1 << W_BITS));
                iw01 = cv_round_f(aa * (1.f - bb) * (float)(1 << W_BITS));
                iw10 = cv_round_f((1.f - aa) * bb * (float)(1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t se = 0;
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++) {
                        int X = iex + x, Y = iey + y;
                        int jv = DESCALE(img_at(&J, X, Y) * iw00 + img_at(&J, X + 1, Y) * iw01 +
                                         img_at(&J, X, Y + 1) * iw10 + img_at(&J, X + 1, Y + 1) * iw11,
                                         W_BITS - 5);
                        int diff = jv - Ibuf[3 * (y * win + x)];
                        se += diff < 0 ? -diff : diff;
                    }
                err[k] = (float)se * (1.f / (float)(32 * win * win));
            }
        }
    }
    free(Ibuf);
}

/* ------------------------------------------------------------------ general warpPolar pieces for the FMT rotation prior
 * (reference FMT.py:36-90 -> parseData.convertPolarImgToLogPolar, parseData.py:56-160): the same OpenCV algorithms as above
 * with free geometry.  PARITY UNPINNED: the reference holds no output of this path (its result is only printed). */

/* inverse linear warp: polar (rows x cols) -> W x H Cartesian, centre (cx, cy), maxRadius; WARP_FILL_OUTLIERS */
void oracle_warp_polar_inverse(const float *polar, int rows, int cols, int64_t stride, int W, int H, float cx, float cy,
                               double maxRadius, float *out)
{
    const double Kangle = 6.283185307179586476925286766559 / rows;
    const double Kmag = maxRadius / cols;
    const float deg2rad = (float)(3.14159265358979323846 / 180.0);
    for (int y = 0; y < H; y++) {
        float fy = (float)y - cy;
        for (int x = 0; x < W; x++) {
            float fx = (float)x - cx;
            float mag = sqrtf(fx * fx + fy * fy);
            float ang = fast_atan2_deg(fy, fx) * deg2rad;
            double rho = (double)mag / Kmag, phi = (double)ang / Kangle;
            float mx = (float)rho, my = (float)phi + 1.f;
            int sx = cv_round_f(mx * 32.f), sy = cv_round_f(my * 32.f);
            int ix = sx >> 5, iy = sy >> 5, fxq = sx & 31, fyq = sy & 31;
            float wx1 = (float)fxq * (1.f / 32.f), wx0 = 1.f - wx1, wy1 = (float)fyq * (1.f / 32.f), wy0 = 1.f - wy1;
            float v = polar_tap(polar, rows, cols, stride, iy, ix) * (wy0 * wx0);
            v = v + polar_tap(polar, rows, cols, stride, iy, ix + 1) * (wy0 * wx1);
            v = v + polar_tap(polar, rows, cols, stride, iy + 1, ix) * (wy1 * wx0);
            v = v + polar_tap(polar, rows, cols, stride, iy + 1, ix + 1) * (wy1 * wx1);
            out[(int64_t)y * W + x] = v;
        }
    }
}

static inline float cart_tap(const float *img, int W, int H, int y, int x)
{
    return (x < 0 || x >= W || y < 0 || y >= H) ? 0.f : img[(int64_t)y * W + x];
}

/* forward semilog warp: Cartesian (W x H) -> log-polar (dw x dh): rho column, phi row; WARP_POLAR_LOG | WARP_FILL_OUTLIERS */
void oracle_warp_polar_forward_log(const float *cart, int W, int H, int dw, int dh, float cx, float cy, double maxRadius, float *out)
{
    const double Kangle = 6.283185307179586476925286766559 / dh;
    const double Kmag = log(maxRadius) / dw;
    for (int phi = 0; phi < dh; phi++) {
        const double KKy = Kangle * phi, cp = cos(KKy), sp = sin(KKy);
        for (int rho = 0; rho < dw; rho++) {
            const float br = (float)(exp(rho * Kmag) - 1.0);
            const float mx = (float)((double)br * cp + (double)cx), my = (float)((double)br * sp + (double)cy);
            int sx = cv_round_f(mx * 32.f), sy = cv_round_f(my * 32.f);
            int ix = sx >> 5, iy = sy >> 5, fxq = sx & 31, fyq = sy & 31;
            float wx1 = (float)fxq * (1.f / 32.f), wx0 = 1.f - wx1, wy1 = (float)fyq * (1.f / 32.f), wy0 = 1.f - wy1;
            float v = cart_tap(cart, W, H, iy, ix) * (wy0 * wx0);
            v = v + cart_tap(cart, W, H, iy, ix + 1) * (wy0 * wx1);
            v = v + cart_tap(cart, W, H, iy + 1, ix) * (wy1 * wx0);
            v = v + cart_tap(cart, W, H, iy + 1, ix + 1) * (wy1 * wx1);
            out[(int64_t)phi * dw + rho] = v;
        }
    }
}
