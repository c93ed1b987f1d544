/* ORACLE — TEST INFRASTRUCTURE ONLY (see peaks.c header).
 *
 * CPU restatement of the image-scale part of skimage.feature.blob_doh as called by
 * getFeatures.getBlobsFromCart (reference getFeatures.py:22-53 with DEFAULT_FEATURE_PARAMS
 * :13-18: min_sigma 0.01, max_sigma 10, num_sigma 3, threshold 0.0005, method "doh").
 * scikit-image (0.19.2 pinned by the reference) is an un-vendored third-party dependency that
 * is absent from this image => PARITY UNPINNED; what follows is its published algorithm:
 *   integral image (cumsum over rows then columns, float64),
 *   _hessian_matrix_det: box-filter approximation of the Hessian determinant
 *       size = int(3*sigma), s2 = (size-1)/2, s3 = size/3 (C division), w = size, w_i = 1/size^2,
 *       dxy from four s3 x s3 boxes, dxx / dyy from a (2*s3-1) x w box minus 3x its centre third,
 *       det = dxx*dyy - 0.81*dxy^2, every box sum clipped at the image border and at 0,
 *   peak_local_max(cube, threshold_abs, footprint 3x3x3, exclude_border=False): v > threshold
 *       and v == max over the 3x3x3 neighbourhood (outside the cube = 0).
 * With sigma = 0.01 the box size is 0, w_i = inf and the whole layer is NaN; scipy's maximum
 * filter (probed in this image, scipy 1.15.3) ignores that layer, and NaN > threshold is
 * false, so the layer neither yields peaks nor suppresses any: layers with size == 0 are
 * skipped here.  Output: the local maxima in C (row, col, sigma-index) order with their values;
 * the ordering by response and the overlap pruning (_prune_blobs) are done by the caller.
 */
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int clipi(int x, int lo, int hi) { return x > hi ? hi : (x < lo ? lo : x); }

static inline double integ(const double *S, int H, int W, int r, int c, int rl, int cl)
{
    r = clipi(r, 0, H - 1);
    c = clipi(c, 0, W - 1);
    int r2 = clipi(r + rl, 0, H - 1);
    int c2 = clipi(c + cl, 0, W - 1);
    double ans = S[(int64_t)r * W + c] + S[(int64_t)r2 * W + c2] - S[(int64_t)r * W + c2] - S[(int64_t)r2 * W + c];
    return ans < 0 ? 0 : ans;
}

void oracle_integral_image(const float *img, int H, int W, double *S)
{
    for (int c = 0; c < W; c++) {                 /* cumsum(axis=0) */
        double acc = 0;
        for (int r = 0; r < H; r++) { acc += (double)img[(int64_t)r * W + c]; S[(int64_t)r * W + c] = acc; }
    }
    for (int r = 0; r < H; r++) {                 /* cumsum(axis=1) */
        double acc = 0;
        for (int c = 0; c < W; c++) { acc += S[(int64_t)r * W + c]; S[(int64_t)r * W + c] = acc; }
    }
}

void oracle_hessian_det(const double *S, int H, int W, double sigma, double *out)
{
    const int size = (int)(3 * sigma);
    const int s2 = (size - 1) / 2, s3 = size / 3, w = size;
    const double w_i = 1.0 / size / size;
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) {
            double tl = integ(S, H, W, r - s3, c - s3, s3, s3);
            double br = integ(S, H, W, r + 1, c + 1, s3, s3);
            double bl = integ(S, H, W, r - s3, c + 1, s3, s3);
            double tr = integ(S, H, W, r + 1, c - s3, s3, s3);
            double dxy = bl + tr - tl - br;
            dxy = -dxy * w_i;
            double mid = integ(S, H, W, r - s3 + 1, c - s2, 2 * s3 - 1, w);
            double side = integ(S, H, W, r - s3 + 1, c - s3 / 2, 2 * s3 - 1, s3);
            double dxx = mid - 3 * side;
            dxx = -dxx * w_i;
            mid = integ(S, H, W, r - s2, c - s3 + 1, w, 2 * s3 - 1);
            side = integ(S, H, W, r - s3 / 2, c - s3 + 1, s3, 2 * s3 - 1);
            double dyy = mid - 3 * side;
            dyy = -dyy * w_i;
            out[(int64_t)r * W + c] = dxx * dyy - 0.81 * (dxy * dxy);
        }
}

/* layers: nl pointers (NULL = degenerate NaN layer, ignored).  out_rcs (cap,3) int32, out_val (cap).
 * returns the number of maxima (may exceed cap). */
int64_t oracle_doh_maxima(const double *const *layers, int nl, int H, int W, double thr,
                          int32_t *out_rcs, double *out_val, int64_t cap)
{
    int64_t n = 0;
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++)
            for (int s = 0; s < nl; s++) {
                if (!layers[s]) continue;
                double v = layers[s][(int64_t)r * W + c];
                if (!(v > thr)) continue;
                int ok = 1;
                for (int ds = -1; ds <= 1 && ok; ds++) {
                    int ss = s + ds;
                    for (int dr = -1; dr <= 1 && ok; dr++)
                        for (int dc = -1; dc <= 1; dc++) {
                            double u = 0.0;             /* mode='constant', cval=0 */
                            int rr = r + dr, cc = c + dc;
                            if (ss >= 0 && ss < nl && rr >= 0 && rr < H && cc >= 0 && cc < W) {
                                if (!layers[ss]) continue;      /* NaN layer: ignored by the max filter */
                                u = layers[ss][(int64_t)rr * W + cc];
                            }
                            if (u > v) { ok = 0; break; }
                        }
                }
                if (ok) {
                    if (n < cap) { out_rcs[3 * n] = r; out_rcs[3 * n + 1] = c; out_rcs[3 * n + 2] = s; out_val[n] = v; }
                    n++;
                }
            }
    return n;
}
