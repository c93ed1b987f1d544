/* ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product
 * path (radarslampy_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker / the timed CPU baseline.
 *
 * CPU restatement of getPointCloud.getPointCloudPolarInd (reference getPointCloud.py:11-54):
 *   per azimuth row: scipy.signal.find_peaks(row) with distance=prominence=None
 *   (= scipy.signal._peak_finding_utils._local_maxima_1d: strict rise, plateau -> midpoint
 *   (l+r)//2, first/last sample never a peak), heights h = row[peaks];
 *   keep h >= mean(h) + std(h) where numpy evaluates mean/std in float32 with its
 *   pairwise summation (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum, blocks of
 *   128, 8 accumulators) — restated here so that ties at the threshold resolve the same way.
 * Input is either the clipped float32 polar image (parseData.py:40,49-51: u8/255 in f32)
 * or the u8 power codes themselves (fused decode).
 * Pinned by tests/golden/peaks.npz (outputs of the reference on real + synthetic scans).
 */
#include <stdint.h>
#include <math.h>
#include <stdlib.h>

/* numpy pairwise float32 sum of a contiguous array */
static float pairwise_sum_f32(const float *a, int64_t n)
{
    if (n < 8) {
        float res = 0.f;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum_f32(a, n2) + pairwise_sum_f32(a + n2, n - n2);
    }
}

float oracle_pairwise_sum_f32(const float *a, int64_t n) { return pairwise_sum_f32(a, n); }

/* one row; returns number of valid peaks written to out_idx (ascending) */
static int row_peaks(const float *x, int n, int32_t *out_idx, float *h, int32_t *mid)
{
    int m = 0;
    int i = 1, i_max = n - 1;
    while (i < i_max) {
        if (x[i - 1] < x[i]) {
            int ia = i + 1;
            while (ia < i_max && x[ia] == x[i]) ia++;
            if (x[ia] < x[i]) {
                mid[m] = (i + ia - 1) / 2;
                m++;
                i = ia;
            }
        }
        i++;
    }
    if (m == 0) return 0;                      /* numpy: mean of empty = NaN -> nothing passes */
    for (int k = 0; k < m; k++) h[k] = x[mid[k]];
    float mean = pairwise_sum_f32(h, m) / (float)m;
    float *sq = h + m;                         /* scratch behind h (caller sizes 2n) */
    for (int k = 0; k < m; k++) { float d = h[k] - mean; sq[k] = d * d; }
    float var = pairwise_sum_f32(sq, m) / (float)m;
    float sd = sqrtf(var);
    float thr = mean + sd;
    int c = 0;
    for (int k = 0; k < m; k++)
        if (h[k] >= thr) out_idx[c++] = mid[k];
    return c;
}

/* polar: rows x cols float32 (row stride = stride floats).  out: (cap,2) int32 [az, rng].
 * returns total count (may exceed cap; only the first cap pairs are written). */
int64_t oracle_peaks_f32(const float *polar, int rows, int cols, int64_t stride,
                         int32_t *out, int64_t cap)
{
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)cols * 2);
    float *h = (float *)malloc(sizeof(float) * (size_t)cols * 2);
    int64_t tot = 0;
    for (int r = 0; r < rows; r++) {
        int c = row_peaks(polar + (int64_t)r * stride, cols, idx, h, idx + cols);
        for (int k = 0; k < c; k++, tot++)
            if (tot < cap) { out[2 * tot] = r; out[2 * tot + 1] = idx[k]; }
    }
    free(idx); free(h);
    return tot;
}

/* raw Oxford record rows: rec[r*stride + payload_off + i], i < clip; value = u8/255 in f32 */
int64_t oracle_peaks_u8(const uint8_t *rec, int rows, int64_t stride, int payload_off, int clip,
                        int32_t *out, int64_t cap)
{
    float *row = (float *)malloc(sizeof(float) * (size_t)clip);
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)clip * 2);
    float *h = (float *)malloc(sizeof(float) * (size_t)clip * 2);
    int64_t tot = 0;
    for (int r = 0; r < rows; r++) {
        const uint8_t *p = rec + (int64_t)r * stride + payload_off;
        for (int i = 0; i < clip; i++) row[i] = (float)p[i] / 255.f;
        int c = row_peaks(row, clip, idx, h, idx + clip);
        for (int k = 0; k < c; k++, tot++)
            if (tot < cap) { out[2 * tot] = r; out[2 * tot + 1] = idx[k]; }
    }
    free(row); free(idx); free(h);
    return tot;
}
