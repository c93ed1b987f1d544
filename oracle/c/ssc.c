/* ORACLE — TEST INFRASTRUCTURE ONLY (see peaks.c header).
 *
 * CPU restatement of ANMS.ssc (reference ANMS.py:5-102), "suppression via square
 * covering": binary search over the square width; for every width a greedy pass over the
 * keypoints in the given order accepts a keypoint when its grid cell (cell size c=width/2)
 * is not covered yet and then covers the (2*floor(width/c)+1)^2 block of cells around it.
 * A later keypoint j is therefore rejected exactly when an accepted keypoint i has
 * |row_i-row_j| <= w and |col_i-col_j| <= w in cell units (the clipping of the covered
 * block at the grid border cannot matter for cells that contain keypoints), which is how
 * the greedy pass is evaluated here — no (2024/c)^2 grid is materialised, so the very
 * small widths that make the reference run out of memory (B < 180) stay computable.
 * Pinned by tests/golden/ssc.npz.
 */
#include <stdint.h>
#include <math.h>
#include <stdlib.h>

/* kp: (B,3) f64 rows [row, col, sigma]; sel_out: up to B indices; returns count */
int oracle_ssc(const double *kp, int B, int num_ret, double tol, int cols, int rows, int32_t *sel_out)
{
    double exp1 = rows + cols + 2 * num_ret;
    double exp2 = 4.0 * cols + 4.0 * num_ret + 4.0 * rows * num_ret + (double)rows * rows +
                  (double)cols * cols - 2.0 * rows * cols + 4.0 * rows * cols * num_ret;
    double exp3 = sqrt(exp2);
    double exp4 = num_ret - 1;
    double sol1 = -rint((exp1 + exp3) / exp4);
    double sol2 = -rint((exp1 - exp3) / exp4);
    double high = sol1 > sol2 ? sol1 : sol2;
    double low = floor(sqrt((double)B / num_ret));
    double prev_width = -1;
    int k_min = (int)rint(num_ret - num_ret * tol);
    int k_max = (int)rint(num_ret + num_ret * tol);
    int32_t *res = (int32_t *)malloc(sizeof(int32_t) * (size_t)(B > 0 ? B : 1) * 3);
    int32_t *cr = res + B, *cc = cr + B;
    int nres = 0, nsel = 0;
    for (;;) {
        double width = low + (high - low) / 2;
        if (width == prev_width || low > high || width == 0) {
            for (int i = 0; i < nres; i++) sel_out[i] = res[i];
            nsel = nres;
            break;
        }
        double c = width / 2;
        int w = (int)floor(width / c);
        nres = 0;
        for (int i = 0; i < B; i++) {
            int r = (int)floor(kp[3 * i] / c), q = (int)floor(kp[3 * i + 1] / c);
            int covered = 0;
            for (int a = 0; a < nres; a++) {
                int dr = cr[a] - r, dc = cc[a] - q;
                if (dr < 0) dr = -dr;
                if (dc < 0) dc = -dc;
                if (dr <= w && dc <= w) { covered = 1; break; }
            }
            if (!covered) { res[nres] = i; cr[nres] = r; cc[nres] = q; nres++; }
        }
        if (nres >= k_min && nres <= k_max) {
            for (int i = 0; i < nres; i++) sel_out[i] = res[i];
            nsel = nres;
            break;
        } else if (nres < k_min) high = width - 1;
        else low = width + 1;
        prev_width = width;
    }
    free(res);
    return nsel;
}
