/* ORACLE — TEST INFRASTRUCTURE ONLY (see peaks.c header).
 *
 * CPU restatement of MotionDistortionSolver (reference motionDistortion.py):
 *   update_problem :80-99, infer_velocity :101-105, compute_time_deltas :107-124,
 *   undistort :126-153, error_vector/error :162-205, optimize_library :295-325.
 * optimize_library calls scipy.optimize.least_squares(fun, x0, jac='2-point', method='lm')
 * = MINPACK lmdif (third-party, SciPy 1.7.3 pinned by the reference, 1.15.3 in this
 * image) with ftol=xtol=gtol=1e-8, maxfev=100*n*(n+1), epsfcn=2.22e-16, factor=100,
 * mode=2 with diag=1.  The MINPACK algorithm (More, Garbow, Hillstrom 1980: lmdif, fdjac2,
 * qrfac, lmpar, qrsolv, enorm) is restated below from its published description.
 * Pinned by tests/golden/mds.npz (outputs of the reference run on seeded problems).
 */
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPSMCH 2.220446049250313e-16
#define DWARF 2.2250738585072014e-308

typedef void (*resid_fn)(const double *x, double *f, void *ctx);

static double enorm(int n, const double *x)
{
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    double agiant = rgiant / (double)n;
    for (int i = 0; i < n; i++) {
        double xabs = fabs(x[i]);
        if (xabs > rdwarf && xabs < agiant) { s2 += xabs * xabs; continue; }
        if (xabs <= rdwarf) {
            if (xabs > x3max) { double r = x3max / xabs; s3 = 1 + s3 * r * r; x3max = xabs; }
            else if (xabs != 0) { double r = xabs / x3max; s3 += r * r; }
        } else {
            if (xabs > x1max) { double r = x1max / xabs; s1 = 1 + s1 * r * r; x1max = xabs; }
            else { double r = xabs / x1max; s1 += r * r; }
        }
    }
    if (s1 != 0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0) {
        if (s2 >= x3max) return sqrt(s2 * (1 + (x3max / s2) * (x3max * s3)));
        return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * sqrt(s3);
}

#define A(i, j) a[(size_t)(j) * m + (i)]

static void qrfac(int m, int n, double *a, int *ipvt, double *rdiag, double *acnorm, double *wa)
{
    for (int j = 0; j < n; j++) {
        acnorm[j] = enorm(m, &A(0, j));
        rdiag[j] = acnorm[j]; wa[j] = rdiag[j]; ipvt[j] = j;
    }
    int minmn = m < n ? m : n;
    for (int j = 0; j < minmn; j++) {
        int kmax = j;
        for (int k = j; k < n; k++) if (rdiag[k] > rdiag[kmax]) kmax = k;
        if (kmax != j) {
            for (int i = 0; i < m; i++) { double t = A(i, j); A(i, j) = A(i, kmax); A(i, kmax) = t; }
            rdiag[kmax] = rdiag[j]; wa[kmax] = wa[j];
            int t = ipvt[j]; ipvt[j] = ipvt[kmax]; ipvt[kmax] = t;
        }
        double ajnorm = enorm(m - j, &A(j, j));
        if (ajnorm != 0) {
            if (A(j, j) < 0) ajnorm = -ajnorm;
            for (int i = j; i < m; i++) A(i, j) /= ajnorm;
            A(j, j) += 1;
            for (int k = j + 1; k < n; k++) {
                double sum = 0;
                for (int i = j; i < m; i++) sum += A(i, j) * A(i, k);
                double temp = sum / A(j, j);
                for (int i = j; i < m; i++) A(i, k) -= temp * A(i, j);
                if (rdiag[k] != 0) {
                    temp = A(j, k) / rdiag[k];
                    double d = 1 - temp * temp;
                    rdiag[k] *= sqrt(d > 0 ? d : 0);
                    double q = rdiag[k] / wa[k];
                    if (0.05 * (q * q) <= EPSMCH) {
                        rdiag[k] = enorm(m - j - 1, &A(j + 1, k));
                        wa[k] = rdiag[k];
                    }
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
}

static void qrsolv(int n, int m, double *a, const int *ipvt, const double *diag, const double *qtb,
                   double *x, double *sdiag, double *wa)
{
    for (int j = 0; j < n; j++) {
        for (int i = j; i < n; i++) A(i, j) = A(j, i);
        x[j] = A(j, j); wa[j] = qtb[j];
    }
    for (int j = 0; j < n; j++) {
        int l = ipvt[j];
        if (diag[l] != 0) {
            for (int k = j; k < n; k++) sdiag[k] = 0;
            sdiag[j] = diag[l];
            double qtbpj = 0;
            for (int k = j; k < n; k++) {
                if (sdiag[k] == 0) continue;
                double c, s;
                if (fabs(A(k, k)) < fabs(sdiag[k])) {
                    double cot = A(k, k) / sdiag[k];
                    s = 0.5 / sqrt(0.25 + 0.25 * (cot * cot)); c = s * cot;
                } else {
                    double t = sdiag[k] / A(k, k);
                    c = 0.5 / sqrt(0.25 + 0.25 * (t * t)); s = c * t;
                }
                A(k, k) = c * A(k, k) + s * sdiag[k];
                double temp = c * wa[k] + s * qtbpj;
                qtbpj = -s * wa[k] + c * qtbpj;
                wa[k] = temp;
                for (int i = k + 1; i < n; i++) {
                    temp = c * A(i, k) + s * sdiag[i];
                    sdiag[i] = -s * A(i, k) + c * sdiag[i];
                    A(i, k) = temp;
                }
            }
        }
        sdiag[j] = A(j, j);
        A(j, j) = x[j];
    }
    int nsing = n;
    for (int j = 0; j < n; j++) {
        if (sdiag[j] == 0 && nsing == n) nsing = j;
        if (nsing < n) wa[j] = 0;
    }
    for (int k = 0; k < nsing; k++) {
        int j = nsing - 1 - k;
        double sum = 0;
        for (int i = j + 1; i < nsing; i++) sum += A(i, j) * wa[i];
        wa[j] = (wa[j] - sum) / sdiag[j];
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa[j];
}

static void lmpar(int n, int m, double *a, const int *ipvt, const double *diag, const double *qtb,
                  double delta, double *par, double *x, double *sdiag, double *wa1, double *wa2)
{
    int nsing = n;
    for (int j = 0; j < n; j++) {
        wa1[j] = qtb[j];
        if (A(j, j) == 0 && nsing == n) nsing = j;
        if (nsing < n) wa1[j] = 0;
    }
    for (int k = 0; k < nsing; k++) {
        int j = nsing - 1 - k;
        wa1[j] /= A(j, j);
        double temp = wa1[j];
        for (int i = 0; i < j; i++) wa1[i] -= A(i, j) * temp;
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa1[j];
    int iter = 0;
    for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm(n, wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) { *par = 0; return; }
    double parl = 0;
    if (nsing >= n) {
        for (int j = 0; j < n; j++) { int l = ipvt[j]; wa1[j] = diag[l] * (wa2[l] / dxnorm); }
        for (int j = 0; j < n; j++) {
            double sum = 0;
            for (int i = 0; i < j; i++) sum += A(i, j) * wa1[i];
            wa1[j] = (wa1[j] - sum) / A(j, j);
        }
        double temp = enorm(n, wa1);
        parl = ((fp / delta) / temp) / temp;
    }
    for (int j = 0; j < n; j++) {
        double sum = 0;
        for (int i = 0; i <= j; i++) sum += A(i, j) * qtb[i];
        wa1[j] = sum / diag[ipvt[j]];
    }
    double gnorm = enorm(n, wa1);
    double paru = gnorm / delta;
    if (paru == 0) paru = DWARF / (delta < 0.1 ? delta : 0.1);
    if (*par < parl) *par = parl;
    if (*par > paru) *par = paru;
    if (*par == 0) *par = gnorm / dxnorm;
    for (;;) {
        iter++;
        if (*par == 0) { double t = 0.001 * paru; *par = DWARF > t ? DWARF : t; }
        double temp = sqrt(*par);
        for (int j = 0; j < n; j++) wa1[j] = temp * diag[j];
        qrsolv(n, m, a, ipvt, wa1, qtb, x, sdiag, wa2);
        for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
        dxnorm = enorm(n, wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0 && fp <= temp && temp < 0) || iter == 10) break;
        for (int j = 0; j < n; j++) { int l = ipvt[j]; wa1[j] = diag[l] * (wa2[l] / dxnorm); }
        for (int j = 0; j < n; j++) {
            wa1[j] /= sdiag[j];
            double t = wa1[j];
            for (int i = j + 1; i < n; i++) wa1[i] -= A(i, j) * t;
        }
        temp = enorm(n, wa1);
        double parc = ((fp / delta) / temp) / temp;
        if (fp > 0 && *par > parl) parl = *par;
        if (fp < 0 && *par < paru) paru = *par;
        double np_ = *par + parc;
        *par = parl > np_ ? parl : np_;
    }
}

/* returns MINPACK info; x is updated in place */
int oracle_lmdif(resid_fn fcn, void *ctx, int m, int n, double *x, double ftol, double xtol,
                 double gtol, int maxfev, double epsfcn, double factor, int *nfev_out)
{
    double *fvec = (double *)malloc(sizeof(double) * (size_t)m * (n + 3));
    double *a = fvec + m;                       /* m x n column major */
    double *wa4 = a + (size_t)m * n;
    double *wf = wa4 + m;                       /* fdjac scratch */
    double diag[16], qtf[16], wa1[16], wa2[16], wa3[16];
    int ipvt[16];
    int info = 0, nfev = 0, iter = 1;
    double par = 0, delta = 0, xnorm = 0, fnorm, gnorm = 0;
    for (int j = 0; j < n; j++) diag[j] = 1.0;
    fcn(x, fvec, ctx); nfev = 1;
    fnorm = enorm(m, fvec);
    for (;;) {
        /* forward-difference jacobian */
        double eps = sqrt(epsfcn > EPSMCH ? epsfcn : EPSMCH);
        for (int j = 0; j < n; j++) {
            double temp = x[j];
            double h = eps * fabs(temp);
            if (h == 0) h = eps;
            x[j] = temp + h;
            fcn(x, wf, ctx);
            x[j] = temp;
            for (int i = 0; i < m; i++) A(i, j) = (wf[i] - fvec[i]) / h;
        }
        nfev += n;
        qrfac(m, n, a, ipvt, wa1, wa2, wa3);
        if (iter == 1) {
            for (int j = 0; j < n; j++) wa3[j] = diag[j] * x[j];
            xnorm = enorm(n, wa3);
            delta = factor * xnorm;
            if (delta == 0) delta = factor;
        }
        memcpy(wa4, fvec, sizeof(double) * m);
        for (int j = 0; j < n; j++) {
            if (A(j, j) != 0) {
                double sum = 0;
                for (int i = j; i < m; i++) sum += A(i, j) * wa4[i];
                double temp = -sum / A(j, j);
                for (int i = j; i < m; i++) wa4[i] += A(i, j) * temp;
            }
            A(j, j) = wa1[j];
            qtf[j] = wa4[j];
        }
        gnorm = 0;
        if (fnorm != 0)
            for (int j = 0; j < n; j++) {
                int l = ipvt[j];
                if (wa2[l] != 0) {
                    double sum = 0;
                    for (int i = 0; i <= j; i++) sum += A(i, j) * (qtf[i] / fnorm);
                    double g = fabs(sum / wa2[l]);
                    if (g > gnorm) gnorm = g;
                }
            }
        if (gnorm <= gtol) { info = 4; break; }
        double ratio;
        do {
            double sd[16], p[16];
            lmpar(n, m, a, ipvt, diag, qtf, delta, &par, p, sd, wa3, wa4);
            for (int j = 0; j < n; j++) { wa1[j] = -p[j]; wa2[j] = x[j] + wa1[j]; wa3[j] = diag[j] * wa1[j]; }
            double pnorm = enorm(n, wa3);
            if (iter == 1 && pnorm < delta) delta = pnorm;
            fcn(wa2, wa4, ctx); nfev++;
            double fnorm1 = enorm(m, wa4);
            double actred = -1;
            if (0.1 * fnorm1 < fnorm) { double r = fnorm1 / fnorm; actred = 1 - r * r; }
            for (int j = 0; j < n; j++) {
                wa3[j] = 0;
                double temp = wa1[ipvt[j]];
                for (int i = 0; i <= j; i++) wa3[i] += A(i, j) * temp;
            }
            double temp1 = enorm(n, wa3) / fnorm;
            double temp2 = (sqrt(par) * pnorm) / fnorm;
            double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            double dirder = -(temp1 * temp1 + temp2 * temp2);
            ratio = 0;
            if (prered != 0) ratio = actred / prered;
            if (ratio <= 0.25) {
                double temp;
                if (actred >= 0) temp = 0.5;
                else temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                double dm = pnorm / 0.1;
                delta = temp * (delta < dm ? delta : dm);
                par /= temp;
            } else if (par == 0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par *= 0.5;
            }
            if (ratio >= 1e-4) {
                for (int j = 0; j < n; j++) { x[j] = wa2[j]; wa2[j] = diag[j] * x[j]; }
                memcpy(fvec, wa4, sizeof(double) * m);
                xnorm = enorm(n, wa2);
                fnorm = fnorm1;
                iter++;
            }
            if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1 && info == 2) info = 3;
            if (info != 0) goto done;
            if (nfev >= maxfev) info = 5;
            if (fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1) info = 6;
            if (delta <= EPSMCH * xnorm) info = 7;
            if (gnorm <= EPSMCH) info = 8;
            if (info != 0) goto done;
        } while (ratio < 1e-4);
    }
done:
    if (nfev_out) *nfev_out = nfev;
    free(fvec);
    return info;
}

/* ------------------------------------------------------------------ MDS residual */
typedef struct {
    int N;
    const double *p_w;     /* N x 2 */
    const double *p_jt;    /* N x 2 */
    double *dT;            /* N */
    double T0inv[9];
    double info_p[2], info_v[3];
    double period;
} Mds;

static double wrap_pi(double a)
{
    const double twopi = 2 * M_PI;
    double r = fmod(a + M_PI, twopi);
    if (r < 0) r += twopi;              /* python % : result has the sign of the divisor */
    return r - M_PI;
}

static void mds_resid(const double *x, double *f, void *vctx)
{
    const Mds *c = (const Mds *)vctx;
    double th = x[5], tx = x[3], ty = x[4];
    double ct = cos(th), st = sin(th);
    for (int i = 0; i < c->N; i++) {
        double dT = c->dT[i];
        double a = x[2] * dT, ddx = x[0] * dT, ddy = x[1] * dT;
        double ca = cos(a), sa = sin(a);
        double px = c->p_jt[2 * i], py = c->p_jt[2 * i + 1];
        double ux = ca * px - sa * py + ddx;
        double uy = sa * px + ca * py + ddy;
        double wx = c->p_w[2 * i] - tx, wy = c->p_w[2 * i + 1] - ty;
        double ex = ct * wx + st * wy;          /* inv(T) p_w */
        double ey = -st * wx + ct * wy;
        double nx = ex - ux, ny = ey - uy;
        f[2 * i] = c->info_p[0] * log(nx * nx / 2 + 1);
        f[2 * i + 1] = c->info_p[1] * log(ny * ny / 2 + 1);
    }
    const double *I = c->T0inv;
    double m00 = I[0] * ct + I[1] * st, m10 = I[3] * ct + I[4] * st;
    double mdx = I[0] * tx + I[1] * ty + I[2], mdy = I[3] * tx + I[4] * ty + I[5];
    double dth = atan2(m10, m00);
    double vp[3] = {mdx / c->period, mdy / c->period, dth / c->period};
    double d0 = x[0] - vp[0], d1 = x[1] - vp[1], d2 = wrap_pi(x[2] - vp[2]);
    double Nn = (double)c->N;
    f[2 * c->N] = c->info_v[0] * (d0 * Nn);
    f[2 * c->N + 1] = c->info_v[1] * (d1 * Nn);
    f[2 * c->N + 2] = c->info_v[2] * (d2 * Nn);
}

static void inv_se2(const double *T, double *I)
{
    /* general 3x3 affine inverse of [[a b x],[c d y],[0 0 1]] */
    double a = T[0], b = T[1], x = T[2], c = T[3], d = T[4], y = T[5];
    double det = a * d - b * c;
    I[0] = d / det; I[1] = -b / det; I[3] = -c / det; I[4] = a / det;
    I[2] = -(I[0] * x + I[1] * y); I[5] = -(I[3] * x + I[4] * y);
    I[6] = 0; I[7] = 0; I[8] = 1;
}

void oracle_time_deltas(const double *pts, int N, double period, double *dT)
{
    for (int i = 0; i < N; i++) dT[i] = period * atan2(-pts[2 * i + 1], -pts[2 * i]) / (2 * M_PI);
}

/* undistort (N,2) points with velocity v: out (N,2) */
void oracle_undistort(const double *v, const double *pts, int N, double period, double *out)
{
    for (int i = 0; i < N; i++) {
        double dT = period * atan2(-pts[2 * i + 1], -pts[2 * i]) / (2 * M_PI);
        double a = v[2] * dT, ca = cos(a), sa = sin(a);
        out[2 * i] = ca * pts[2 * i] - sa * pts[2 * i + 1] + v[0] * dT;
        out[2 * i + 1] = sa * pts[2 * i] + ca * pts[2 * i + 1] + v[1] * dT;
    }
}

/* sigma5 = [sp_x, sp_y, sv_x, sv_y, sv_th] (the covariance diagonals; weights = 1/sigma)
 * x0_out (optional, 6) = the start vector; r0_out (optional, 2N+3) = residual at x0 */
int oracle_mds_solve(const double *T_wj0, const double *p_w, const double *p_jt, int N,
                     const double *T_init, const double *sigma5, double period,
                     double *out6, int *nfev, double *x0_out, double *r0_out)
{
    Mds c;
    c.N = N; c.p_w = p_w; c.p_jt = p_jt; c.period = period;
    c.dT = (double *)malloc(sizeof(double) * (size_t)(N > 0 ? N : 1));
    oracle_time_deltas(p_jt, N, period, c.dT);
    inv_se2(T_wj0, c.T0inv);
    c.info_p[0] = 1 / sigma5[0]; c.info_p[1] = 1 / sigma5[1];
    c.info_v[0] = 1 / sigma5[2]; c.info_v[1] = 1 / sigma5[3]; c.info_v[2] = 1 / sigma5[4];
    const double *I = c.T0inv;
    double x[6];
    double r00 = I[0] * T_init[0] + I[1] * T_init[3], r10 = I[3] * T_init[0] + I[4] * T_init[3];
    x[0] = (I[0] * T_init[2] + I[1] * T_init[5] + I[2]) / period;
    x[1] = (I[3] * T_init[2] + I[4] * T_init[5] + I[5]) / period;
    x[2] = atan2(r10, r00) / period;
    x[3] = T_init[2]; x[4] = T_init[5]; x[5] = atan2(T_init[3], T_init[0]);
    if (x0_out) memcpy(x0_out, x, sizeof(x));
    if (r0_out) mds_resid(x, r0_out, &c);
    int info = oracle_lmdif(mds_resid, &c, 2 * N + 3, 6, x, 1e-8, 1e-8, 1e-8, 100 * 6 * 7,
                            EPSMCH, 100.0, nfev);
    memcpy(out6, x, sizeof(x));
    free(c.dT);
    return info;
}
