// 2-D Kabsch rigid fit (a10) and motion-distortion Levenberg-Marquardt (a11-a14).
//
// kabsch_kernel replaces getTransformKLT.calculateTransformSVD (reference
//   getTransformKLT.py:129-162): src ~= R tgt + h.  The 2x2 SVD + reflection fix of the
//   reference equals the closed-form proper rotation theta = atan2(C10-C01, C00+C11) of the
//   cross-covariance C = sum (s-ms)(t-mt)^T; means and C are wave/block reductions in f64.
// mds_lm_kernel replaces MotionDistortionSolver.update_problem + optimize_library
//   (motionDistortion.py:80-124,162-205,295-325): MINPACK lmdif (forward-difference
//   Jacobian, pivoted Householder QR, More's lmpar) restated for one 256-thread workgroup
//   per problem: residual rows, Jacobian columns, Householder reflections and norms are
//   row-parallel with block reductions; the 6x6 trust-region algebra (lmpar/qrsolv) is
//   serial on thread 0.  Working set (fvec, 6 Jacobian columns, 2 scratch columns, dT)
//   lives in LDS when it fits 64 KB (N <= 400) and in an L2-resident slab otherwise.
//   float64 throughout; no MFMA (largest matrix is (2N+3) x 6).
#include "roam_internal.h"

#define EPSMCH 2.220446049250313e-16
#define DWARF 2.2250738585072014e-308
#define LM_TMAX 256
#define TWO_PI 6.283185307179586476925286766559

#include "kabsch_body.inc"

__global__ __launch_bounds__(256) void kabsch_kernel(const double *__restrict__ src, const double *__restrict__ tgt,
                                                     const int32_t *__restrict__ count, int N, int nstride,
                                                     double *__restrict__ out6)
{
    __shared__ double red[8];
    const int b = blockIdx.x;
    kabsch_body(src + (int64_t)b * nstride * 2, tgt + (int64_t)b * nstride * 2, count ? count[b] : N, red, out6 + (int64_t)b * 6);
}

hipError_t launch_kabsch(hipStream_t st, const double *src, const double *tgt, const int32_t *count,
                         int N, int nstride, int B, double *out6)
{
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(kabsch_kernel, dim3(B), dim3(256), 0, st, src, tgt, count, N, nstride, out6);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ MDS / LM
struct LmShared {
    double x[6], xtry[6], diag[6], qtf[6], wa1[6], wa2[6], wa3[6], p[6], sd[6], tmp[6];
    double T0inv[6];
    double info_p[2], info_v[3];
    double par, delta, xnorm, fnorm, fnorm1, gnorm, pnorm, ratio, period;
    int ipvt[6];
    int info, nfev, iter, cont_inner, accept, N;
    double red[8];
};

__device__ __forceinline__ double enorm6(const double *v)
{
    double s = 0;
    for (int i = 0; i < 6; i++) s += v[i] * v[i];
    return sqrt(s);
}

__device__ __forceinline__ double wrap_pi(double a)
{
    double r = fmod(a + M_PI, TWO_PI);
    if (r < 0) r += TWO_PI;
    return r - M_PI;
}

// the three velocity-prior residuals of error_vector for parameters x (ct, st = cos / sin of x[5])
__device__ __forceinline__ void mds_vel_resid(const LmShared *S, const double *x, double ct, double st, double *out3)
{
    const double tx = x[3], ty = x[4];
    const double *I = S->T0inv;
    const double m00 = I[0] * ct + I[1] * st, m10 = I[3] * ct + I[4] * st;
    const double mdx = I[0] * tx + I[1] * ty + I[2], mdy = I[3] * tx + I[4] * ty + I[5];
    const double dth = atan2(m10, m00);
    const double d0 = x[0] - mdx / S->period, d1 = x[1] - mdy / S->period;
    const double d2 = wrap_pi(x[2] - dth / S->period);
    const double Nn = (double)S->N;
    out3[0] = S->info_v[0] * (d0 * Nn);
    out3[1] = S->info_v[1] * (d1 * Nn);
    out3[2] = S->info_v[2] * (d2 * Nn);
}

// error_vector (motionDistortion.py:162-205).  Thread t owns points t, t+256, ...; thread 0
// also owns the three velocity residuals.  No barrier inside.
__device__ void mds_resid(const LmShared *S, const double *x, const double *p_w, const double *p_jt,
                          const double *dT, double *f)
{
    const int N = S->N;
    const double th = x[5], tx = x[3], ty = x[4];
    const double ct = cos(th), st = sin(th);
    for (int i = threadIdx.x; i < N; i += (int)blockDim.x) {
        const double d = dT[i];
        const double a = x[2] * d, ddx = x[0] * d, ddy = x[1] * d;
        const double ca = cos(a), sa = sin(a);
        const double px = p_jt[2 * i], py = p_jt[2 * i + 1];
        const double ux = ca * px - sa * py + ddx;
        const double uy = sa * px + ca * py + ddy;
        const double wx = p_w[2 * i] - tx, wy = p_w[2 * i + 1] - ty;
        const double ex = ct * wx + st * wy;
        const double ey = -st * wx + ct * wy;
        const double nx = ex - ux, ny = ey - uy;
        f[2 * i] = S->info_p[0] * log(nx * nx / 2 + 1);
        f[2 * i + 1] = S->info_p[1] * log(ny * ny / 2 + 1);
    }
    if (threadIdx.x == 0) mds_vel_resid(S, x, ct, st, f + 2 * N);
}

#define A_(i, j) a[(size_t)(j) * m + (i)]

// fdjac2 for error_vector: the six forward-difference columns (x[j] -> xp[j] = x[j] + h[j]) of the rows
// this thread owns.  Every column repeats the operations mds_resid would perform on the perturbed vector,
// but subexpressions a perturbation leaves untouched are evaluated once (cos/sin of the rotation angles,
// the rotated points, ...) - same operations on the same operands, so the quotients are bit-identical to
// six full evaluations, and a component a parameter does not enter differences to exactly +0.  The
// transcendental count per point drops from 6 sincos + 12 log to 2 sincos + 10 log, and the ten logs are
// independent (the solve is latency-bound).  Thread blockDim-1-j also takes the velocity rows of column j.
__device__ void mds_jac(const LmShared *S, const double *x, const double *xp, const double *h, const double *p_w,
                        const double *p_jt, const double *dT, const double *fvec, double *a, int m)
{
    const int N = S->N;
    const double tx = x[3], ty = x[4];
    const double ct0 = cos(x[5]), st0 = sin(x[5]);
    const double ct5 = cos(xp[5]), st5 = sin(xp[5]);
    const double ip0 = S->info_p[0], ip1 = S->info_p[1];
    for (int i = threadIdx.x; i < N; i += (int)blockDim.x) {
        const double d = dT[i];
        const double px = p_jt[2 * i], py = p_jt[2 * i + 1];
        const double pwx = p_w[2 * i], pwy = p_w[2 * i + 1];
        const double f0x = fvec[2 * i], f0y = fvec[2 * i + 1];
        const double a0 = x[2] * d, a2 = xp[2] * d;
        const double ca0 = cos(a0), sa0 = sin(a0), ca2 = cos(a2), sa2 = sin(a2);
        const double rx0 = ca0 * px - sa0 * py, ry0 = sa0 * px + ca0 * py;
        const double ux0 = rx0 + x[0] * d, uy0 = ry0 + x[1] * d;
        const double wx0 = pwx - tx, wy0 = pwy - ty;
        const double ex0 = ct0 * wx0 + st0 * wy0, ey0 = -st0 * wx0 + ct0 * wy0;
        double nx, ny;
        // column 0 (v_x): only the x component moves
        nx = ex0 - (rx0 + xp[0] * d);
        A_(2 * i, 0) = (ip0 * log(nx * nx / 2 + 1) - f0x) / h[0];
        A_(2 * i + 1, 0) = 0.0;
        // column 1 (v_y)
        ny = ey0 - (ry0 + xp[1] * d);
        A_(2 * i, 1) = 0.0;
        A_(2 * i + 1, 1) = (ip1 * log(ny * ny / 2 + 1) - f0y) / h[1];
        // column 2 (omega)
        nx = ex0 - ((ca2 * px - sa2 * py) + x[0] * d);
        ny = ey0 - ((sa2 * px + ca2 * py) + x[1] * d);
        A_(2 * i, 2) = (ip0 * log(nx * nx / 2 + 1) - f0x) / h[2];
        A_(2 * i + 1, 2) = (ip1 * log(ny * ny / 2 + 1) - f0y) / h[2];
        // column 3 (t_x)
        {
            const double wx = pwx - xp[3];
            nx = (ct0 * wx + st0 * wy0) - ux0;
            ny = (-st0 * wx + ct0 * wy0) - uy0;
            A_(2 * i, 3) = (ip0 * log(nx * nx / 2 + 1) - f0x) / h[3];
            A_(2 * i + 1, 3) = (ip1 * log(ny * ny / 2 + 1) - f0y) / h[3];
        }
        // column 4 (t_y)
        {
            const double wy = pwy - xp[4];
            nx = (ct0 * wx0 + st0 * wy) - ux0;
            ny = (-st0 * wx0 + ct0 * wy) - uy0;
            A_(2 * i, 4) = (ip0 * log(nx * nx / 2 + 1) - f0x) / h[4];
            A_(2 * i + 1, 4) = (ip1 * log(ny * ny / 2 + 1) - f0y) / h[4];
        }
        // column 5 (theta)
        nx = (ct5 * wx0 + st5 * wy0) - ux0;
        ny = (-st5 * wx0 + ct5 * wy0) - uy0;
        A_(2 * i, 5) = (ip0 * log(nx * nx / 2 + 1) - f0x) / h[5];
        A_(2 * i + 1, 5) = (ip1 * log(ny * ny / 2 + 1) - f0y) / h[5];
    }
    const int jv = (int)blockDim.x - 1 - (int)threadIdx.x;      // velocity rows of column jv
    if (jv < 6) {
        double xv[6], out3[3];
#pragma unroll
        for (int k = 0; k < 6; k++) xv[k] = (k == jv) ? xp[k] : x[k];
        mds_vel_resid(S, xv, jv == 5 ? ct5 : ct0, jv == 5 ? st5 : st0, out3);
        for (int k = 0; k < 3; k++) A_(2 * N + k, jv) = (out3[k] - fvec[2 * N + k]) / h[jv];
    }
}

// serial 6x6 pieces (thread 0 only), operating on the first 6 rows of `a`
__device__ void qrsolv6(int m, double *a, const int *ipvt, const double *diag, const double *qtb,
                        double *x, double *sdiag, double *wa)
{
    const int n = 6;
    for (int j = 0; j < n; j++) {
        for (int i = j; i < n; i++) A_(i, j) = A_(j, i);
        x[j] = A_(j, j); wa[j] = qtb[j];
    }
    for (int j = 0; j < n; j++) {
        const int l = ipvt[j];
        if (diag[l] != 0) {
            for (int k = j; k < n; k++) sdiag[k] = 0;
            sdiag[j] = diag[l];
            double qtbpj = 0;
            for (int k = j; k < n; k++) {
                if (sdiag[k] == 0) continue;
                double c, s;
                if (fabs(A_(k, k)) < fabs(sdiag[k])) {
                    double cot = A_(k, k) / sdiag[k];
                    s = 0.5 / sqrt(0.25 + 0.25 * (cot * cot)); c = s * cot;
                } else {
                    double t = sdiag[k] / A_(k, k);
                    c = 0.5 / sqrt(0.25 + 0.25 * (t * t)); s = c * t;
                }
                A_(k, k) = c * A_(k, k) + s * sdiag[k];
                double temp = c * wa[k] + s * qtbpj;
                qtbpj = -s * wa[k] + c * qtbpj;
                wa[k] = temp;
                for (int i = k + 1; i < n; i++) {
                    temp = c * A_(i, k) + s * sdiag[i];
                    sdiag[i] = -s * A_(i, k) + c * sdiag[i];
                    A_(i, k) = temp;
                }
            }
        }
        sdiag[j] = A_(j, j);
        A_(j, j) = x[j];
    }
    int nsing = n;
    for (int j = 0; j < n; j++) {
        if (sdiag[j] == 0 && nsing == n) nsing = j;
        if (nsing < n) wa[j] = 0;
    }
    for (int k = 0; k < nsing; k++) {
        const int j = nsing - 1 - k;
        double sum = 0;
        for (int i = j + 1; i < nsing; i++) sum += A_(i, j) * wa[i];
        wa[j] = (wa[j] - sum) / sdiag[j];
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa[j];
}

__device__ void lmpar6(int m, double *a, const int *ipvt, const double *diag, const double *qtb,
                       double delta, double *par, double *x, double *sdiag, double *wa1, double *wa2)
{
    const int n = 6;
    int nsing = n;
    for (int j = 0; j < n; j++) {
        wa1[j] = qtb[j];
        if (A_(j, j) == 0 && nsing == n) nsing = j;
        if (nsing < n) wa1[j] = 0;
    }
    for (int k = 0; k < nsing; k++) {
        const int j = nsing - 1 - k;
        wa1[j] /= A_(j, j);
        const double temp = wa1[j];
        for (int i = 0; i < j; i++) wa1[i] -= A_(i, j) * temp;
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa1[j];
    int iter = 0;
    for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm6(wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) { *par = 0; return; }
    double parl = 0;
    if (nsing >= n) {
        for (int j = 0; j < n; j++) { const int l = ipvt[j]; wa1[j] = diag[l] * (wa2[l] / dxnorm); }
        for (int j = 0; j < n; j++) {
            double sum = 0;
            for (int i = 0; i < j; i++) sum += A_(i, j) * wa1[i];
            wa1[j] = (wa1[j] - sum) / A_(j, j);
        }
        const double temp = enorm6(wa1);
        parl = ((fp / delta) / temp) / temp;
    }
    for (int j = 0; j < n; j++) {
        double sum = 0;
        for (int i = 0; i <= j; i++) sum += A_(i, j) * qtb[i];
        wa1[j] = sum / diag[ipvt[j]];
    }
    const double gnorm = enorm6(wa1);
    double paru = gnorm / delta;
    if (paru == 0) paru = DWARF / (delta < 0.1 ? delta : 0.1);
    if (*par < parl) *par = parl;
    if (*par > paru) *par = paru;
    if (*par == 0) *par = gnorm / dxnorm;
    for (;;) {
        iter++;
        if (*par == 0) { const double t = 0.001 * paru; *par = DWARF > t ? DWARF : t; }
        double temp = sqrt(*par);
        for (int j = 0; j < n; j++) wa1[j] = temp * diag[j];
        qrsolv6(m, a, ipvt, wa1, qtb, x, sdiag, wa2);
        for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
        dxnorm = enorm6(wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0 && fp <= temp && temp < 0) || iter == 10) break;
        for (int j = 0; j < n; j++) { const int l = ipvt[j]; wa1[j] = diag[l] * (wa2[l] / dxnorm); }
        for (int j = 0; j < n; j++) {
            wa1[j] /= sdiag[j];
            const double t = wa1[j];
            for (int i = j + 1; i < n; i++) wa1[i] -= A_(i, j) * t;
        }
        temp = enorm6(wa1);
        const double parc = ((fp / delta) / temp) / temp;
        if (fp > 0 && *par > parl) parl = *par;
        if (fp < 0 && *par < paru) paru = *par;
        const double np_ = *par + parc;
        *par = parl > np_ ? parl : np_;
    }
}

#include "lm_wave.inc"

#define LM_LDS_BYTES 65536

// 3 wavefronts per SIMD (<= 168 VGPRs, a few values spill): alone the solve is 15 % slower than at 219 VGPRs, but in
// the pipelined engine it shares every SIMD with the warp / pyramid / tracker of other steps for its whole (latency-
// bound) duration, and two 219-register wavefronts left room for nothing else (A/B: +1.9 % scan-pairs/s)
#define LM_WPE 3
__device__ __forceinline__ void mds_lm_block_solve(const MdsProblemDesc &P, const int b, double *__restrict__ work_g,
                                                   double *__restrict__ out6, int32_t *__restrict__ nfev_out,
                                                   int32_t *__restrict__ info_out, double *__restrict__ x0_out,
                                                   double *__restrict__ r0_out, int lds_bytes, int wave_rows, unsigned char *lm_smem, LmShared &S)
{
    const int t = threadIdx.x;
    const int N = P.count ? min(P.count[b], P.nmax) : P.N;
    const int m = 2 * N + 3, n = 6;
    if (N + 2 <= wave_rows) return;                   // mds_lm_wave_kernel's problem (launch_mds_solve)
    if (N < 2) {                                      // lmdif needs m >= n; the caller keeps the previous pose
        if (t == 0) { nfev_out[b] = 0; info_out[b] = -1; for (int j = 0; j < 6; j++) out6[(size_t)b * 6 + j] = 0.0; }
        return;
    }
    const int mmax = 2 * P.nmax + 3;
    // working set sized by THIS problem's point count: it lives in LDS whenever it fits the dynamic
    // LDS the launch provided (lds_bytes, sized from the host-side bound nmax and capped at 64 KB),
    // otherwise in the L2-resident global slab
    const size_t need = ((size_t)m * 9 + N) * sizeof(double);
    const bool in_lds = need <= (size_t)lds_bytes;
    double *work = in_lds ? reinterpret_cast<double *>(lm_smem) : work_g + (size_t)b * ((size_t)mmax * 9 + P.nmax);
    const int ms = in_lds ? m : mmax;                  // slab stride
    double *fvec = work, *a = work + ms, *wa4 = a + (size_t)6 * ms, *wf = wa4 + ms, *dT = wf + ms;
    // NOTE: columns of `a` are spaced m apart (A_ macro), all inside the 6*ms slab.
    const double *p_w = P.p_w + (size_t)b * P.nstride * 2;
    const double *p_jt = P.p_jt + (size_t)b * P.nstride * 2;
    const double *T0 = P.T_wj0 + (size_t)b * 9, *Ti = P.T_init + (size_t)b * 9;
    double *red = S.red;

    if (t == 0) {
        S.N = N; S.period = P.period;
        const double ta = T0[0], tb = T0[1], tx = T0[2], tc = T0[3], td = T0[4], ty = T0[5];
        const double det = ta * td - tb * tc;
        double *I = S.T0inv;
        I[0] = td / det; I[1] = -tb / det; I[3] = -tc / det; I[4] = ta / det;
        I[2] = -(I[0] * tx + I[1] * ty); I[5] = -(I[3] * tx + I[4] * ty);
        S.info_p[0] = 1 / P.sigma5[0]; S.info_p[1] = 1 / P.sigma5[1];
        S.info_v[0] = 1 / P.sigma5[2]; S.info_v[1] = 1 / P.sigma5[3]; S.info_v[2] = 1 / P.sigma5[4];
        const double r00 = I[0] * Ti[0] + I[1] * Ti[3], r10 = I[3] * Ti[0] + I[4] * Ti[3];
        S.x[0] = (I[0] * Ti[2] + I[1] * Ti[5] + I[2]) / P.period;
        S.x[1] = (I[3] * Ti[2] + I[4] * Ti[5] + I[5]) / P.period;
        S.x[2] = atan2(r10, r00) / P.period;
        S.x[3] = Ti[2]; S.x[4] = Ti[5]; S.x[5] = atan2(Ti[3], Ti[0]);
        for (int j = 0; j < 6; j++) S.diag[j] = 1.0;
        S.info = 0; S.nfev = 1; S.iter = 1; S.par = 0; S.delta = 0; S.xnorm = 0; S.gnorm = 0;
    }
    for (int i = t; i < N; i += (int)blockDim.x) dT[i] = P.period * atan2(-p_jt[2 * i + 1], -p_jt[2 * i]) / TWO_PI;
    __syncthreads();
    double xl[6];
#pragma unroll
    for (int j = 0; j < 6; j++) xl[j] = S.x[j];
    mds_resid(&S, xl, p_w, p_jt, dT, fvec);
    __syncthreads();
    if (x0_out && t < 6) x0_out[(size_t)b * 6 + t] = S.x[t];
    if (r0_out) for (int i = t; i < m; i += (int)blockDim.x) r0_out[(size_t)b * mmax + i] = fvec[i];
    {
        double s = 0;
        for (int i = t; i < m; i += (int)blockDim.x) s += fvec[i] * fvec[i];
        s = block_sum_d(s, red);
        if (t == 0) S.fnorm = sqrt(s);
        __syncthreads();
    }
    const double eps = sqrt(EPSMCH);            // epsfcn = EPS -> sqrt(max(epsfcn, epsmch))
#ifdef LM_TIMING
    unsigned long long tk[6] = {0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime(), tn;
#define LM_TICK(k) { tn = __builtin_amdgcn_s_memtime(); tk[k] += tn - tl; tl = tn; }
#else
#define LM_TICK(k)
#endif

    for (;;) {   // ------------------------------------------------ outer loop
        // forward-difference Jacobian; rows are owned by fixed threads (no barrier needed
        // between the perturbed evaluation and the difference quotient)
#pragma unroll
        for (int j = 0; j < 6; j++) xl[j] = S.x[j];
        double xp[6], hh[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double h = eps * fabs(xl[j]);
            if (h == 0) h = eps;
            hh[j] = h;
            xp[j] = xl[j] + h;
        }
        LM_TICK(0)
        mds_jac(&S, xl, xp, hh, p_w, p_jt, dT, fvec, a, m);
        __syncthreads();
        LM_TICK(1)
        // ---- qrfac with column pivoting
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int i = t; i < m; i += (int)blockDim.x) s += A_(i, j) * A_(i, j);
            s = block_sum_d(s, red);
            if (t == 0) { S.wa2[j] = sqrt(s); S.wa1[j] = S.wa2[j]; S.wa3[j] = S.wa2[j]; S.ipvt[j] = j; }
        }
        if (t == 0) { S.nfev += n; }
        __syncthreads();
        for (int j = 0; j < n; j++) {
            int kmax = j;
            for (int k = j; k < n; k++) if (S.wa1[k] > S.wa1[kmax]) kmax = k;
            __syncthreads();
            if (kmax != j) {
                for (int i = t; i < m; i += (int)blockDim.x) { double tmp = A_(i, j); A_(i, j) = A_(i, kmax); A_(i, kmax) = tmp; }
                if (t == 0) {
                    S.wa1[kmax] = S.wa1[j]; S.wa3[kmax] = S.wa3[j];
                    int tp = S.ipvt[j]; S.ipvt[j] = S.ipvt[kmax]; S.ipvt[kmax] = tp;
                }
                __syncthreads();
            }
            double s = 0;
            for (int i = j + t; i < m; i += (int)blockDim.x) s += A_(i, j) * A_(i, j);
            s = block_sum_d(s, red);
            double ajnorm = sqrt(s);
            if (ajnorm != 0) {
                if (A_(j, j) < 0) ajnorm = -ajnorm;
                __syncthreads();
                for (int i = j + t; i < m; i += (int)blockDim.x) A_(i, j) /= ajnorm;
                __syncthreads();
                if (t == 0) A_(j, j) += 1;
                __syncthreads();
                for (int k = j + 1; k < n; k++) {
                    double sum = 0;
                    for (int i = j + t; i < m; i += (int)blockDim.x) sum += A_(i, j) * A_(i, k);
                    sum = block_sum_d(sum, red);
                    const double temp = sum / A_(j, j);
                    __syncthreads();
                    for (int i = j + t; i < m; i += (int)blockDim.x) A_(i, k) -= temp * A_(i, j);
                    __syncthreads();
                    // rdiag down-date (uniform decisions from shared values)
                    double rk = S.wa1[k];
                    int recompute = 0;
                    if (rk != 0) {
                        const double tq = A_(j, k) / rk;
                        const double dd = 1 - tq * tq;
                        rk *= sqrt(dd > 0 ? dd : 0);
                        const double q = rk / S.wa3[k];
                        if (0.05 * (q * q) <= EPSMCH) recompute = 1;
                    }
                    double s2 = 0;
                    if (recompute) {
                        for (int i = j + 1 + t; i < m; i += (int)blockDim.x) s2 += A_(i, k) * A_(i, k);
                        s2 = block_sum_d(s2, red);
                        rk = sqrt(s2);
                    }
                    __syncthreads();
                    if (t == 0) { S.wa1[k] = rk; if (recompute) S.wa3[k] = rk; }
                    __syncthreads();
                }
            }
            if (t == 0) S.wa1[j] = -ajnorm;
            __syncthreads();
        }
        LM_TICK(2)
        if (t == 0 && S.iter == 1) {
            for (int j = 0; j < n; j++) S.tmp[j] = S.diag[j] * S.x[j];
            S.xnorm = enorm6(S.tmp);
            S.delta = 100.0 * S.xnorm;
            if (S.delta == 0) S.delta = 100.0;
        }
        // ---- Q^T fvec -> qtf
        for (int i = t; i < m; i += (int)blockDim.x) wa4[i] = fvec[i];
        __syncthreads();
        for (int j = 0; j < n; j++) {
            const double ajj = A_(j, j);
            if (ajj != 0) {
                double sum = 0;
                for (int i = j + t; i < m; i += (int)blockDim.x) sum += A_(i, j) * wa4[i];
                sum = block_sum_d(sum, red);
                const double temp = -sum / ajj;
                __syncthreads();
                for (int i = j + t; i < m; i += (int)blockDim.x) wa4[i] += A_(i, j) * temp;
            }
            __syncthreads();
            if (t == 0) { A_(j, j) = S.wa1[j]; S.qtf[j] = wa4[j]; }
            __syncthreads();
        }
        // ---- gradient norm + inner loop prologue (thread 0)
        if (t == 0) {
            double gnorm = 0;
            if (S.fnorm != 0)
                for (int j = 0; j < n; j++) {
                    const int l = S.ipvt[j];
                    if (S.wa2[l] != 0) {
                        double sum = 0;
                        for (int i = 0; i <= j; i++) sum += A_(i, j) * (S.qtf[i] / S.fnorm);
                        const double g = fabs(sum / S.wa2[l]);
                        if (g > gnorm) gnorm = g;
                    }
                }
            S.gnorm = gnorm;
            if (gnorm <= 1e-8) S.info = 4;
        }
        __syncthreads();
        LM_TICK(3)
        if (S.info != 0) break;
        for (;;) {   // -------------------------------------------- inner loop
            if (t == 0) {
                lmpar6(m, a, S.ipvt, S.diag, S.qtf, S.delta, &S.par, S.p, S.sd, S.wa3, S.tmp);
                for (int j = 0; j < n; j++) {
                    S.wa1[j] = -S.p[j];
                    S.xtry[j] = S.x[j] + S.wa1[j];
                    S.wa3[j] = S.diag[j] * S.wa1[j];
                }
                S.pnorm = enorm6(S.wa3);
                if (S.iter == 1 && S.pnorm < S.delta) S.delta = S.pnorm;
            }
            __syncthreads();
            LM_TICK(4)
#pragma unroll
            for (int j = 0; j < 6; j++) xl[j] = S.xtry[j];
            mds_resid(&S, xl, p_w, p_jt, dT, wa4);
            __syncthreads();
            double s = 0;
            for (int i = t; i < m; i += (int)blockDim.x) s += wa4[i] * wa4[i];
            s = block_sum_d(s, red);
            if (t == 0) {
                S.nfev++;
                const double fnorm1 = sqrt(s), fnorm = S.fnorm;
                S.fnorm1 = fnorm1;
                double actred = -1;
                if (0.1 * fnorm1 < fnorm) { const double r = fnorm1 / fnorm; actred = 1 - r * r; }
                for (int j = 0; j < n; j++) {
                    S.wa3[j] = 0;
                    const double temp = S.wa1[S.ipvt[j]];
                    for (int i = 0; i <= j; i++) S.wa3[i] += A_(i, j) * temp;
                }
                const double temp1 = enorm6(S.wa3) / fnorm;
                const double temp2 = (sqrt(S.par) * S.pnorm) / fnorm;
                const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
                const double dirder = -(temp1 * temp1 + temp2 * temp2);
                double ratio = 0;
                if (prered != 0) ratio = actred / prered;
                if (ratio <= 0.25) {
                    double temp;
                    if (actred >= 0) temp = 0.5;
                    else temp = 0.5 * dirder / (dirder + 0.5 * actred);
                    if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                    const double dm = S.pnorm / 0.1;
                    S.delta = temp * (S.delta < dm ? S.delta : dm);
                    S.par /= temp;
                } else if (S.par == 0 || ratio >= 0.75) {
                    S.delta = S.pnorm / 0.5;
                    S.par *= 0.5;
                }
                S.accept = 0;
                if (ratio >= 1e-4) {
                    for (int j = 0; j < n; j++) { S.x[j] = S.xtry[j]; S.wa2[j] = S.diag[j] * S.x[j]; }
                    S.xnorm = enorm6(S.wa2);
                    S.fnorm = fnorm1;
                    S.iter++;
                    S.accept = 1;
                }
                int info = 0;
                if (fabs(actred) <= 1e-8 && prered <= 1e-8 && 0.5 * ratio <= 1) info = 1;
                if (S.delta <= 1e-8 * S.xnorm) info = 2;
                if (fabs(actred) <= 1e-8 && prered <= 1e-8 && 0.5 * ratio <= 1 && info == 2) info = 3;
                if (info == 0) {
                    if (S.nfev >= 100 * 6 * 7) info = 5;
                    if (fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1) info = 6;
                    if (S.delta <= EPSMCH * S.xnorm) info = 7;
                    if (S.gnorm <= EPSMCH) info = 8;
                }
                S.info = info;
                S.ratio = ratio;
            }
            __syncthreads();
            if (S.accept) for (int i = t; i < m; i += (int)blockDim.x) fvec[i] = wa4[i];
            const int info = S.info;
            const double ratio = S.ratio;
            __syncthreads();
            LM_TICK(5)
            if (info != 0 || !(ratio < 1e-4)) break;
        }
        if (S.info != 0) break;
    }
#ifdef LM_TIMING
    if (x0_out && t == 0) for (int k = 0; k < 6; k++) x0_out[(size_t)b * 6 + k] = (double)tk[k];
#endif
    if (t < 6) out6[(size_t)b * 6 + t] = S.x[t];
    if (t == 0) { nfev_out[b] = S.nfev; info_out[b] = S.info; }
}

// one workgroup per problem - or, when the wave form has listed the problems it left (P.big), a few workgroups walking that list: with
// a bound of 320 points and typical problems of 150, the 4 096 workgroups of a step that only had to return cost 0.46 ms on the
// step's critical path (each waits for 32 KB of LDS beside the warp of the next step before it can do so)
__global__ __launch_bounds__(LM_TMAX, LM_WPE) void mds_lm_kernel(MdsProblemDesc P, double *__restrict__ work_g,
                                                      double *__restrict__ out6, int32_t *__restrict__ nfev_out,
                                                      int32_t *__restrict__ info_out, double *__restrict__ x0_out,
                                                      double *__restrict__ r0_out, int lds_bytes, int wave_rows)
{
    extern __shared__ __align__(16) unsigned char lm_smem[];
    __shared__ LmShared S;
    if (!P.big || wave_rows == 0) {
        mds_lm_block_solve(P, blockIdx.x, work_g, out6, nfev_out, info_out, x0_out, r0_out, lds_bytes, wave_rows, lm_smem, S);
        return;
    }
    const int32_t *L = P.big + (size_t)P.big_slot * (1 + P.B);
    const int nbig = L[0];
    for (int q = blockIdx.x; q < nbig; q += (int)gridDim.x) {
        mds_lm_block_solve(P, L[1 + q], work_g, out6, nfev_out, info_out, x0_out, r0_out, lds_bytes, wave_rows, lm_smem, S);
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) P.big[(size_t)(P.big_slot ^ 1) * (1 + P.B)] = 0;     // the next solve's list (its wave kernel runs after this one)
}

hipError_t launch_mds_solve(hipStream_t st, const MdsProblemDesc &p, double *work, double *out6,
                            int32_t *nfev, int32_t *info, double *x0_out, double *r0_out)
{
    if (p.B <= 0) return hipSuccess;
    // problems of up to 254 points: one wavefront each, everything in registers (lm_wave.inc); larger ones: the workgroup form below.
    // Only the device knows a problem's size, so both kernels are launched when the host-side bound allows both, and each returns
    // at once from the other's problems.  ROAM_LM_BLOCK=1 keeps the workgroup form for everything (A/B).
    static const bool block_form = getenv("ROAM_LM_BLOCK") != nullptr && atoi(getenv("ROAM_LM_BLOCK")) != 0;
    int wave_rows = 0;                                // N + 2 <= wave_rows: the wave kernel's
    static const int wpe_env = getenv("ROAM_LM_WPE") ? atoi(getenv("ROAM_LM_WPE")) : 0;
    const int wpe = wpe_env == 1 ? 1 : 2;
    MdsProblemDesc pw = p;
    hipStream_t st_block = st;
    bool join = false;
    if (!block_form) {
        const int ppt = std::min(4, (p.nmax + 2 + 63) / 64);
        wave_rows = 64 * ppt;
        const bool both = p.nmax + 2 > wave_rows;
        // a batch's workgroup form BESIDE the wave form, on the caller's side stream: behind it, the few problems above 254 points were
        // 1.3 ms of the back end's chain in every step (each is a 0.5 ms latency chain whatever their number)
        if (both && p.side && p.ev_fork && p.ev_join) {
            pw.big = nullptr;
            hipError_t ef = hipEventRecord(p.ev_fork, st);
            if (ef == hipSuccess) ef = hipStreamWaitEvent(p.side, p.ev_fork, 0);
            if (ef != hipSuccess) return ef;
            st_block = p.side;
            join = true;
        }
        // register budget: two wavefronts per SIMD (<= 256 registers; the solve wants ~290).  4 096 solves alone: 1.46-1.93 ms against
        // 1.93-2.51 unconstrained (one wave per SIMD), 1.9-2.4 at three, 2.1-2.7 at four, 3.56-4.48 for the workgroup form; a lone
        // solve is as fast either way (profiles/r06_lm_experiments.txt).  ROAM_LM_WPE=1: unconstrained (A/B)
#define LMW_LAUNCH(PPT_, WPE_) hipLaunchKernelGGL((mds_lm_wave_kernel<PPT_, WPE_>), dim3(p.B), dim3(64), 0, st, pw, out6, nfev, info, x0_out, r0_out)
#define LMW_LAUNCH_W(PPT_) { if (wpe == 2) LMW_LAUNCH(PPT_, 2); else LMW_LAUNCH(PPT_, 1); }
        if (!join) {
            switch (ppt) {
            case 1: LMW_LAUNCH_W(1) break;
            case 2: LMW_LAUNCH_W(2) break;
            case 3: LMW_LAUNCH_W(3) break;
            default: LMW_LAUNCH_W(4) break;
            }
            hipError_t e = hipGetLastError();
            if (e != hipSuccess || !both) return e;
        }
    }
    const size_t mmax = 2 * (size_t)p.nmax + 3;
    size_t need = (mmax * 9 + p.nmax) * sizeof(double);
    // LDS per workgroup caps how many solves a CU runs concurrently, and the solve is latency-bound:
    // 32 KB (N <= 214) keeps 4-5 workgroups per CU; measured 64 KB -> 2 per CU was 30 % slower
    const size_t cap = 32768;
    size_t lds = need <= cap ? need : cap;
    // the solve is a long chain of short reductions: with few points a single wavefront per problem
    // (workgroup barriers degenerate to no-ops, reductions stay in registers) has the lowest latency
    const int threads = p.nmax <= 192 ? 64 : (p.nmax <= 448 ? 128 : LM_TMAX);
    static const int big_grid = getenv("ROAM_LM_BIG_GRID") ? atoi(getenv("ROAM_LM_BIG_GRID")) : 512;
    const int grid = (pw.big && wave_rows) ? std::min(p.B, big_grid) : p.B;
    hipLaunchKernelGGL(mds_lm_kernel, dim3(grid), dim3(threads), lds, st_block, pw, work, out6, nfev, info, x0_out, r0_out, (int)lds, wave_rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !join) return e;
    if ((e = hipEventRecord(p.ev_join, st_block)) != hipSuccess) return e;
    LMW_LAUNCH_W(4)                                   // (both forms are needed: the bound is above 254 points, four slots)
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return hipStreamWaitEvent(st, p.ev_join, 0);
}

__global__ __launch_bounds__(256) void mds_undistort_kernel(const double *__restrict__ v3, const double *__restrict__ pts,
                                                            int N, double period, double *__restrict__ out_xy,
                                                            double *__restrict__ dT_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double x = pts[2 * i], y = pts[2 * i + 1];
    const double dT = period * atan2(-y, -x) / TWO_PI;
    const double a = v3[2] * dT, ca = cos(a), sa = sin(a);
    if (out_xy) {
        out_xy[2 * i] = ca * x - sa * y + v3[0] * dT;
        out_xy[2 * i + 1] = sa * x + ca * y + v3[1] * dT;
    }
    if (dT_out) dT_out[i] = dT;
}

hipError_t launch_mds_undistort(hipStream_t st, const double *v3, const double *pts, int N,
                                double period, double *out_xy, double *dT)
{
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(mds_undistort_kernel, dim3((N + 255) / 256), dim3(256), 0, st, v3, pts, N, period, out_xy, dT);
    return hipGetLastError();
}
