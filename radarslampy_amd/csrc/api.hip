// C-ABI: context management + "stage" entry points (host arrays in / out, one reference
// function each).  See include/roam_abi.h for the contract and the reference citations.
#include "roam_internal.h"
#include <cstdlib>
#include <new>

extern "C" {

const char *roam_version(void) { return "radarslampy_amd 0.1 (gfx950, HIP)"; }

int32_t roam_create(int32_t device_id, roam_ctx **out)
{
    if (!out) return ROAM_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ROAM_E_NODEVICE;
    if (device_id < 0 || device_id >= ndev) return ROAM_E_ARG;
    roam_ctx *ctx = new (std::nothrow) roam_ctx();
    if (!ctx) return ROAM_E_HIP;
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) { delete ctx; return ROAM_E_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { delete ctx; return ROAM_E_HIP; }
    ctx->cu_count = prop.multiProcessorCount;
    {
        // ROAM_MAIN_PRIORITY=1 (experiment, round 6): the main stream - back end and detection, the critical path of a step - above the
        // front-end streams in the hardware queues
        const char *pv = getenv("ROAM_MAIN_PRIORITY");
        int lo = 0, hi = 0;
        if (pv && pv[0] == '1' && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) {
            if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, hi) != hipSuccess) { delete ctx; return ROAM_E_HIP; }
        } else if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return ROAM_E_HIP; }
    }
    // front-end stream: peaks, warp and pyramid of a step run here so that they can overlap the back end
    // (KLT ... LM) of the previous step; equal priority measured best once the two stages are pipelined
    if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess) { hipStreamDestroy(ctx->stream); delete ctx; return ROAM_E_HIP; }
    if (hipStreamCreateWithFlags(&ctx->stream4, hipStreamNonBlocking) != hipSuccess) { hipStreamDestroy(ctx->stream2); hipStreamDestroy(ctx->stream); delete ctx; return ROAM_E_HIP; }
    if (hipStreamCreateWithFlags(&ctx->stream5, hipStreamNonBlocking) != hipSuccess) { hipStreamDestroy(ctx->stream4); hipStreamDestroy(ctx->stream2); hipStreamDestroy(ctx->stream); delete ctx; return ROAM_E_HIP; }
    if (hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_up, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fence, hipEventDisableTiming) != hipSuccess) {
        delete ctx; return ROAM_E_HIP;
    }
    *out = ctx;
    return ROAM_OK;
}

int32_t roam_engine_destroy(roam_ctx *ctx);

int32_t roam_destroy(roam_ctx *ctx)
{
    if (!ctx) return ROAM_E_ARG;
    hipSetDevice(ctx->device);
    roam_comm_destroy(ctx);
    roam_engine_destroy(ctx);
    hipStreamSynchronize(ctx->stream);
    for (auto &s : ctx->scratch) if (s.p) hipFree(s.p);
    hipStreamSynchronize(ctx->stream2);
    hipStreamSynchronize(ctx->stream4);
    hipStreamSynchronize(ctx->stream5);
    hipStreamSynchronize(ctx->stream3);
    hipEventDestroy(ctx->ev_up); hipEventDestroy(ctx->ev_fence);
    hipStreamDestroy(ctx->stream3);
    hipStreamDestroy(ctx->stream5);
    hipStreamDestroy(ctx->stream4);
    hipStreamDestroy(ctx->stream2);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return ROAM_OK;
}

const char *roam_last_error(const roam_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int32_t roam_device_info(roam_ctx *ctx, char *name, int32_t name_cap, int32_t *cu_count,
                         int64_t *hbm_bytes, char *arch, int32_t arch_cap)
{
    if (!ctx) return ROAM_E_ARG;
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_cap > 0) {
        strncpy(name, prop.name, name_cap - 1); name[name_cap - 1] = 0;
        if (!name[0]) {
            // some driver stacks leave the marketing name empty: say what the properties say (gfx950 with 256 CUs is the MI350 series,
            // its 2.4 GHz part the MI355X)
            const bool mi35x = strncmp(prop.gcnArchName, "gfx950", 6) == 0 && prop.multiProcessorCount == 256;
            snprintf(name, (size_t)name_cap, "%s (%.6s, %d CUs, %.0f GB, %.2f GHz)",
                     mi35x ? (prop.clockRate >= 2300000 ? "AMD Instinct MI355X" : "AMD Instinct MI350X") : "AMD GPU", prop.gcnArchName,
                     prop.multiProcessorCount, (double)prop.totalGlobalMem / 1073741824.0, prop.clockRate * 1e-6);
        }
    }
    if (arch && arch_cap > 0) { strncpy(arch, prop.gcnArchName, arch_cap - 1); arch[arch_cap - 1] = 0; }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return ROAM_OK;
}

int32_t roam_synchronize(roam_ctx *ctx)
{
    if (!ctx) return ROAM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream3));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ROAM_OK;
}

int32_t roam_host_alloc(roam_ctx *ctx, int64_t bytes, void **out)
{
    if (!ctx || !out || bytes <= 0) return ROAM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return ROAM_OK;
}

int32_t roam_host_free(roam_ctx *ctx, void *p)
{
    if (!ctx) return ROAM_E_ARG;
    if (p) HIP_TRY(ctx, hipHostFree(p));
    return ROAM_OK;
}

}  // extern "C"

void *roam_scratch(roam_ctx *ctx, int slot, size_t bytes)
{
    DevBuf &b = ctx->scratch[slot];
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return b.p;
    if (b.p) { hipStreamSynchronize(ctx->stream); hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    size_t want = (bytes + 4095) & ~(size_t)4095;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        ROAM_SET_ERR(ctx, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        b.p = nullptr;
        return nullptr;
    }
    b.bytes = want;
    return b.p;
}

#define SCRATCH(var, type, slot, bytes)                         \
    type *var = (type *)roam_scratch(ctx, slot, bytes);         \
    if (!var) return ROAM_E_HIP

#define H2D(dst, src, bytes) HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream))
#define D2H(dst, src, bytes) HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream))
#define SYNC() HIP_TRY(ctx, hipStreamSynchronize(ctx->stream))
#define ENTER()                                   \
    if (!ctx) return ROAM_E_ARG;                  \
    HIP_TRY(ctx, hipSetDevice(ctx->device))

static int32_t peaks_common(roam_ctx *ctx, PeakSrc src, int rows, int cols, int32_t *out, int64_t cap, int64_t *n_out)
{
    const int stage_cap = (cols + 1) / 2;
    SCRATCH(stage, uint16_t, S_TMP0, sizeof(uint16_t) * (size_t)rows * stage_cap);
    SCRATCH(rcount, int32_t, S_TMP1, sizeof(int32_t) * (size_t)rows);
    const int64_t dcap = (int64_t)rows * stage_cap;
    SCRATCH(dout, int32_t, S_OUT0, sizeof(int32_t) * 2 * (size_t)dcap);
    SCRATCH(dn, int32_t, S_OUT1, sizeof(int32_t));
    HIP_TRY(ctx, launch_peaks(ctx->stream, src, 1, rows, cols, stage, stage_cap, rcount, dout, (int32_t)dcap, dn));
    int32_t n = 0;
    D2H(&n, dn, sizeof(int32_t));
    SYNC();
    *n_out = n;
    const int64_t ncopy = n < cap ? n : cap;
    if (ncopy > 0) { D2H(out, dout, sizeof(int32_t) * 2 * (size_t)ncopy); SYNC(); }
    if (n > cap) { ROAM_SET_ERR(ctx, "peaks: %d found, capacity %lld", n, (long long)cap); return ROAM_E_CAPACITY; }
    return ROAM_OK;
}

extern "C" int32_t roam_peaks_polar_f32(roam_ctx *ctx, const float *polar, int32_t rows, int32_t cols,
                                        int32_t *out, int64_t cap, int64_t *n_out)
{
    ENTER();
    ARG_CHECK(ctx, polar && out && n_out && rows > 0 && cols >= 1 && cols <= ROAM_MAX_COLS && cap >= 0);
    SCRATCH(din, float, S_IN0, sizeof(float) * (size_t)rows * cols);
    H2D(din, polar, sizeof(float) * (size_t)rows * cols);
    PeakSrc src = {din, 0, cols, 0, 0, nullptr};
    return peaks_common(ctx, src, rows, cols, out, cap, n_out);
}

extern "C" int32_t roam_peaks_record_u8(roam_ctx *ctx, const uint8_t *rec, int32_t rows, int64_t stride,
                                        int32_t payload_off, int32_t clip, int32_t *out, int64_t cap,
                                        int64_t *n_out)
{
    ENTER();
    ARG_CHECK(ctx, rec && out && n_out && rows > 0 && clip >= 1 && clip <= ROAM_MAX_COLS && payload_off >= 0 &&
                       stride >= payload_off + clip && cap >= 0);
    SCRATCH(din, uint8_t, S_IN0, (size_t)rows * stride);
    H2D(din, rec, (size_t)rows * stride);
    PeakSrc src = {din, 0, stride, payload_off, 1, nullptr};
    return peaks_common(ctx, src, rows, clip, out, cap, n_out);
}

static int32_t warp_common(roam_ctx *ctx, WarpSrc src, int rows, int cols, float *cart_f32, uint8_t *cart_u8)
{
    const int W = 2 * (cols / 2);
    const size_t npx = (size_t)W * W;
    float *df = nullptr;
    uint8_t *du = nullptr;
    if (cart_f32) { df = (float *)roam_scratch(ctx, S_OUT0, sizeof(float) * npx); if (!df) return ROAM_E_HIP; }
    if (cart_u8) { du = (uint8_t *)roam_scratch(ctx, S_OUT1, npx); if (!du) return ROAM_E_HIP; }
    HIP_TRY(ctx, launch_polar_to_cart(ctx->stream, src, 1, rows, cols, du, (int64_t)npx, df, (int64_t)npx));
    if (cart_f32) D2H(cart_f32, df, sizeof(float) * npx);
    if (cart_u8) D2H(cart_u8, du, npx);
    SYNC();
    return ROAM_OK;
}

extern "C" int32_t roam_polar_to_cart_f32(roam_ctx *ctx, const float *polar, int32_t rows, int32_t cols,
                                          float *cart_f32, uint8_t *cart_u8)
{
    ENTER();
    ARG_CHECK(ctx, polar && rows > 0 && cols >= 2 && (cart_f32 || cart_u8));
    SCRATCH(din, float, S_IN0, sizeof(float) * (size_t)rows * cols);
    H2D(din, polar, sizeof(float) * (size_t)rows * cols);
    WarpSrc src = {din, 0, cols, 0, 0, nullptr};
    return warp_common(ctx, src, rows, cols, cart_f32, cart_u8);
}

extern "C" int32_t roam_polar_to_cart_record_u8(roam_ctx *ctx, const uint8_t *rec, int32_t rows, int64_t stride,
                                                int32_t payload_off, int32_t clip, float *cart_f32,
                                                uint8_t *cart_u8)
{
    ENTER();
    ARG_CHECK(ctx, rec && rows > 0 && clip >= 2 && payload_off >= 0 && stride >= payload_off + clip && (cart_f32 || cart_u8));
    SCRATCH(din, uint8_t, S_IN0, (size_t)rows * stride);
    H2D(din, rec, (size_t)rows * stride);
    WarpSrc src = {din, 0, stride, payload_off, 1, nullptr};
    return warp_common(ctx, src, rows, clip, cart_f32, cart_u8);
}

extern "C" int32_t roam_pyr_down_u8(roam_ctx *ctx, const uint8_t *src, int32_t w, int32_t h, uint8_t *dst)
{
    ENTER();
    ARG_CHECK(ctx, src && dst && w >= 2 && h >= 2);
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    SCRATCH(ds, uint8_t, S_IN0, (size_t)w * h);
    SCRATCH(dd, uint8_t, S_OUT0, (size_t)dw * dh);
    H2D(ds, src, (size_t)w * h);
    HIP_TRY(ctx, launch_pyr_down(ctx->stream, ds, 0, w, h, dd, 0, 1));
    D2H(dst, dd, (size_t)dw * dh);
    SYNC();
    return ROAM_OK;
}

static int32_t klt_common(roam_ctx *ctx, uint8_t *pyrA, uint8_t *pyrB, const PyrDesc &d, const float *pts, int K,
                          float *next_pts, uint8_t *status, float *err)
{
    HIP_TRY(ctx, launch_build_pyramid(ctx->stream, pyrA, d, 1));
    HIP_TRY(ctx, launch_build_pyramid(ctx->stream, pyrB, d, 1));
    SCRATCH(dpts, float, S_IN2, sizeof(float) * 2 * (size_t)K);
    SCRATCH(dnext, float, S_OUT0, sizeof(float) * 2 * (size_t)K);
    SCRATCH(dst, uint8_t, S_OUT1, (size_t)K);
    SCRATCH(derr, float, S_OUT2, sizeof(float) * (size_t)K);
    H2D(dpts, pts, sizeof(float) * 2 * (size_t)K);
    HIP_TRY(ctx, launch_klt(ctx->stream, pyrA, pyrB, d, dpts, nullptr, K, K, 1, dnext, dst, derr));
    D2H(next_pts, dnext, sizeof(float) * 2 * (size_t)K);
    D2H(status, dst, (size_t)K);
    D2H(err, derr, sizeof(float) * (size_t)K);
    SYNC();
    return ROAM_OK;
}

extern "C" int32_t roam_klt_track_u8(roam_ctx *ctx, const uint8_t *prev_img, const uint8_t *next_img,
                                     int32_t w, int32_t h, const float *pts, int32_t K,
                                     float *next_pts, uint8_t *status, float *err)
{
    ENTER();
    ARG_CHECK(ctx, prev_img && next_img && pts && next_pts && status && err && w >= 16 && h >= 16 && K >= 0);
    if (K == 0) return ROAM_OK;
    PyrDesc d;
    pyr_desc_init(&d, w, h);
    SCRATCH(pa, uint8_t, S_PYR_A, (size_t)d.lane_stride);
    SCRATCH(pb, uint8_t, S_PYR_B, (size_t)d.lane_stride);
    H2D(pa, prev_img, (size_t)w * h);
    H2D(pb, next_img, (size_t)w * h);
    return klt_common(ctx, pa, pb, d, pts, K, next_pts, status, err);
}

extern "C" int32_t roam_klt_track_f32(roam_ctx *ctx, const float *prev_img, const float *next_img,
                                      int32_t w, int32_t h, const float *pts, int32_t K,
                                      float *next_pts, uint8_t *status, float *err)
{
    ENTER();
    ARG_CHECK(ctx, prev_img && next_img && pts && next_pts && status && err && w >= 16 && h >= 16 && K >= 0);
    if (K == 0) return ROAM_OK;
    PyrDesc d;
    pyr_desc_init(&d, w, h);
    const size_t npx = (size_t)w * h;
    SCRATCH(pa, uint8_t, S_PYR_A, (size_t)d.lane_stride);
    SCRATCH(pb, uint8_t, S_PYR_B, (size_t)d.lane_stride);
    SCRATCH(fa, float, S_IN0, sizeof(float) * npx);
    SCRATCH(fb, float, S_IN1, sizeof(float) * npx);
    H2D(fa, prev_img, sizeof(float) * npx);
    H2D(fb, next_img, sizeof(float) * npx);
    HIP_TRY(ctx, launch_quantize_u8(ctx->stream, fa, (int64_t)npx, pa));
    HIP_TRY(ctx, launch_quantize_u8(ctx->stream, fb, (int64_t)npx, pb));
    return klt_common(ctx, pa, pb, d, pts, K, next_pts, status, err);
}

extern "C" int32_t roam_reject_outliers(roam_ctx *ctx, const float *prev, const float *next, int32_t K,
                                        double thr_px, int64_t node_limit, uint8_t *mask_out,
                                        int32_t *n_inliers, int32_t *flags_out, uint64_t *adj_out)
{
    ENTER();
    ARG_CHECK(ctx, K >= 0 && K <= ROAM_MAX_FEATURES && mask_out && n_inliers && (K == 0 || (prev && next)));
    if (K == 0) { *n_inliers = 0; if (flags_out) *flags_out = 1; return ROAM_OK; }
    const int nw = (K + 63) / 64;
    SCRATCH(dp, float, S_IN0, sizeof(float) * 2 * (size_t)K);
    SCRATCH(dn, float, S_IN1, sizeof(float) * 2 * (size_t)K);
    SCRATCH(dadj, uint64_t, S_TMP0, sizeof(uint64_t) * (size_t)K * nw);
    SCRATCH(dstk, uint64_t, S_TMP1, sizeof(uint64_t) * (size_t)(K + 2) * 2 * nw);
    SCRATCH(dmask, uint8_t, S_OUT0, (size_t)K);
    SCRATCH(dres, int32_t, S_OUT1, sizeof(int32_t) * 2);
    H2D(dp, prev, sizeof(float) * 2 * (size_t)K);
    H2D(dn, next, sizeof(float) * 2 * (size_t)K);
    HIP_TRY(ctx, launch_consistency_graph(ctx->stream, dp, dn, nullptr, K, K, 1, thr_px, dadj, nw));
    HIP_TRY(ctx, launch_max_clique(ctx->stream, dadj, nullptr, K, K, nw, 1, node_limit, dstk, dmask, dres, dres + 1));
    int32_t res[2];
    D2H(mask_out, dmask, (size_t)K);
    D2H(res, dres, sizeof(res));
    if (adj_out) D2H(adj_out, dadj, sizeof(uint64_t) * (size_t)K * nw);
    SYNC();
    *n_inliers = res[0];
    if (flags_out) *flags_out = res[1];
    return ROAM_OK;
}

// throughput of a8 on one correspondence set replicated `copies` times (each copy its own problem, as in an engine step):
// average milliseconds per launch of the consistency-graph and the maximum-clique kernel over `reps` launches
extern "C" int32_t roam_time_reject_outliers(roam_ctx *ctx, const float *prev, const float *next, int32_t K, int32_t copies,
                                             double thr_px, int64_t node_limit, int32_t reps, float *graph_ms, float *clique_ms,
                                             int32_t *n_inliers, int32_t *proven)
{
    ENTER();
    ARG_CHECK(ctx, K >= 1 && K <= ROAM_MAX_FEATURES && prev && next && copies >= 1 && copies <= 65536 && reps >= 1 && graph_ms && clique_ms);
    const int nw = (K + 63) / 64;
    SCRATCH(dp, float, S_IN0, sizeof(float) * 2 * (size_t)K * copies);
    SCRATCH(dn, float, S_IN1, sizeof(float) * 2 * (size_t)K * copies);
    SCRATCH(dadj, uint64_t, S_TMP0, sizeof(uint64_t) * (size_t)K * nw * copies);
    SCRATCH(dstk, uint64_t, S_TMP1, sizeof(uint64_t) * (size_t)(K + 2) * 2 * nw * copies);
    SCRATCH(dmask, uint8_t, S_OUT0, (size_t)K * copies);
    SCRATCH(dres, int32_t, S_OUT1, sizeof(int32_t) * 2 * (size_t)copies);
    for (int c = 0; c < copies; c++) {
        H2D(dp + 2 * (size_t)K * c, prev, sizeof(float) * 2 * (size_t)K);
        H2D(dn + 2 * (size_t)K * c, next, sizeof(float) * 2 * (size_t)K);
    }
    hipEvent_t e0, e1, e2;
    HIP_TRY(ctx, hipEventCreate(&e0)); HIP_TRY(ctx, hipEventCreate(&e1)); HIP_TRY(ctx, hipEventCreate(&e2));
    double g = 0, q = 0;
    for (int r = 0; r < reps; r++) {
        HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        HIP_TRY(ctx, launch_consistency_graph(ctx->stream, dp, dn, nullptr, K, K, copies, thr_px, dadj, nw));
        HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        HIP_TRY(ctx, launch_max_clique(ctx->stream, dadj, nullptr, K, K, nw, copies, node_limit, dstk, dmask, dres, dres + copies));
        HIP_TRY(ctx, hipEventRecord(e2, ctx->stream));
        SYNC();
        float a = 0, b = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&a, e0, e1)); HIP_TRY(ctx, hipEventElapsedTime(&b, e1, e2));
        g += a; q += b;
    }
    hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2);
    *graph_ms = (float)(g / reps); *clique_ms = (float)(q / reps);
    int32_t res[2];
    D2H(&res[0], dres + (copies - 1), sizeof(int32_t));
    D2H(&res[1], dres + copies + (copies - 1), sizeof(int32_t));
    SYNC();
    if (n_inliers) *n_inliers = res[0];
    if (proven) *proven = res[1];
    return ROAM_OK;
}

extern "C" int32_t roam_kabsch2d(roam_ctx *ctx, const double *src, const double *tgt, int32_t N,
                                 double *R, double *h)
{
    ENTER();
    ARG_CHECK(ctx, src && tgt && R && h && N >= 1);
    SCRATCH(ds, double, S_IN0, sizeof(double) * 2 * (size_t)N);
    SCRATCH(dt, double, S_IN1, sizeof(double) * 2 * (size_t)N);
    SCRATCH(dout, double, S_OUT0, sizeof(double) * 6);
    H2D(ds, src, sizeof(double) * 2 * (size_t)N);
    H2D(dt, tgt, sizeof(double) * 2 * (size_t)N);
    HIP_TRY(ctx, launch_kabsch(ctx->stream, ds, dt, nullptr, N, N, 1, dout));
    double o[6];
    D2H(o, dout, sizeof(o));
    SYNC();
    R[0] = o[0]; R[1] = o[1]; R[2] = o[2]; R[3] = o[3]; h[0] = o[4]; h[1] = o[5];
    return ROAM_OK;
}

extern "C" int32_t roam_mds_solve(roam_ctx *ctx, const double *T_wj0, const double *p_w, const double *p_jt,
                                  int32_t N, const double *T_wj_init, const double *sigma5, double period,
                                  double *out6, int32_t *nfev, int32_t *info, double *x0_out, double *r0_out)
{
    ENTER();
    ARG_CHECK(ctx, T_wj0 && p_w && p_jt && T_wj_init && sigma5 && out6 && N >= 2 && N <= 8 * ROAM_MAX_FEATURES && period > 0);
    const size_t m = 2 * (size_t)N + 3;
    SCRATCH(dT, double, S_IN0, sizeof(double) * 18);
    SCRATCH(dpw, double, S_IN1, sizeof(double) * 2 * (size_t)N);
    SCRATCH(dpj, double, S_IN2, sizeof(double) * 2 * (size_t)N);
    SCRATCH(dwork, double, S_TMP0, sizeof(double) * (m * 9 + N));
    SCRATCH(dout, double, S_OUT0, sizeof(double) * (6 + 6));
    SCRATCH(dr0, double, S_OUT1, sizeof(double) * m);
    SCRATCH(dint, int32_t, S_OUT2, sizeof(int32_t) * 2);
    double Tb[18];
    memcpy(Tb, T_wj0, sizeof(double) * 9);
    memcpy(Tb + 9, T_wj_init, sizeof(double) * 9);
    H2D(dT, Tb, sizeof(Tb));
    H2D(dpw, p_w, sizeof(double) * 2 * (size_t)N);
    H2D(dpj, p_jt, sizeof(double) * 2 * (size_t)N);
    MdsProblemDesc P;
    P.T_wj0 = dT; P.T_init = dT + 9; P.p_w = dpw; P.p_jt = dpj; P.count = nullptr;
    P.N = N; P.nstride = N; P.nmax = N; P.B = 1; P.period = period;
    for (int i = 0; i < 5; i++) P.sigma5[i] = sigma5[i];
    HIP_TRY(ctx, launch_mds_solve(ctx->stream, P, dwork, dout, dint, dint + 1, dout + 6, dr0));
    double o[12];
    int32_t ii[2];
    D2H(o, dout, sizeof(o));
    D2H(ii, dint, sizeof(ii));
    if (r0_out) D2H(r0_out, dr0, sizeof(double) * m);
    SYNC();
    memcpy(out6, o, sizeof(double) * 6);
    if (x0_out) memcpy(x0_out, o + 6, sizeof(double) * 6);
    if (nfev) *nfev = ii[0];
    if (info) *info = ii[1];
    return ROAM_OK;
}

extern "C" int32_t roam_mds_undistort(roam_ctx *ctx, const double *v3, const double *pts, int32_t N,
                                      double period, double *out_xy, double *dT_out)
{
    ENTER();
    ARG_CHECK(ctx, v3 && pts && N >= 0 && (out_xy || dT_out));
    if (N == 0) return ROAM_OK;
    SCRATCH(dv, double, S_IN0, sizeof(double) * 3);
    SCRATCH(dp, double, S_IN1, sizeof(double) * 2 * (size_t)N);
    SCRATCH(dxy, double, S_OUT0, sizeof(double) * 2 * (size_t)N);
    SCRATCH(ddt, double, S_OUT1, sizeof(double) * (size_t)N);
    H2D(dv, v3, sizeof(double) * 3);
    H2D(dp, pts, sizeof(double) * 2 * (size_t)N);
    HIP_TRY(ctx, launch_mds_undistort(ctx->stream, dv, dp, N, period, dxy, ddt));
    if (out_xy) D2H(out_xy, dxy, sizeof(double) * 2 * (size_t)N);
    if (dT_out) D2H(dT_out, ddt, sizeof(double) * (size_t)N);
    SYNC();
    return ROAM_OK;
}
