// Internal declarations shared by the .hip translation units of libroam_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../include/roam_abi.h"

// HIP's __fsqrt_rn maps to the APPROXIMATE v_sqrt_f32 (ocml native sqrt) unless
// OCML_BASIC_ROUNDED_OPERATIONS is defined; __builtin_sqrtf lowers to the correctly
// rounded expansion (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).  The other
// __f*_rn intrinsics are plain operators, so every translation unit is compiled with
// -ffp-contract=off (and says so with the pragma below) to keep a*b+c un-fused.
#pragma clang fp contract(off)
__device__ __forceinline__ float rn_sqrtf(float x) { return __builtin_sqrtf(x); }

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct Engine;
struct Comm;

struct roam_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;     // stage A stream: peaks + warp (depend only on the raw scan)
    hipStream_t stream4 = nullptr;     // stage B stream: pyramid of the warped image
    hipStream_t stream5 = nullptr;     // polar peaks (independent of everything but the raw scan)
    hipStream_t stream3 = nullptr;     // copy stream: asynchronous record uploads from pinned host memory
    hipEvent_t ev_up = nullptr, ev_fence = nullptr;
    char err[512] = {0};
    // growable scratch buffers for the stage API (indexed by role)
    DevBuf scratch[24];
    int cu_count = 0;
    Engine *engine = nullptr;
    Comm *comm = nullptr;              // RCCL communicator (comm.hip), optional
};

#define ROAM_SET_ERR(ctx, ...) snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__)

#define HIP_TRY(ctx, call)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            ROAM_SET_ERR(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return ROAM_E_HIP;                                                              \
        }                                                                                   \
    } while (0)

#define ARG_CHECK(ctx, cond)                                                                \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            ROAM_SET_ERR(ctx, "bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__);       \
            return ROAM_E_ARG;                                                              \
        }                                                                                   \
    } while (0)

// grow-only scratch allocation; returns nullptr on failure (error text set)
void *roam_scratch(roam_ctx *ctx, int slot, size_t bytes);

enum ScratchSlot {
    S_IN0 = 0, S_IN1, S_IN2, S_IN3, S_OUT0, S_OUT1, S_OUT2, S_OUT3,
    S_TMP0, S_TMP1, S_TMP2, S_TMP3, S_PYR_A, S_PYR_B, S_TMP4, S_TMP5, S_TMP6, S_TMP7
};

// ---------------------------------------------------------------- kernel launchers
// (all asynchronous on `st`; device pointers; B = number of lanes/problems in the batch)

struct PeakSrc {
    const void *base;      // f32 rows or u8 record rows
    int64_t lane_stride;   // elements between lanes (floats or bytes)
    int64_t row_stride;    // elements between rows
    int32_t payload_off;   // u8 only
    int32_t is_u8;
    const int32_t *lane_index;  // optional indirection: lane b reads base + lane_index[b]*lane_stride
};
// row_stage: B x rows x stage_cap u16; row_count: B x rows i32; out: B x cap x 2 i32; n_out: B i32
hipError_t launch_peaks(hipStream_t st, PeakSrc src, int B, int rows, int cols, uint16_t *row_stage,
                        int stage_cap, int32_t *row_count, int32_t *out, int32_t cap, int32_t *n_out);

struct WarpSrc {
    const void *base;
    int64_t lane_stride, row_stride;
    int32_t payload_off, is_u8;
    const int32_t *lane_index;
};
// cart_u8 / cart_f32: B x W x W (either may be null); W = 2*(cols/2)
hipError_t launch_polar_to_cart(hipStream_t st, WarpSrc src, int B, int rows, int cols,
                                uint8_t *cart_u8, int64_t u8_lane_stride, float *cart_f32,
                                int64_t f32_lane_stride);
// engine path: precomputed sampling map (W x W u32) + gather
hipError_t launch_warp_map(hipStream_t st, int rows, int cols, uint32_t *map);
// dark_stays_zero: the tiles beyond the maximum range (a fifth of the image, zero whatever the scan holds) are not written - for
// destinations that were zero-filled once and are written by this kernel only (the engine's pyramids)
hipError_t launch_warp_gather(hipStream_t st, const uint32_t *map, WarpSrc src, int B, int rows, int cols,
                              uint8_t *cart_u8, int64_t u8_lane_stride, bool dark_stays_zero = false);
hipError_t launch_quantize_u8(hipStream_t st, const float *img, int64_t n, uint8_t *out);

// pyramid storage: level l of lane b at base + b*lane_stride + level_off[l]
struct PyrDesc {
    int32_t w[ROAM_PYR_LEVELS], h[ROAM_PYR_LEVELS];
    int64_t off[ROAM_PYR_LEVELS];
    int64_t lane_stride;
};
void pyr_desc_init(PyrDesc *d, int w, int h);
hipError_t launch_pyr_down(hipStream_t st, const uint8_t *src, int64_t src_lane_stride, int w, int h,
                           uint8_t *dst, int64_t dst_lane_stride, int B, const unsigned long long *dark = nullptr);
// dark_l0 (optional, launch_pyr_dark): the lanes of the 2024 -> 1012 kernel whose pixels lie beyond the maximum range neither load nor
// store - for pyramids that were zero-filled once and whose level 0 comes from launch_warp_gather(..., dark_stays_zero)
hipError_t launch_build_pyramid(hipStream_t st, uint8_t *pyr, const PyrDesc &d, int B, const unsigned long long *dark_l0 = nullptr);
size_t pyr_dark_words(int h);
hipError_t launch_pyr_dark(hipStream_t st, const uint32_t *map, int w, int h, int cols, unsigned long long *dark);

// KLT: pts/next: B x kstride x 2 f32; count[b] features per lane (or all K if count==null)
hipError_t launch_klt(hipStream_t st, const uint8_t *prev_pyr, const uint8_t *next_pyr,
                      const PyrDesc &d, const float *pts, const int32_t *count, int K, int kstride,
                      int B, float *next, uint8_t *status, float *err);

// consistency graph: adj: B x K_stride rows x nw words
hipError_t launch_consistency_graph(hipStream_t st, const float *prev, const float *next,
                                    const int32_t *count, int K, int kstride, int B, double thr,
                                    uint64_t *adj, int nw);
// max clique (lexicographically smallest maximum clique); stack scratch: B x (kstride+2) x 2 x nw words
hipError_t launch_max_clique(hipStream_t st, const uint64_t *adj, const int32_t *count, int K,
                             int kstride, int nw, int B, int64_t node_limit, uint64_t *stack,
                             uint8_t *mask, int32_t *n_in, int32_t *flags, int32_t *order = nullptr);      // order: B ints of scratch - the
                                                                                                           // problems are then started largest first

// order[0..B) = the indices 0..B-1 by falling min(count, cmax) >> shift (at most 1024 distinct keys): one workgroup, a counting sort
hipError_t launch_order_by_count(hipStream_t st, const int32_t *count, int B, int cmax, int32_t *order, int shift);

// Kabsch on f64 pairs: src/tgt B x nstride x 2; out: B x 6 doubles [R00 R01 R10 R11 hx hy]
hipError_t launch_kabsch(hipStream_t st, const double *src, const double *tgt, const int32_t *count,
                         int N, int nstride, int B, double *out6);

struct MdsProblemDesc {
    const double *T_wj0;     // B x 9
    const double *p_w;       // B x nstride x 2
    const double *p_jt;      // B x nstride x 2
    const double *T_init;    // B x 9
    const int32_t *count;    // B (or null -> N)
    int N, nstride, B;
    int nmax;                // upper bound of count[] (sizes the LM working set); <= nstride
    double sigma5[5];
    double period;
    // optional (engine): the problems too large for the one-wavefront form, listed by that kernel for the workgroup form:
    // two slots of 1 + B ints (count, problem ids), zero on first use; big_slot alternates between consecutive solves of a stream
    int32_t *big = nullptr;
    int big_slot = 0;
    // optional (engine, batches): the workgroup form runs on `side` BESIDE the wave form instead of behind it (both forms skip each other's
    // problems; no list then): fork / join events of the caller
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};
// work: B x ((2*nmax+3) x 9 + nmax) doubles; out6: B x 6; nfev/info: B; x0/r0 optional
hipError_t launch_mds_solve(hipStream_t st, const MdsProblemDesc &p, double *work, double *out6,
                            int32_t *nfev, int32_t *info, double *x0_out, double *r0_out);
hipError_t launch_mds_undistort(hipStream_t st, const double *v3, const double *pts, int N,
                                double period, double *out_xy, double *dT);

hipError_t launch_ssc(hipStream_t st, const double *kp, int B, int num_ret, double tol, int cols,
                      int rows, int32_t *work, int32_t *sel, int32_t *n_sel);

int32_t roam_doh_maxima_record_device(roam_ctx *ctx, const uint8_t *rec, int rows, int64_t stride, int payload_off,
                                      int clip, const double *sigmas, int32_t num_sigma, double threshold,
                                      int32_t *out_rcs, double *out_val, int32_t cap, int32_t *n_out);

// comm.hip: in-place byte broadcast of a device buffer on ctx->stream (asynchronous), this rank's index
int32_t roam_comm_bcast_bytes(roam_ctx *ctx, void *dev_buf, size_t bytes, int root);
int roam_comm_rank(const roam_ctx *ctx);
int roam_comm_world(const roam_ctx *ctx);
int32_t roam_comm_allgather_bytes(roam_ctx *ctx, const void *send, void *recv, size_t bytes, hipStream_t st);
