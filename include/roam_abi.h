/* roam_abi.h — C ABI of libroam_hip.so, the MI355X-native (gfx950, hand-written HIP)
 * replacement for the per-scan hot path of Samleo8/RadarSLAMPy ("RAW-ROAM").
 *
 * The reference has no FFI: its "plugin API" is a set of Python call signatures.  Each
 * entry point below states the reference callable it replaces (file:line, relative to the
 * reference repo).  radarslampy_amd/_ffi.py binds exactly these symbols with ctypes;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  Every function returns int32 status:
 *     ROAM_OK (0) or a negative ROAM_E_* code; roam_last_error(ctx) has the text.
 *   - The caller owns every host buffer.  Outputs go into caller-allocated, capacity-
 *     checked arrays.  Device memory is owned by the context.
 *   - Arrays are row-major contiguous.  Coordinates are [x, y] float32 pixels; poses are
 *     [x, y, theta] float64; 3x3 transforms are row-major float64[9].
 *   - One context = one GPU + one HIP stream; a context is single-threaded, distinct
 *     contexts are independent.  No global mutable state.
 *   - "stage" entry points take host arrays (H2D, kernel(s), D2H) and mirror one reference
 *     function each.  "engine" entry points keep B independent sequences ("lanes")
 *     resident in HBM and advance all of them by one scan pair per call without touching
 *     the host (the throughput path measured by bench.py).
 */
#ifndef ROAM_ABI_H
#define ROAM_ABI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct roam_ctx roam_ctx;

enum {
    ROAM_OK = 0,
    ROAM_E_ARG = -1,        /* bad argument (null pointer, size out of range)       */
    ROAM_E_HIP = -2,        /* a HIP runtime call failed                              */
    ROAM_E_CAPACITY = -3,   /* caller-provided output capacity too small             */
    ROAM_E_NODEVICE = -4,   /* no usable gfx950 device                                */
    ROAM_E_STATE = -5       /* engine call in the wrong state                         */
};

/* limits of this build */
#define ROAM_MAX_FEATURES 1024      /* K_max per lane (KLT / graph / LM)                */
#define ROAM_MAX_COLS 4096          /* longest polar row the peak kernel accepts        */
#define ROAM_PYR_LEVELS 4           /* LK maxLevel=3 (getTransformKLT.py:77-81)         */

/* ---- context ------------------------------------------------------------------------ */
int32_t roam_create(int32_t device_id, roam_ctx **out);
int32_t roam_destroy(roam_ctx *ctx);
const char *roam_last_error(const roam_ctx *ctx);
const char *roam_version(void);
/* name_cap bytes for the device name; any out pointer may be NULL */
int32_t roam_device_info(roam_ctx *ctx, char *name, int32_t name_cap, int32_t *cu_count,
                         int64_t *hbm_bytes, char *arch, int32_t arch_cap);
int32_t roam_synchronize(roam_ctx *ctx);
/* pinned (page-locked) host memory for asynchronous record uploads */
int32_t roam_host_alloc(roam_ctx *ctx, int64_t bytes, void **out);
int32_t roam_host_free(roam_ctx *ctx, void *p);

/* ---- f2: record ingest from PNG files (parseData.py:160-226: cv2.imread(path, IMREAD_GRAYSCALE) at :178; the reference decodes
 * frame k inside its loop, RawROAMSystem.py:162-165).  Host code only - no device, no context.  The one format of the data set:
 * 8-bit greyscale, non-interlaced; anything else (and a corrupt file) is ROAM_E_ARG.  The image is written row by row into out
 * (out_stride bytes between rows, 0 = the image width; ROAM_E_CAPACITY when (rows - 1) * out_stride + cols > out_bytes) - typically
 * a slot of a roam_host_alloc ring, so that a decoded record is uploaded from where it was decoded.  rows / cols may be NULL. */
int32_t roam_png_decode_gray8(const uint8_t *png, int64_t png_bytes, uint8_t *out, int64_t out_bytes, int64_t out_stride,
                              int32_t *rows, int32_t *cols);
int32_t roam_png_decode_file(const char *path, uint8_t *out, int64_t out_bytes, int64_t out_stride, int32_t *rows, int32_t *cols);
/* a pool of host threads decoding files ahead of the consumer.  submit() queues one file for one destination and returns at once; the
 * caller names the job with a ticket of its choice (unique among the jobs in flight) and wait(ticket) blocks until THAT job is done and
 * returns its status (tickets complete in any order; every submitted ticket must be waited for before its destination is reused).
 * destroy() drops the jobs that have not started and joins the threads. */
typedef struct roam_png_pool roam_png_pool;
int32_t roam_png_pool_create(int32_t workers, roam_png_pool **out);
int32_t roam_png_pool_submit(roam_png_pool *pool, const char *path, uint8_t *dst, int64_t dst_bytes, int64_t dst_stride, int64_t ticket);
int32_t roam_png_pool_wait(roam_png_pool *pool, int64_t ticket, int32_t *rows, int32_t *cols);
int32_t roam_png_pool_destroy(roam_png_pool *pool);

/* ---- a2: getPointCloud.getPointCloudPolarInd (getPointCloud.py:11-54) ------------------
 * polar: rows x cols float32.  out: (cap,2) int32 rows [azimuthIdx, rangeIdx], azimuth-
 * major, range ascending.  *n_out = number of peaks found; if it exceeds cap the call
 * returns ROAM_E_CAPACITY (the first cap pairs are valid). */
int32_t roam_peaks_polar_f32(roam_ctx *ctx, const float *polar, int32_t rows, int32_t cols,
                             int32_t *out, int64_t cap, int64_t *n_out);
/* a1+a2 fused: raw Oxford record rows (parseData.extractDataFromRadarImage, parseData.py:17-53):
 * value = rec[r*stride + payload_off + i] / 255 (float32), i < clip. */
int32_t roam_peaks_record_u8(roam_ctx *ctx, const uint8_t *rec, int32_t rows, int64_t stride,
                             int32_t payload_off, int32_t clip, int32_t *out, int64_t cap,
                             int64_t *n_out);

/* ---- a3: parseData.convertPolarImageToCartesian (parseData.py:100-135) -------------------
 * polar rows x cols f32 -> (2R x 2R), R = cols/2.  cart_f32 and/or cart_u8 may be NULL;
 * cart_u8 is the (img*255).astype(uint8) of getTransformKLT.py:356-357. */
int32_t roam_polar_to_cart_f32(roam_ctx *ctx, const float *polar, int32_t rows, int32_t cols,
                               float *cart_f32, uint8_t *cart_u8);
int32_t roam_polar_to_cart_record_u8(roam_ctx *ctx, const uint8_t *rec, int32_t rows, int64_t stride,
                                     int32_t payload_off, int32_t clip, float *cart_f32,
                                     uint8_t *cart_u8);

/* ---- a7: cv2.calcOpticalFlowPyrLK as used by getTransformKLT.getTrackedPointsKLT
 * (getTransformKLT.py:317-381; LK_PARAMS :77-81: winSize 15, maxLevel 3, 10 iter, eps 0.03).
 * Images are w x h; *_f32 variants quantise (img*255 -> u8, :356-357) on the device.
 * pts (K,2) f32 -> next_pts (K,2) f32, status (K) u8, err (K) f32.  The reference's
 * `status &= err < ERR_THRESHOLD` (:365) is applied by the caller. */
int32_t roam_klt_track_u8(roam_ctx *ctx, const uint8_t *prev_img, const uint8_t *next_img,
                          int32_t w, int32_t h, const float *pts, int32_t K,
                          float *next_pts, uint8_t *status, float *err);
int32_t roam_klt_track_f32(roam_ctx *ctx, const float *prev_img, const float *next_img,
                           int32_t w, int32_t h, const float *pts, int32_t K,
                           float *next_pts, uint8_t *status, float *err);
/* pyrDown (5x5 Gaussian, REFLECT_101): src w x h -> dst ((w+1)/2 x (h+1)/2) */
int32_t roam_pyr_down_u8(roam_ctx *ctx, const uint8_t *src, int32_t w, int32_t h, uint8_t *dst);

/* ---- a8: outlierRejection.rejectOutliers (outlierRejection.py:16-95) --------------------
 * prev/next (K,2) f32.  mask_out (K) u8 = membership of the maximum clique of the
 * |d_prev - d_next| <= thr_px consistency graph; when several maximum cliques exist, the one
 * the reference returns: the first strictly-largest clique in networkx.find_cliques order
 * (outlierRejection.py:63-75; CPython set order restated on the device, csrc/clique.hip).
 * node_limit bounds the branch-and-bound (0 = default); *flags_out bit0 = search completed
 * (result proven maximum and equal to the reference's; otherwise the best clique found). adj_out (optional) receives
 * the K x ((K+63)/64) uint64 adjacency bit rows. */
int32_t roam_reject_outliers(roam_ctx *ctx, const float *prev, const float *next, int32_t K,
                             double thr_px, int64_t node_limit, uint8_t *mask_out,
                             int32_t *n_inliers, int32_t *flags_out, uint64_t *adj_out);

/* measurement aid: the same correspondence set replicated `copies` times (one problem each, as in an engine step);
 * average ms per launch of the consistency-graph and the maximum-clique kernel over `reps` launches */
int32_t roam_time_reject_outliers(roam_ctx *ctx, const float *prev, const float *next, int32_t K, int32_t copies,
                                  double thr_px, int64_t node_limit, int32_t reps, float *graph_ms, float *clique_ms,
                                  int32_t *n_inliers, int32_t *proven);

/* ---- a10: getTransformKLT.calculateTransformSVD (getTransformKLT.py:129-162) -------------
 * src ~= R tgt + h over N pairs of float64 [x,y]; R row-major [4], h [2]. */
int32_t roam_kabsch2d(roam_ctx *ctx, const double *src, const double *tgt, int32_t N,
                      double *R, double *h);

/* ---- a11-a14: motionDistortion.MotionDistortionSolver (motionDistortion.py:70-205,295-325)
 * update_problem + optimize_library in one call.  sigma5 = [sp_x, sp_y, sv_x, sv_y, sv_th]
 * (covariance diagonals; residual weights are 1/sigma as in :96-99).  out6 = [v(3), pose(3)].
 * x0_out (6) and r0_out (2N+3) are optional: start vector and error_vector(x0). */
int32_t roam_mds_solve(roam_ctx *ctx, const double *T_wj0, const double *p_w, const double *p_jt,
                       int32_t N, const double *T_wj_init, const double *sigma5, double period,
                       double *out6, int32_t *nfev, int32_t *info, double *x0_out, double *r0_out);
/* MotionDistortionSolver.undistort (:126-153) / compute_time_deltas (:107-124): pts (N,2) f64 */
int32_t roam_mds_undistort(roam_ctx *ctx, const double *v3, const double *pts, int32_t N,
                           double period, double *out_xy, double *dT_out);

/* ---- a5: ANMS.ssc (ANMS.py:5-102) ----------------------------------------------------------
 * kp (B,3) f64 rows [row, col, sigma] in priority order; sel_out (cap >= B) receives the
 * selected indices in input order. */
int32_t roam_ssc(roam_ctx *ctx, const double *kp, int32_t B, int32_t num_ret, double tol,
                 int32_t cols, int32_t rows, int32_t *sel_out, int32_t *n_sel);

/* ---- a4: getFeatures.getBlobsFromCart (skimage blob_doh, getFeatures.py:22-53) ------------
 * image-scale part of blob_doh: float64 integral image, box-filter Hessian determinant for
 * every sigma, 3x3x3 local maxima above `threshold`.  img w x h f32.  out_rcs (cap,3) int32 rows
 * [row, col, sigma_index] in C (row, col, sigma) order, out_val (cap) the determinant values.
 * Ordering by response and overlap pruning (_prune_blobs) are host bookkeeping. */
int32_t roam_doh_maxima(roam_ctx *ctx, const float *img, int32_t w, int32_t h, const double *sigmas,
                        int32_t num_sigma, double threshold, int32_t *out_rcs, double *out_val,
                        int32_t cap, int32_t *n_out);

/* ---- a4/a5 bookkeeping whose ORDER the reference's pinned third-party stack fixes (host code, no GPU; the engine's
 * device-side retrack runs the same functions on the GPU):
 * roam_prune_blobs   = skimage.feature.blob._prune_blobs as blob_doh calls it (getFeatures.py:47-51): candidate pairs in the
 *                      iteration order of the Python set that scipy cKDTree.query_pairs fills.  blobs (n,3) f64 rows
 *                      [row, col, sigma] in peak_local_max order, integer rows / cols; keep_out (n) u8.
 * roam_argsort_np122 = np.argsort of the pinned NumPy 1.22.3 (unstable introsort) that adaptiveNMS applies to the
 *                      two-valued sigmas (getFeatures.py:69); order_out (n) i32. */
int32_t roam_prune_blobs(const double *blobs, int32_t n, double overlap, uint8_t *keep_out);
int32_t roam_argsort_np122(const double *keys, int32_t n, int32_t *order_out);

/* ---- f4: FMT.getRotationUsingFMT (FMT.py:36-90; called first by Tracker.track, Tracker.py:62-63) ---------------------
 * Fourier-Mellin rotation prior between two polar images (rows x cols float32): range clip (clip_px bins, <= 0: none),
 * cv2.resize to clip / downsample columns, polar -> Cartesian -> log-polar, Hanning-windowed phase correlation.
 * angle_rad: R(angle) src = target; scale and response are optional. */
int32_t roam_fmt_rotation(roam_ctx *ctx, const float *src_polar, const float *tgt_polar, int32_t rows, int32_t cols,
                          int32_t clip_px, int32_t downsample, double *angle_rad, double *scale, double *response);

/* ---- engine: B resident lanes, one scan pair per lane per step ---------------------------
 * Replaces the body of the RawROAMSystem.run loop (RawROAMSystem.py:162-298) minus plotting:
 * a1/a2 ingest+peaks, a3 warp+quantise, pyramid, a7 KLT against the lane's previous
 * pyramid, status &= err<10, a8 outlier rejection, a10 Kabsch, a15 glue, a11-a14 LM,
 * keyframe bookkeeping (Mapping.py:37-66,118-125,149-174). */
typedef struct roam_engine_cfg {
    int32_t lanes;            /* B                                                       */
    int32_t rows;             /* 400                                                      */
    int32_t stride;           /* 3779 bytes per record row                                */
    int32_t payload_off;      /* 11                                                       */
    int32_t clip;             /* 2025                                                     */
    int32_t pool_scans;       /* number of raw records kept resident in HBM               */
    int32_t peaks_cap;        /* per-lane capacity of the peak list                       */
    int32_t reject_outliers;  /* paramFlags["rejectOutliers"] (Tracker.py:93)             */
    int32_t motion_distortion;/* 1: LM pose (RawROAMSystem.py:208-237); 0: Kabsch dead reckoning (:236,301-317) */
    int64_t clique_node_limit;
    double sigma5[5];
    int32_t retrack_on_device;/* 1: lanes that run out of features (<= 60 inliers, RawROAMSystem.py:250-271) re-detect
                                 (appendNewFeatures: DoH blobs + ANMS, getFeatures.py:74-118) inside roam_engine_step */
    int32_t retrack_slots;    /* lanes whose detection scratch (33 MB each) is resident at once; 0 = min(lanes, 2048) */
    /* Map.isGoodKeyframe (Mapping.py:149-174): a keyframe is added when the pose moved this far from the last one.  <= 0: the
     * reference's constants, TRANS_THRESHOLD = 2.0 m and ROT_THRESHOLD = 0.2 rad (Mapping.py:13-15).  (The pictures the reference keeps
     * of its data/tiny run were made with a keyframe on EVERY frame - DESIGN.md section 4; 1e-9 reproduces that.) */
    double keyframe_trans_m;
    double keyframe_rot_rad;
} roam_engine_cfg;

typedef struct roam_lane_result {
    double pose[3];           /* latest pose [x,y,th]                                    */
    double velocity[3];
    double kabsch_R[4];
    double kabsch_h[2];       /* metres                                                   */
    int32_t n_tracked;        /* K fed to KLT                                             */
    int32_t n_good;           /* after status & err<10                                    */
    int32_t n_inliers;        /* after outlier rejection                                  */
    int32_t n_peaks;          /* polar peaks of the current scan                          */
    int32_t lm_nfev;
    int32_t lm_info;
    int32_t flags;            /* bit0 clique proven, bit1 keyframe added, bit2 retrack wanted (features ran out),
                                 bit3 retrack done on the device in this step; bits 8..11 detection overflows
                                 (candidates > 2048, k-d tree, pairs > 32767, features > 1024): never set on real scans */
    int32_t n_after_retrack;  /* feature count after the device-side append (bit3), else 0 */
} roam_lane_result;

int32_t roam_engine_create(roam_ctx *ctx, const roam_engine_cfg *cfg);
int32_t roam_engine_destroy(roam_ctx *ctx);
/* copy one raw record (rows x stride u8) into pool slot idx */
int32_t roam_engine_upload_scan(roam_ctx *ctx, int32_t pool_idx, const uint8_t *rec);
/* raw-record ingest (reference parseData.py:160-226 loads one PNG per frame; here n records of rows x stride u8,
 * `host_stride` bytes apart in PINNED host memory, are copied to pool slots pool_idx0.. (only the payload_off + clip
 * bytes of every row that the path reads cross PCIe) on a copy stream that
 * overlaps the compute stream).  roam_engine_step waits for the uploads that wrote the pool slots IT reads (the newest of them; not for
 * uploads of other slots enqueued meanwhile - those may themselves be waiting, behind a fence, for earlier steps);
 * roam_engine_fence makes later uploads wait for the steps enqueued so far (double-buffered pools). */
/* host_records must be pinned / registered host memory (the copy kernel reads it from the GPU): a pageable pointer is refused
 * with ROAM_E_ARG */
int32_t roam_engine_upload_scans_async(roam_ctx *ctx, int32_t pool_idx0, int32_t n, const uint8_t *host_records, int64_t host_stride);
int32_t roam_engine_fence(roam_ctx *ctx);
/* device-to-device copy of a resident record (lets a benchmark give every lane its own copy of a
 * scan so that input reads are real HBM traffic rather than L2 / Infinity-Cache hits) */
int32_t roam_engine_copy_scan(roam_ctx *ctx, int32_t dst_idx, int32_t src_idx);
/* initialise a lane from a pool scan: pyramid of that scan becomes "previous", features =
 * pts (K,2) f32 pixel [x,y], pose = pose3.  (First-frame feature detection is a4/a5.) */
int32_t roam_engine_init_lane(roam_ctx *ctx, int32_t lane, int32_t pool_idx, const float *pts,
                              int32_t K, const double *pose3);
/* advance every lane by one scan pair: lane i consumes pool scan scan_idx[i] as its current
 * scan.  Asynchronous: the call only enqueues work (scan_idx is copied before it returns).  Steps
 * enqueued back to back are pipelined on the device - peaks / warp of step N+2, the pyramid of step
 * N+1 and the tracking + pose solve of step N run concurrently - with results identical to
 * synchronised execution; roam_engine_results and the other blocking accessors return the state
 * after the LAST enqueued step.  The host may run at most three steps ahead of the device. */
int32_t roam_engine_step(roam_ctx *ctx, const int32_t *scan_idx);
/* scan_idx[i] | ROAM_STEP_NEW_SEQUENCE: lane i starts a NEW sequence on this scan - its features are dropped before the pair,
 * nothing is tracked, the pose stays, and the first-frame detection (appendNewFeatures(prevImgCart, empty),
 * RawROAMSystem.py:150) runs on this scan inside the step (needs cfg.retrack_on_device).  A stream of finite sequences
 * per lane without any host synchronisation between them. */
#define ROAM_STEP_NEW_SEQUENCE 0x40000000
/* blocking: fetch the per-lane results of the last step */
int32_t roam_engine_results(roam_ctx *ctx, roam_lane_result *out, int32_t n);
/* per-step results without draining the pipeline: every step's records are copied to pinned host memory right behind the step
 * on the compute stream (ring of the last 8 steps); this call waits for step `step` only (0-based count of roam_engine_step calls) - poses and
 * flags of step N can be consumed while steps N+1.. are still running.  ROAM_E_STATE if the step left the ring. */
int32_t roam_engine_step_results(roam_ctx *ctx, int64_t step, roam_lane_result *out, int32_t n);
int32_t roam_engine_steps_enqueued(roam_ctx *ctx, int64_t *nstep);
/* device-side retrack of the following steps: 0 = suspended (flags are still raised), 1 = lanes that ran out of features
 * (default), 2 = every lane in every step (measurement of the detection cost) */
int32_t roam_engine_set_retrack(roam_ctx *ctx, int32_t mode);
/* like roam_engine_init_lane, but the initial features are DETECTED on the device from the pool scan
 * (appendNewFeatures(prevImgCart, empty), RawROAMSystem.py:150); needs cfg.retrack_on_device */
int32_t roam_engine_init_lane_detect(roam_ctx *ctx, int32_t lane, int32_t pool_idx, const double *pose3);
/* the same for the n lanes lane0 .. lane0 + n - 1 in one pass (pool_idx[n], poses3[n][3]): one warp / pyramid launch and one
 * detection pass over all of them instead of n single-lane passes */
int32_t roam_engine_init_lanes_detect(roam_ctx *ctx, int32_t lane0, int32_t n, const int32_t *pool_idx, const double *poses3);
/* blocking: current feature set of a lane (cap rows), and its peak list */
int32_t roam_engine_lane_features(roam_ctx *ctx, int32_t lane, float *pts, int32_t cap, int32_t *K);
int32_t roam_engine_lane_peaks(roam_ctx *ctx, int32_t lane, int32_t *out, int64_t cap, int64_t *n);
/* a4 on a resident scan: DoH maxima of the float32 Cartesian warp of pool scan `pool_idx`
 * (appendNewFeatures(currImgCart, ...) of RawROAMSystem.py:264 without moving image data) */
int32_t roam_engine_doh_maxima(roam_ctx *ctx, int32_t pool_idx, const double *sigmas, int32_t num_sigma, double threshold,
                               int32_t *out_rcs, double *out_val, int32_t cap, int32_t *n_out);
/* blocking: level `level` (0..3) of the lane's most recent Cartesian u8 pyramid (w*h bytes, row-major) */
int32_t roam_engine_lane_image(roam_ctx *ctx, int32_t lane, int32_t level, uint8_t *out, int64_t cap);
/* replace a lane's feature set (retrack append, getFeatures.appendNewFeatures getFeatures.py:98-118) */
int32_t roam_engine_set_features(roam_ctx *ctx, int32_t lane, const float *pts, int32_t K);

/* ---- device-resident keyframe map (reference Mapping.Map.keyframes / addKeyframe,
 * Mapping.py:118-147; keyframes are created at RawROAMSystem.py:186-190 and :250-270).
 * Every keyframe of a lane stays in HBM.  The LIVE keyframe is the one the tracker prunes each frame
 * (Keyframe.pruneFeaturePoints, Mapping.py:118-125); when it is replaced - inside roam_engine_step when
 * the pose moved >= 0.2 rad / 2 m or the features ran out, or by roam_engine_set_features - its final
 * state {pose, velocity at creation, undistorted pruned locals, pool scan it was created on} is copied
 * device-to-device into the lane's ring.  Index 0 is the oldest keyframe, count-1 the live one.
 * A full ring drops further keyframes (count stays at keyframes_per_lane + 1). */
int32_t roam_engine_map_reserve(roam_ctx *ctx, int32_t keyframes_per_lane);
int32_t roam_engine_map_count(roam_ctx *ctx, int32_t lane, int32_t *count);
/* blocking read-back of one keyframe: locals_xy receives n (x, y) pairs in metres (prunedUndistortedLocals) */
int32_t roam_engine_map_get(roam_ctx *ctx, int32_t lane, int32_t index, double *pose3, double *vel3, double *locals_xy,
                            int32_t cap_pts, int32_t *n_out, int32_t *scan_out);
/* per-stage device time of the last step in milliseconds (hipEvent pairs on the stream);
 * names_out receives n pointers to static strings. */
int32_t roam_engine_stage_times(roam_ctx *ctx, float *ms_out, const char **names_out, int32_t cap,
                                int32_t *n);
/* the timestamp events behind roam_engine_stage_times and the "doh_*" figures of roam_engine_kernel_avg / _kernel_chunk_ms: recorded by
 * default (on = 1).  Thirteen timestamp packets in the back end's chain of dependent launches are nothing in a batch step and ~50 us of a
 * single-sequence pair (1 600 -> 1 740 scan-pairs/s without motion distortion): the streaming driver switches them off (on = 0), after which
 * those calls return ROAM_E_STATE.  ROAM_STAGE_EVENTS=0 / 1 in the environment of roam_engine_create overrides this call. */
int32_t roam_engine_set_stage_events(roam_ctx *ctx, int32_t on);
/* average in-step launch time (ms) of a front-end kernel ("ingest_peaks" | "warp_quantise" | "pyramid") over the
 * last `last_steps` steps (<= 64), from HIP event pairs recorded on the stream the kernel runs on; no
 * synchronisation happens inside the steps themselves.  "doh_integral" | "doh_det_maxima" (engines with
 * retrack_on_device): the image-scale kernels of the FIRST detection chunk of each of those steps, i.e. of
 * min(retrack_slots, lanes that re-detected in the step) detections; n_used counts the steps that had device-side
 * detection switched on */
int32_t roam_engine_kernel_avg(roam_ctx *ctx, const char *name, int32_t last_steps, float *avg_ms, int32_t *n_used);
/* "doh_integral" | "doh_det_maxima", chunk by chunk: ms_out[s * *chunks + c] = launch duration of chunk c in the s-th of the last
 * *steps_out (<= last_steps, <= 64) steps, oldest first; -1 for a step without device-side detection.  A step launches
 * ceil(lanes / retrack_slots) chunks (the first 16 are traced) whatever the number n of lanes that re-detect - only the device
 * knows it; chunk c holds clamp(n - c * retrack_slots, 0, retrack_slots) detections, n is in the step's result records.
 * cap (floats) >= last_steps * 16 always suffices.  "doh_integral": a chunk of fewer than 200 detections ran the two-pass kernels
 * (three times the one-sweep kernel's traffic; both forms are launched, the one whose regime it is not returns at once) - leave such
 * chunks out of a roofline average, as bench.py does */
int32_t roam_engine_kernel_chunk_ms(roam_ctx *ctx, const char *name, int32_t last_steps, float *ms_out, int32_t cap, int32_t *chunks,
                                    int32_t *steps_out);
/* detections per launch ("chunk") of a detection kernel inside a step: retrack_slots - or 1 024 when an engine of >= 2 048 lanes and
 * slots runs the determinants of a chunk on a second stream beside the next chunk's integral images (the default there; ROAM_DET_SIDE=0
 * in the environment of roam_engine_create: off).  The chunks of roam_engine_kernel_chunk_ms are these */
int32_t roam_engine_detect_chunk(roam_ctx *ctx, int32_t *chunk);
/* time `reps` launches of the dominant streaming kernel (warp+quantise of all lanes) with
 * HIP events on the context stream; returns average ms per launch. */
int32_t roam_engine_time_kernel(roam_ctx *ctx, const char *name, int32_t reps, float *avg_ms,
                                double *algo_bytes_per_launch);
/* diagnostics of the device-side detector (getFeatures.py:39-51: integral image -> Hessian determinants -> 3 x 3 x 3 maxima): runs the
 * image-scale kernels for n_slots detections (scratch slot i = the scan lane i % lanes stepped last) as two kernels (fused = 0: image in HBM)
 * or as the fused kernel (fused = 1: image never leaves the CU) and returns, per slot, the candidate count and the first cap_per_slot
 * candidates in (row, column, layer) order (rc = row << 16 | column << 2 | layer, val = determinant); S_out (optional, W x W float64) =
 * the integral image of slot s_slot as that form computed it.  Needs retrack_on_device and one step; n_slots <= retrack_slots, lanes. */
int32_t roam_engine_debug_detect(roam_ctx *ctx, int32_t fused, int32_t n_slots, int32_t cap_per_slot, uint32_t *rc_out, double *val_out,
                                 int32_t *n_out, int32_t s_slot, double *S_out);

/* ---- multi-GPU.  One process per GPU, sequences sharded by rank, no data-path collective.  The only exchange
 * of the path is handing a keyframe to a global map (reference Mapping.Map.addKeyframe, Mapping.py:118-147; BASELINE
 * config 5): the owning rank broadcasts the DEVICE-RESIDENT keyframe of one lane with ncclBroadcast (RCCL over xGMI).
 * RCCL is bound at run time (librccl.so); without it these calls return ROAM_E_STATE.
 * Rendezvous is the caller's business: rank 0 obtains the id, ships the 128 bytes to the other ranks by any means
 * (bench.py: a file), then every rank calls roam_comm_init collectively. */
#define ROAM_COMM_ID_BYTES 128
/* 1 if librccl.so can be bound in this process (nothing is initialised): lets the ranks agree on a fallback BEFORE any of them
 * enters the collective roam_comm_init */
int32_t roam_comm_available(void);
int32_t roam_comm_unique_id(uint8_t *id_out /* [ROAM_COMM_ID_BYTES] */);
int32_t roam_comm_init(roam_ctx *ctx, const uint8_t *id, int32_t rank, int32_t world);
int32_t roam_comm_destroy(roam_ctx *ctx);
/* rank / size as RCCL reports them (ncclCommUserRank / ncclCommCount) */
int32_t roam_comm_info(roam_ctx *ctx, int32_t *rank, int32_t *world);
/* blocking in-place all-reduce of n <= 8 doubles, op 0 = max, 1 = sum (max-over-ranks wall time); barrier = sum of ones */
int32_t roam_comm_allreduce_f64(roam_ctx *ctx, double *inout, int32_t n, int32_t op);
int32_t roam_comm_barrier(roam_ctx *ctx);

typedef struct roam_keyframe_hdr {
    double pose[3];           /* keyframe pose [x,y,th]                                   */
    double velocity[3];       /* velocity the keyframe's points were undistorted with      */
    int32_t n_features;       /* rows of locals_xy (prunedUndistortedLocals, metres)       */
    int32_t n_peaks;          /* rows of the polar point cloud [azimuthIdx, rangeIdx]      */
    int32_t scan;             /* pool scan the keyframe was created on                     */
    int32_t lane;             /* lane of the root rank it belongs to                       */
} roam_keyframe_hdr;

/* collective: every rank of the communicator calls it with the same root and lane.  The root packs the LIVE keyframe of
 * its lane `lane` {pose, velocity, prunedUndistortedLocals, latest polar peaks} on the device, all ranks receive it in a
 * device buffer (two ncclBroadcast calls: header + features, then the peak list), then copy it to the caller's arrays:
 * locals_xy (cap_pts, 2) f64, peaks (peaks_cap, 2) i32.  Either array may be NULL (its part is then not copied out). */
int32_t roam_bcast_keyframe(roam_ctx *ctx, int32_t root, int32_t lane, roam_keyframe_hdr *hdr_out, double *locals_xy,
                            int32_t cap_pts, int32_t *peaks, int64_t peaks_cap);

/* BASELINE config 5 as a loop (reference RawROAMSystem.py:250-262 + Mapping.Map.addKeyframe, Mapping.py:176-180: "after each
 * keyframe the owning GPU broadcasts"): collective and NON-BLOCKING, called by every rank once after each roam_engine_step with
 * its own lane.  Which rank has a new keyframe in a given step is only known on that rank's device, so the schedule is fixed: every
 * rank contributes ONE fixed-size record per step - its lane's new keyframe {header, prunedUndistortedLocals, up to 32768 polar
 * peaks} if the step made one (result flag bit 1), an empty record otherwise - to one ncclAllGather on the engine's exchange
 * stream, and appends every non-empty record it receives to its remote map on the device, in rank order.  Nothing waits on the host
 * and the step pipeline is not drained; roam_remote_map_count / _get wait for the exchange stream.  Needs roam_comm_init and
 * roam_remote_map_reserve; do not mix with roam_bcast_keyframe on one engine. */
int32_t roam_keyframe_exchange(roam_ctx *ctx, int32_t lane);

/* The consumer of that broadcast: Map.addKeyframe (Mapping.py:118-147) on EVERY rank.  After roam_remote_map_reserve(n) each
 * roam_bcast_keyframe also appends the received payload {header, prunedUndistortedLocals, polar peaks} device-to-device to a
 * ring of n keyframes in this rank's HBM (the oldest is overwritten); the sending rank included, so all ranks hold the same
 * global map.  count: keyframes received so far / resident now; get: index 0 = the oldest resident one. */
int32_t roam_remote_map_reserve(roam_ctx *ctx, int32_t keyframes);
/* test / debug: the receive half of roam_keyframe_exchange on ONE GPU, without a communicator.  recv (host) = `world` records as the
 * all-gather of a `world`-rank job leaves them in a rank's receive buffer (record r = rank r's; *rec_bytes apart; roam_keyframe_hdr at
 * 0, n_features x 2 float64 at *locals_off, n_peaks x 2 int32 at *peaks_off; n_features < 0 = "no keyframe in this step").  The records
 * go through the kernel the exchange runs (Map.addKeyframe in rank order, Mapping.py:176-180) into this context's remote map: read it
 * back with roam_remote_map_count / _get.  recv == NULL: only the layout is returned.  Needs roam_remote_map_reserve(>= world). */
int32_t roam_debug_keyframe_append(roam_ctx *ctx, const uint8_t *recv, int32_t world, int64_t *rec_bytes, int32_t *locals_off,
                                   int32_t *peaks_off, int32_t *max_peaks);
int32_t roam_remote_map_count(roam_ctx *ctx, int64_t *received, int32_t *resident);
int32_t roam_remote_map_get(roam_ctx *ctx, int32_t index, roam_keyframe_hdr *hdr_out, int32_t *root_out, double *locals_xy,
                            int32_t cap_pts, int32_t *peaks, int64_t peaks_cap);

#ifdef __cplusplus
}
#endif
#endif /* ROAM_ABI_H */
