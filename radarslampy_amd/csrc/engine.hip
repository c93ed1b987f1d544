// Batched, device-resident scan-pair engine.
//
// B independent sequences ("lanes") live in HBM: the raw Oxford records (pool), a ring of
// four u8 pyramids per lane (previous / current / in the pyramid stage / being warped), the
// tracked feature set, the keyframe state (+ optionally every past keyframe, 8f-f1) and the
// poses.  One roam_engine_step() advances EVERY lane by one scan pair with ~15 kernel launches
// and no host round trip, in three pipelined stages (+ the peak kernel on a stream of its own):
//   P (depends only on the raw scan; joined before the result record):  ingest+peaks
//   A (depends only on the raw scan; issue-bound):                      warp+quantise
//   B (HBM-bound):                                                      pyramid
//   C: KLT -> (status & err<10) compaction -> consistency graph -> max clique ->
//      inlier compaction + keyframe pruning + p_w / p_jt -> Kabsch -> initial transform ->
//      motion-distortion LM (latency-bound) -> pose / keyframe bookkeeping
// so that when steps are enqueued back to back A(N+2), B(N+1) and C(N) share the GPU.
// It restates the body of RawROAMSystem.run's loop (reference RawROAMSystem.py:162-298)
// without the plotting, Tracker.track (Tracker.py:35-106) and the Keyframe bookkeeping
// (Mapping.py:37-66,97-125,149-174).  Scan pairs of different lanes are independent, so the
// batch dimension is what fills the 256 CUs; within a lane the chain is sequential.
#include "roam_internal.h"
#include "retrack.h"
#include "kabsch_body.inc"
#include <algorithm>
#include <new>

#define KS ROAM_MAX_FEATURES
#define CART_CENTER 1012.0
#define M_PER_PX 0.0864
#define N_RETRACK 60             // getFeatures.py:57 (import-time binding used by RawROAMSystem.py:7,251)
#define ROT_THR 0.2              // Mapping.py:13
#define TRANS_THR_SQ 4.0         // Mapping.py:14-15
#define ERR_THR 10.0f            // getTransformKLT.py:84
#define TWO_PI 6.283185307179586476925286766559

#define RES_RING 8
enum { ST_PEAKS = 0, ST_WARP, ST_PYR, ST_KLT, ST_GRAPH, ST_CLIQUE, ST_KABSCH, ST_LM, ST_GLUE, ST_RETRACK, ST_COUNT };
static const char *kStageNames[ST_COUNT] = {"ingest_peaks", "warp_quantise", "pyramid", "klt", "consistency_graph",
                                            "max_clique", "kabsch", "mds_lm", "glue", "retrack"};

struct Engine {
    roam_engine_cfg cfg;
    int B = 0, W = 0, stage_cap = 0;
    PyrDesc pd;
    size_t rec_bytes = 0;
    uint8_t *pool = nullptr;
    uint8_t *pyr[4] = {nullptr, nullptr, nullptr, nullptr};   // ring: previous / current / in the pyramid stage / being warped
    uint32_t *warp_map = nullptr;       // W x W sampling map (geometry only)
    int cur = 0;                        // pyr[cur] = previous image pyramids
    uint16_t *row_stage = nullptr;
    int32_t *row_count = nullptr;
    int32_t *peaks_out[3] = {nullptr, nullptr, nullptr}, *peaks_n[3] = {nullptr, nullptr, nullptr};   // ring of 3 (stage A runs 2 steps ahead of g4)
    int32_t *scan_idx[3] = {nullptr, nullptr, nullptr};
    int32_t *scan_host = nullptr;       // pinned staging of the scan indices + new-sequence flags (3 x 2B)
    int pk = 0;                         // ring slot of the latest step (valid peak / scan-index buffers)
    int64_t nstep = 0;
    float *feat = nullptr;              // B x KS x 2
    int32_t *feat_n = nullptr;
    float *klt_next = nullptr, *klt_err = nullptr;
    uint8_t *klt_status = nullptr;
    float *good_old = nullptr, *good_new = nullptr;
    int32_t *good_idx = nullptr, *good_n = nullptr;
    uint64_t *adj = nullptr, *cq_stack = nullptr;
    int32_t *cq_order = nullptr;     // B: the step's clique problems, largest first (launch_max_clique)
    uint8_t *cq_mask = nullptr;
    int32_t *cq_n = nullptr, *cq_flags = nullptr;
    double *kab_src = nullptr, *kab_tgt = nullptr, *kab_out = nullptr;
    int32_t *in_n = nullptr;
    double *kf_pose = nullptr, *kf_und = nullptr, *kf_und_tmp = nullptr;   // B x 3, B x KS x 2
    // f1: device-resident keyframe map (Mapping.Map.keyframes): the live keyframe's creation velocity / scan,
    // and a per-lane ring of frozen keyframes {pose, velocity, n, scan | n x 2 undistorted locals}
    double *kf_vel = nullptr;                                              // B x 3
    int32_t *kf_scan = nullptr, *kf_fresh = nullptr, *kf_live = nullptr;   // B
    double *map_store = nullptr;                                           // B x map_cap x MAP_SLOT
    int32_t *map_n = nullptr;                                              // B
    int map_cap = 0;
    std::vector<int32_t> last_scan;                                        // host: scan of each lane's latest step
    double *pose = nullptr, *vel = nullptr;                                // B x 3
    double *T_wj0 = nullptr, *T_init = nullptr, *p_w = nullptr, *p_jt = nullptr;
    double *lm_work = nullptr, *lm_out = nullptr;
    int32_t *lm_nfev = nullptr, *lm_info = nullptr;
    hipEvent_t ev_lm[2] = {};            // fork / join of the LM's workgroup form on the peaks' stream (batches)
    int32_t *lm_big = nullptr;           // MdsProblemDesc::big: the solves left to the workgroup form, two alternating lists
    int lm_big_slot = 0;
    roam_lane_result *results = nullptr;           // ring of RES_RING per-step records (RES_RING x B)
    roam_lane_result *results_host = nullptr;      // pinned mirror, filled asynchronously after every step
    hipEvent_t ev_res[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_pool = nullptr;
    bool pool_dirty = false;
    RtArgs rt;                                      // device-side retrack (retrack.hip)
    bool rt_on = false;
    int rt_mode = 1;                                // 0 = suspended, 1 = lanes that ran out of features, 2 = every lane (measurement)
    uint8_t *kfb = nullptr;                         // 8e: packed keyframe payload (RCCL broadcast buffer)
    // 8e consumer: the global map of Mapping.Map.addKeyframe on EVERY rank - a ring of received keyframe payloads in HBM
    uint8_t *rmap = nullptr;
    int rmap_cap = 0;
    int64_t rmap_n = 0;                             // keyframes received so far (slot = index % rmap_cap)
    int64_t rmap_n_bcast = 0;                       // ... of which through roam_bcast_keyframe (host-side appends; never mixed with the exchange)
    std::vector<int32_t> rmap_root;                 // sending rank of each slot
    // per-step keyframe exchange (roam_keyframe_exchange): fixed-size records, one all-gather per step on its own stream, the
    // received keyframes appended on the device - the host never waits
    uint8_t *kfx_send = nullptr, *kfx_recv = nullptr;
    size_t kfx_rec = 0;                             // bytes of one record
    int kfx_peaks = 0;                              // peaks carried per record
    int kfx_world = 0;                              // ranks the receive buffer was sized for
    int64_t *rmap_n_dev = nullptr;                  // keyframes appended by the exchange (device-side count)
    int32_t *rmap_root_dev = nullptr;               // their sending ranks, per slot
    int64_t kfx_calls = 0;
    bool kfx_ready = false;                         // every resource of the exchange exists (set last: a failed set-up is retried, never half used)
    hipStream_t st_comm = nullptr;
    hipEvent_t ev_kfx[3] = {};                      // the PACK kernel of the exchange that read peak-ring slot i (the streams that overwrite its inputs wait for it)
    int kfx_last = -1;                              // slot of the latest pack
    hipEvent_t ev[ST_COUNT + 1] = {};
    bool stage_ev_forced = false;                   // ... the environment said so
    bool stage_ev = true;                           // record them (roam_engine_stage_times); ROAM_STAGE_EVENTS=0: ten timestamp packets fewer in the back end's chain
    hipEvent_t ev_join = nullptr, ev_pk0 = nullptr, ev_pk1 = nullptr;            // end of the front end (the back end waits for it) + the peak kernel's timing pair
    hipEvent_t ev_klt[4] = {}, ev_g4[4] = {};                // back-end milestones stage A of step N+3 waits for
    hipEvent_t ev_warp = nullptr;                            // end of stage A (stage B waits for it)
    hipEvent_t ev_idx = nullptr, ev_peaks = nullptr;                   // scan indices uploaded / peak kernel done
    // per-step boundaries of the three front-end kernels (before peaks | peaks/warp | warp/pyramid | after pyramid),
    // kept for the last TRACE_RING steps so that a caller can average a kernel's launch time over a timed
    // region without synchronising inside it (roam_engine_kernel_avg)
    hipEvent_t tr_ev[64][6] = {};
    hipEvent_t rt_ev[64][3 * RT_TRACE_CHUNKS] = {};                    // every detection chunk of a step (the first RT_TRACE_CHUNKS): before | integral image | determinants
    bool rt_ev_ok[64] = {};
    hipEvent_t ev_int = nullptr;                    // after the integral images of a step's (first) detection chunk
    hipEvent_t ev_emit = nullptr;                   // after the first bookkeeping kernels behind the determinants (ROAM_PYR_AFTER_EMIT / ROAM_PEAKS_AFTER_EMIT)
    int pyr_after_emit = 0, peaks_after_emit = 0;
    hipEvent_t ev_emit2[2] = {};                    // the same moment, alternating between consecutive steps (ROAM_SWAP_WARP_PYR: the warp of step N + 2 waits for step N's)
    int swap_warp_pyr = 0;
    RtSide det_side = {};                           // determinants of a chunk beside the next chunk's integral images (ROAM_DET_SIDE=chunk, 0: off)
    int det_chunk() const { return (det_side.chunk > 0 && retrack_sided(rt, B, &det_side)) ? det_side.chunk : rt.slots; }     // detections per launch of a detection kernel
    int lm_side = 0;                                // ROAM_LM_SIDE=1 (experiment): the LM's workgroup form on the peaks' stream beside the wave form instead of behind it
    hipEvent_t ev_emit_last = nullptr;              // (swap experiment) the event the last step recorded there
    bool ev_int_valid = false;
    int warp_after_int = 0;                         // ROAM_WARP_AFTER_INTEGRAL (experiment)
    int peaks_after_int = 0;                        // ROAM_PEAKS_AFTER_INTEGRAL (experiment): the peak kernel waits for the same event as the pyramid
    int pyr_after_int = 0;                          // ROAM_PYR_AFTER_INTEGRAL: the next pyramid waits for it (experiment, round 6)
    int64_t rt_image_px = 0;                        // pixels of the integral image that are written and read (the needed tiles of the phase list)
    bool tr_ev_ok[64] = {};                          // the step recorded its front-end event pairs (stage events were on when it was enqueued)
    int64_t stage_ev_step = -1;                     // the step whose ev[] (back-end stage events) are valid, -1: none
    bool tr_ok = false;
    unsigned long long *pyr_dark = nullptr;        // lanes of the 2024 -> 1012 pyramid kernel that see nothing but pixels beyond the maximum range
    static constexpr bool warp_dark_zero = true;   // the pyramids are zero-filled at creation and level 0 is written by the warp only:
                                                    // its tiles beyond the maximum range are never stored (0.6 ms of 11.4 per 4096 scans)
    bool ev_ok = false, stepped = false, uploads_pending = false;
    // asynchronous uploads are numbered; a step waits for the upload that last wrote one of ITS scans, not for the newest one - the
    // newest waits (roam_engine_fence) for the steps before it, and a front end that waited for it ran after the previous step's back end
    // instead of beside it (the single-sequence driver: ~170 us of every pair)
    uint64_t up_seq = 0, up_waited = 0;
    std::vector<uint64_t> slot_seq;
    hipEvent_t ev_up_ring[16] = {};
    std::vector<int> lane_k;            // host-side upper bound of each lane's feature count
    // device-side retracks (mode 1) grow a lane to at most 60 + 256 features without the host knowing which lane; a step in
    // mode 2 re-detects on EVERY lane, whatever it holds: the bound of every lane then grows by the 256 the append may add
    int rt_floor = 320;
    int kmax() const { int m = rt_on ? rt_floor : 64; for (int k : lane_k) m = k > m ? k : m; m = (m + 63) & ~63; return m > KS ? KS : m; }
    std::vector<void *> allocs;
};

template <typename T>
static bool dalloc(roam_ctx *ctx, Engine *e, T **p, size_t count)
{
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    hipError_t err = hipMalloc(&q, bytes);
    if (err != hipSuccess) {
        ROAM_SET_ERR(ctx, "engine: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
        return false;
    }
    hipMemsetAsync(q, 0, bytes, ctx->stream);
    e->allocs.push_back(q);
    *p = (T *)q;
    return true;
}

// ------------------------------------------------------------------------------ glue kernels
__device__ __forceinline__ int blk_excl_scan(int v, int *sh, int *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int n = __shfl_up(inc, d);
        if (lane >= d) inc += n;
    }
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; i++) {
        int s = sh[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ void new_sequence_kernel(int32_t *__restrict__ feat_n, const int32_t *__restrict__ flag, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && flag[b]) feat_n[b] = 0;
}

// G1: status &= err < ERR_THRESHOLD (getTransformKLT.py:365), ordered compaction of the good pairs
__global__ __launch_bounds__(256) void g1_good_kernel(const float *__restrict__ feat, const int32_t *__restrict__ feat_n,
                                                      const float *__restrict__ klt_next, uint8_t *__restrict__ status,
                                                      const float *__restrict__ err, float *__restrict__ good_old,
                                                      float *__restrict__ good_new, int32_t *__restrict__ good_idx,
                                                      int32_t *__restrict__ good_n, int kmax_launch)
{
    __shared__ int sh[8];
    const int b = blockIdx.x, t = threadIdx.x;
    const int K = min(feat_n[b], kmax_launch);          // (the tracker ran on kmax_launch features at most)
    const int items = (KS + 255) / 256;
    const int lo = t * items, hi = min(lo + items, K);
    int c = 0;
    for (int k = lo; k < hi; k++) {
        const int64_t i = (int64_t)b * KS + k;
        uint8_t s = status[i] & (uint8_t)(err[i] < ERR_THR);
        status[i] = s;
        c += s;
    }
    int total;
    int pos = blk_excl_scan(c, sh, &total);
    for (int k = lo; k < hi; k++) {
        const int64_t i = (int64_t)b * KS + k;
        if (status[i]) {
            const int64_t o = (int64_t)b * KS + pos;
            good_old[2 * o] = feat[2 * i]; good_old[2 * o + 1] = feat[2 * i + 1];
            good_new[2 * o] = klt_next[2 * i]; good_new[2 * o + 1] = klt_next[2 * i + 1];
            good_idx[o] = k;
            pos++;
        }
    }
    if (t == 0) good_n[b] = total;
}

__global__ void fill_mask_kernel(uint8_t *mask, const int32_t *good_n, int32_t *cq_n, int32_t *cq_flags)
{
    const int b = blockIdx.x;
    for (int k = threadIdx.x; k < KS; k += blockDim.x) mask[(int64_t)b * KS + k] = 1;
    if (threadIdx.x == 0) { cq_n[b] = good_n[b]; cq_flags[b] = 1; }
}

// G2: inlier compaction (Tracker.py:93-104), keyframe pruning (Mapping.py:118-125), Kabsch inputs,
//     p_w (Mapping.py:97-116), centered_new (RawROAMSystem.py:198-199), next feature set (:296)
__global__ __launch_bounds__(256) void g2_inliers_kernel(const float *__restrict__ good_old, const float *__restrict__ good_new,
                                                         const int32_t *__restrict__ good_idx, const int32_t *__restrict__ good_n,
                                                         const uint8_t *__restrict__ mask, const double *__restrict__ kf_pose,
                                                         const double *__restrict__ kf_und, double *__restrict__ kf_und_tmp,
                                                         double *__restrict__ kab_src, double *__restrict__ kab_tgt,
                                                         double *__restrict__ p_w, double *__restrict__ p_jt,
                                                         float *__restrict__ feat, int32_t *__restrict__ in_n,
                                                         double *__restrict__ kab_out, const double *__restrict__ pose,
                                                         double *__restrict__ T_wj0, double *__restrict__ T_init)
{
    __shared__ int sh[8];
    __shared__ double red[8];
    const int b = blockIdx.x, t = threadIdx.x;
    const int G = good_n[b];
    const int items = (KS + 255) / 256;
    const int lo = t * items, hi = min(lo + items, G);
    int c = 0;
    for (int g = lo; g < hi; g++) c += mask[(int64_t)b * KS + g] ? 1 : 0;
    int total;
    int pos = blk_excl_scan(c, sh, &total);
    const double kx = kf_pose[3 * b], ky = kf_pose[3 * b + 1], kth = kf_pose[3 * b + 2];
    const double ck = cos(kth), sk = sin(kth);
    for (int g = lo; g < hi; g++) {
        const int64_t i = (int64_t)b * KS + g;
        if (!mask[i]) continue;
        const int64_t o = (int64_t)b * KS + pos;
        const float ox = good_old[2 * i], oy = good_old[2 * i + 1];
        const float nx = good_new[2 * i], ny = good_new[2 * i + 1];
        kab_src[2 * o] = (double)ox; kab_src[2 * o + 1] = (double)oy;
        kab_tgt[2 * o] = (double)nx; kab_tgt[2 * o + 1] = (double)ny;
        const int64_t ki = (int64_t)b * KS + good_idx[i];
        const double ux = kf_und[2 * ki], uy = kf_und[2 * ki + 1];
        kf_und_tmp[2 * o] = ux; kf_und_tmp[2 * o + 1] = uy;
        p_w[2 * o] = ck * ux - sk * uy + kx;
        p_w[2 * o + 1] = sk * ux + ck * uy + ky;
        p_jt[2 * o] = ((double)nx - CART_CENTER) * M_PER_PX;
        p_jt[2 * o + 1] = ((double)ny - CART_CENTER) * M_PER_PX;
        pos++;
    }
    __syncthreads();
    // blobCoord = good_new.copy(): write after every read of good_* (distinct buffers, so no hazard)
    pos -= c;
    for (int g = lo; g < hi; g++) {
        const int64_t i = (int64_t)b * KS + g;
        if (!mask[i]) continue;
        const int64_t o = (int64_t)b * KS + pos;
        feat[2 * o] = good_new[2 * i]; feat[2 * o + 1] = good_new[2 * i + 1];
        pos++;
    }
    if (t == 0) in_n[b] = total;
    // The Kabsch fit of the lane's inliers (getTransformKLT.calculateTransformSVD) and, with motion distortion, the LM's two transforms
    // (G3) by the same workgroup: kernels of their own until round 6 - two launches less in the chain of a step (a single sequence's pair is
    // a chain of ~20 launches of 4-6 us each on the device and on the enqueuing thread)
    __threadfence_block();
    __syncthreads();
    double *ko = kab_out + 6 * (int64_t)b;
    kabsch_body(kab_src + (int64_t)b * KS * 2, kab_tgt + (int64_t)b * KS * 2, total, red, ko);
    if (T_wj0 && t == 0) {
        const double x = pose[3 * b], y = pose[3 * b + 1], th = pose[3 * b + 2];
        const double c = cos(th), s = sin(th);
        double *T0 = T_wj0 + 9 * (int64_t)b, *Ti = T_init + 9 * (int64_t)b;
        T0[0] = c; T0[1] = -s; T0[2] = x; T0[3] = s; T0[4] = c; T0[5] = y; T0[6] = 0; T0[7] = 0; T0[8] = 1;
        const double hx = ko[4] * M_PER_PX, hy = ko[5] * M_PER_PX;
        Ti[0] = c * ko[0] - s * ko[2]; Ti[1] = c * ko[1] - s * ko[3]; Ti[2] = c * hx - s * hy + x;
        Ti[3] = s * ko[0] + c * ko[2]; Ti[4] = s * ko[1] + c * ko[3]; Ti[5] = s * hx + c * hy + y;
        Ti[6] = 0; Ti[7] = 0; Ti[8] = 1;
    }
}

// (G3 - h *= 0.0864 (Tracker.py:124-125); T_wj = prev_pose @ [[R,h],[0,0,1]] (RawROAMSystem.py:201) - is the tail of g2_inliers_kernel)

// f1 keyframe map: one slot = 8-double header {pose[3], velocity[3], n, scan} + KS x 2 undistorted locals
#define MAP_SLOT (8 + 2 * KS)
// copy a keyframe that is about to be replaced into the lane's ring (whole block; returns through *ok)
__device__ __forceinline__ void map_freeze(double *__restrict__ map_store, int32_t *__restrict__ map_n, int map_cap, int b,
                                           const double *__restrict__ kfp, const double *__restrict__ kfv, int n,
                                           int scan, const double *__restrict__ locals)
{
    const int slot = map_n[b];
    if (slot >= map_cap) return;                          // ring full: the keyframe is dropped, count stays at cap
    double *d = map_store + ((size_t)b * map_cap + slot) * MAP_SLOT;
    for (int j = threadIdx.x; j < 2 * n; j += blockDim.x) d[8 + j] = locals[j];
    if (threadIdx.x == 0) {
        d[0] = kfp[0]; d[1] = kfp[1]; d[2] = kfp[2];
        d[3] = kfv[0]; d[4] = kfv[1]; d[5] = kfv[2];
        d[6] = (double)n; d[7] = (double)scan;
    }
}

// host-initiated keyframe replacement (roam_engine_set_features): freeze the live keyframe unless the step that
// has just run created it (then g4 already froze its predecessor and the caller is amending the new one)
__global__ __launch_bounds__(256) void map_freeze_kernel(double *__restrict__ map_store, int32_t *__restrict__ map_n, int map_cap,
                                                         int b, const double *__restrict__ kf_pose,
                                                         const double *__restrict__ kf_vel, const int32_t *__restrict__ feat_n,
                                                         const int32_t *__restrict__ kf_scan, const double *__restrict__ kf_und,
                                                         const int32_t *__restrict__ kf_live, const int32_t *__restrict__ kf_fresh)
{
    if (!kf_live[b] || kf_fresh[b]) return;
    const int slot = map_n[b];
    map_freeze(map_store, map_n, map_cap, b, kf_pose + 3 * b, kf_vel + 3 * b, feat_n[b], kf_scan[b], kf_und + (size_t)b * KS * 2);
    __syncthreads();
    if (threadIdx.x == 0 && slot < map_cap) map_n[b] = slot + 1;
}

// G4: pose / velocity update, keyframe criteria (Mapping.py:149-174, RawROAMSystem.py:250-271),
//     possible_kf.updateInfo undistortion (Mapping.py:65), result record
__global__ __launch_bounds__(256) void g4_update_kernel(roam_engine_cfg cfg, const double *__restrict__ lm_out,
                                                        const int32_t *__restrict__ lm_nfev, const int32_t *__restrict__ lm_info,
                                                        const double *__restrict__ kab_out, double *__restrict__ pose,
                                                        double *__restrict__ vel, double *__restrict__ kf_pose,
                                                        double *__restrict__ kf_und, const double *__restrict__ kf_und_tmp,
                                                        const double *__restrict__ p_jt, const int32_t *__restrict__ in_n,
                                                        const int32_t *__restrict__ good_n, int32_t *__restrict__ feat_n,
                                                        const int32_t *__restrict__ peaks_n, const int32_t *__restrict__ cq_flags,
                                                        roam_lane_result *__restrict__ res, const int32_t *__restrict__ scan_idx,
                                                        double *__restrict__ kf_vel, int32_t *__restrict__ kf_scan,
                                                        int32_t *__restrict__ kf_fresh, const int32_t *__restrict__ kf_live,
                                                        double *__restrict__ map_store, int32_t *__restrict__ map_n, int map_cap,
                                                        int collect, int32_t *__restrict__ rt_lane, int32_t *__restrict__ rt_scan,
                                                        int32_t *__restrict__ rt_n)
{
    __shared__ double np_[3], nv_[3];
    __shared__ int newkf;
    const int b = blockIdx.x, t = threadIdx.x;
    const int n = in_n[b];
    if (t == 0) {
        double px = pose[3 * b], py = pose[3 * b + 1], pth = pose[3 * b + 2];
        double v0 = 0, v1 = 0, v2 = 0;
        const double *k = kab_out + 6 * (int64_t)b;
        if (n >= 2) {
            if (cfg.motion_distortion) {
                const double *s = lm_out + 6 * (int64_t)b;
                v0 = s[0]; v1 = s[1]; v2 = s[2]; px = s[3]; py = s[4]; pth = s[5];
            } else {
                // updateTrajectory: convertRandHtoDeltas + appendRelativeDeltas (utils.py:99-103, trajectoryPlotting.py:28-35)
                const double dx = k[4] * M_PER_PX, dy = k[5] * M_PER_PX, dth = atan2(k[2], k[0]);
                const double c = cos(pth), s = sin(pth);
                px += dx * c - dy * s; py += dx * s + dy * c; pth += dth;
            }
        }
        np_[0] = px; np_[1] = py; np_[2] = pth; nv_[0] = v0; nv_[1] = v1; nv_[2] = v2;
        const int retrack = n <= N_RETRACK;
        const double dth = fabs(kf_pose[3 * b + 2] - pth);
        const double ddx = kf_pose[3 * b] - px, ddy = kf_pose[3 * b + 1] - py;
        const int good = (dth >= cfg.keyframe_rot_rad) || (ddx * ddx + ddy * ddy >= cfg.keyframe_trans_m * cfg.keyframe_trans_m);
        newkf = retrack || good;
        roam_lane_result *r = res + b;
        r->pose[0] = px; r->pose[1] = py; r->pose[2] = pth;
        r->velocity[0] = v0; r->velocity[1] = v1; r->velocity[2] = v2;
        r->kabsch_R[0] = k[0]; r->kabsch_R[1] = k[1]; r->kabsch_R[2] = k[2]; r->kabsch_R[3] = k[3];
        r->kabsch_h[0] = k[4] * M_PER_PX; r->kabsch_h[1] = k[5] * M_PER_PX;
        r->n_tracked = feat_n[b]; r->n_good = good_n[b]; r->n_inliers = n; r->n_peaks = peaks_n[b];
        r->lm_nfev = (n >= 2 && cfg.motion_distortion) ? lm_nfev[b] : 0;
        r->lm_info = (n >= 2 && cfg.motion_distortion) ? lm_info[b] : 0;
        // (a device-scope atomic store: the last block of this kernel reads every lane's flags, possibly from another XCD - see below)
        __hip_atomic_store(&r->flags, (cq_flags[b] & 1) | (newkf ? 2 : 0) | (retrack ? 4 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r->n_after_retrack = 0;
    }
    __syncthreads();
    const double v0 = nv_[0], v1 = nv_[1], v2 = nv_[2];
    // a keyframe that is replaced keeps its final state (pose, creation velocity, pruned locals) in the map
    const int mslot = (newkf && map_cap > 0 && kf_live[b]) ? map_n[b] : -1;
    if (mslot >= 0) map_freeze(map_store, map_n, map_cap, b, kf_pose + 3 * b, kf_vel + 3 * b, n, kf_scan[b], kf_und_tmp + (int64_t)b * KS * 2);
    for (int j = t; j < n; j += 256) {
        const int64_t o = (int64_t)b * KS + j;
        if (newkf) {
            const double x = p_jt[2 * o], y = p_jt[2 * o + 1];
            const double dT = 0.25 * atan2(-y, -x) / TWO_PI;
            const double a = v2 * dT, ca = cos(a), sa = sin(a);
            kf_und[2 * o] = ca * x - sa * y + v0 * dT;
            kf_und[2 * o + 1] = sa * x + ca * y + v1 * dT;
        } else {
            kf_und[2 * o] = kf_und_tmp[2 * o]; kf_und[2 * o + 1] = kf_und_tmp[2 * o + 1];
        }
    }
    __syncthreads();
    if (t == 0) {
        pose[3 * b] = np_[0]; pose[3 * b + 1] = np_[1]; pose[3 * b + 2] = np_[2];
        vel[3 * b] = v0; vel[3 * b + 1] = v1; vel[3 * b + 2] = v2;
        if (newkf) {
            kf_pose[3 * b] = np_[0]; kf_pose[3 * b + 1] = np_[1]; kf_pose[3 * b + 2] = np_[2];
            kf_vel[3 * b] = v0; kf_vel[3 * b + 1] = v1; kf_vel[3 * b + 2] = v2;
            kf_scan[b] = scan_idx[b];
            if (mslot >= 0 && mslot < map_cap) map_n[b] = mslot + 1;
        }
        kf_fresh[b] = newkf;
        feat_n[b] = n;
    }
    // The list of the lanes that re-detect (rt_collect_kernel's job: lane order, so one block) by the block that finishes LAST: one launch
    // less in the chain every step enqueues (a kernel that only returns costs ~4-6 us of a single sequence's 290 us pair).
    // collect: 0 none, 1 lanes with flag bit 2, 2 every lane; rt_n[1] counts the finished blocks and is left at zero
    if (collect) {
        __shared__ int last_s, sh[4], base_s;
        // (no __threadfence(): on this GPU it writes the XCD's whole L2 back; the flags travel as device-scope atomics, the counter after them)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) last_s = atomicAdd(rt_n + 1, 1) == (int)gridDim.x - 1;
        __syncthreads();
        if (!last_s) return;
        const int B = (int)gridDim.x;
        if (t == 0) base_s = 0;
        __syncthreads();
        for (int b0 = 0; b0 < B; b0 += 256) {
            const int bb = b0 + t;
            const int f = (bb < B && (collect == 2 || (__hip_atomic_load(&res[bb].flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4))) ? 1 : 0;
            const int lane = t & 63, w = t >> 6;
            int inc = f;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int nn = __shfl_up(inc, d); if (lane >= d) inc += nn; }
            if (lane == 63) sh[w] = inc;
            __syncthreads();
            int off = base_s, tot = 0;
            for (int i = 0; i < 4; i++) { if (i < w) off += sh[i]; tot += sh[i]; }
            if (f) { rt_lane[off + inc - 1] = bb; rt_scan[off + inc - 1] = scan_idx[bb]; }
            __syncthreads();
            if (t == 0) base_s += tot;
            __syncthreads();
        }
        if (t == 0) { rt_n[0] = base_s; rt_n[1] = 0; }
    }
}

// features -> keyframe locals: undistort(velocity, (pts - center) * m/px) (Mapping.py:59-66)
__global__ void lane_kf_reset_kernel(const float *__restrict__ feat, int K, const double *__restrict__ vel,
                                     double *__restrict__ kf_und)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= K) return;
    const double x = ((double)feat[2 * j] - CART_CENTER) * M_PER_PX, y = ((double)feat[2 * j + 1] - CART_CENTER) * M_PER_PX;
    const double dT = 0.25 * atan2(-y, -x) / TWO_PI;
    const double a = vel[2] * dT, ca = cos(a), sa = sin(a);
    kf_und[2 * j] = ca * x - sa * y + vel[0] * dT;
    kf_und[2 * j + 1] = sa * x + ca * y + vel[1] * dT;
}

// 8e: keyframe payload of one lane for the RCCL broadcast, packed on the device:
//   [0..63]   roam_keyframe_hdr (pose, velocity, n_features, n_peaks, scan, lane)
//   [64.. ]   KS x 2 f64 prunedUndistortedLocals
//   [64 + KS*16 ..]  peaks_cap x 2 i32 polar peaks of the lane's latest scan
#define KFB_HDR 64
#define KFB_LOCALS_OFF KFB_HDR
#define KFB_PEAKS_OFF (KFB_HDR + KS * 16)
__global__ __launch_bounds__(256) void kf_pack_kernel(uint8_t *__restrict__ buf, int lane, const double *__restrict__ kf_pose,
                                                      const double *__restrict__ kf_vel, const int32_t *__restrict__ feat_n,
                                                      const int32_t *__restrict__ kf_scan, const double *__restrict__ kf_und,
                                                      const int32_t *__restrict__ peaks_n, const int32_t *__restrict__ peaks,
                                                      int peaks_cap)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    int n = feat_n[lane], P = peaks_n[lane];
    if (n > KS) n = KS;
    if (P > peaks_cap) P = peaks_cap;
    if (t == 0) {
        roam_keyframe_hdr *h = reinterpret_cast<roam_keyframe_hdr *>(buf);
        for (int i = 0; i < 3; i++) { h->pose[i] = kf_pose[3 * lane + i]; h->velocity[i] = kf_vel[3 * lane + i]; }
        h->n_features = n; h->n_peaks = P; h->scan = kf_scan[lane]; h->lane = lane;
    }
    double *loc = reinterpret_cast<double *>(buf + KFB_LOCALS_OFF);
    const double *src = kf_und + (size_t)lane * KS * 2;
    for (int j = t; j < 2 * n; j += nt) loc[j] = src[j];
    int32_t *pk = reinterpret_cast<int32_t *>(buf + KFB_PEAKS_OFF);
    const int32_t *ps = peaks + (size_t)lane * peaks_cap * 2;
    for (int j = t; j < 2 * P; j += nt) pk[j] = ps[j];
}

// per-step exchange: this rank's record = the live keyframe of `lane` if the step just made it one (result flag bit 1), else an
// empty record (n_features = -1); peaks are capped at `pk_cap` (n_peaks says how many travelled)
__global__ __launch_bounds__(256) void kfx_pack_kernel(uint8_t *__restrict__ buf, int lane, const roam_lane_result *__restrict__ res,
                                                       const double *__restrict__ kf_pose, const double *__restrict__ kf_vel,
                                                       const int32_t *__restrict__ feat_n, const int32_t *__restrict__ kf_scan,
                                                       const double *__restrict__ kf_und, const int32_t *__restrict__ peaks_n,
                                                       const int32_t *__restrict__ peaks, int peaks_cap, int pk_cap)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    const bool valid = (res[lane].flags & 2) != 0;
    int n = feat_n[lane], P = peaks_n[lane];
    if (n > KS) n = KS;
    if (P > pk_cap) P = pk_cap;
    if (t == 0) {
        roam_keyframe_hdr *h = reinterpret_cast<roam_keyframe_hdr *>(buf);
        for (int i = 0; i < 3; i++) { h->pose[i] = kf_pose[3 * lane + i]; h->velocity[i] = kf_vel[3 * lane + i]; }
        h->n_features = valid ? n : -1; h->n_peaks = valid ? P : 0; h->scan = kf_scan[lane]; h->lane = lane;
    }
    if (!valid) return;
    double *loc = reinterpret_cast<double *>(buf + KFB_LOCALS_OFF);
    const double *src = kf_und + (size_t)lane * KS * 2;
    for (int j = t; j < 2 * n; j += nt) loc[j] = src[j];
    int32_t *pk = reinterpret_cast<int32_t *>(buf + KFB_PEAKS_OFF);
    const int32_t *ps = peaks + (size_t)lane * peaks_cap * 2;
    for (int j = t; j < 2 * P; j += nt) pk[j] = ps[j];
}

// Map.addKeyframe on this rank for every non-empty record of the gathered buffer, in rank order (one workgroup: the slot of
// record r depends on the valid records before it)
__global__ __launch_bounds__(256) void kfx_append_kernel(const uint8_t *__restrict__ recv, int world, size_t rec_bytes,
                                                         uint8_t *__restrict__ rmap, size_t slot_bytes, int cap,
                                                         int64_t *__restrict__ n_dev, int32_t *__restrict__ root_dev, int max_feat, int max_peaks)
{
    __shared__ int64_t base;
    if (threadIdx.x == 0) base = *n_dev;
    __syncthreads();
    int64_t cnt = base;
    for (int r = 0; r < world; r++) {
        const uint8_t *src = recv + (size_t)r * rec_bytes;
        const roam_keyframe_hdr *h = reinterpret_cast<const roam_keyframe_hdr *>(src);
        const int n = h->n_features, P = h->n_peaks;
        if (n < 0) continue;                                                 // the rank made no keyframe in this step
        if (n > max_feat || P < 0 || P > max_peaks) continue;               // (a record no pack kernel writes: never copied past a slot)
        uint8_t *dst = rmap + (size_t)(cnt % cap) * slot_bytes;
        const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src);
        uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
        const size_t w_hdr = (KFB_HDR + (size_t)n * 16) / 4, w_pk0 = KFB_PEAKS_OFF / 4, w_pk = (size_t)P * 2;
        for (size_t j = threadIdx.x; j < w_hdr; j += blockDim.x) d4[j] = s4[j];
        for (size_t j = threadIdx.x; j < w_pk; j += blockDim.x) d4[w_pk0 + j] = s4[w_pk0 + j];
        if (threadIdx.x == 0) root_dev[cnt % cap] = r;
        cnt++;
    }
    __syncthreads();
    if (threadIdx.x == 0) *n_dev = cnt;
}

// f2 ingest: record i, rows 4 per workgroup (one wavefront per row): bytes [0, width) of every row from pinned host memory
// into the pool.  Source rows start at arbitrary alignment (stride 3779), so a lane loads 16 bytes at byte granularity
// through a packed unaligned vector type and stores them unaligned as well; the tail is copied byte-wise.
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
__global__ __launch_bounds__(256) void ingest_rows_kernel(const uint8_t *__restrict__ host, int64_t host_stride, uint8_t *__restrict__ pool,
                                                          int64_t rec_bytes, int rows, int stride, int width)
{
    const int rec = blockIdx.y, row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const uint8_t *s = host + (int64_t)rec * host_stride + (int64_t)row * stride;
    uint8_t *d = pool + (int64_t)rec * rec_bytes + (int64_t)row * stride;
    const int nvec = width / 16;
    for (int v = lane; v < nvec; v += 64) *reinterpret_cast<u32x4_a1 *>(d + 16 * v) = *reinterpret_cast<const u32x4_a1 *>(s + 16 * v);
    for (int b = nvec * 16 + lane; b < width; b += 64) d[b] = s[b];
}

// ------------------------------------------------------------------------------ API
extern "C" {

int32_t roam_engine_destroy(roam_ctx *ctx)
{
    if (!ctx) return ROAM_E_ARG;
    Engine *e = ctx->engine;
    if (!e) return ROAM_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (void *p : e->allocs) hipFree(p);
    // every event handle starts out null, so a creation that failed half way leaks nothing
    auto kill = [](hipEvent_t &ev) { if (ev) { hipEventDestroy(ev); ev = nullptr; } };
    for (auto &ev : e->ev) kill(ev);
    for (int i = 0; i < 4; i++) { kill(e->det_side.ev_i[i]); kill(e->det_side.ev_d[i]); }
    kill(e->ev_int); kill(e->ev_emit); kill(e->ev_emit2[0]); kill(e->ev_emit2[1]); kill(e->ev_lm[0]); kill(e->ev_lm[1]);
    kill(e->ev_join); kill(e->ev_pk0); kill(e->ev_pk1); kill(e->ev_warp); kill(e->ev_idx); kill(e->ev_peaks);
    for (int i = 0; i < 4; i++) { kill(e->ev_klt[i]); kill(e->ev_g4[i]); }
    for (auto &row : e->tr_ev) for (auto &ev : row) kill(ev);
    for (auto &row : e->rt_ev) for (auto &ev : row) kill(ev);
    if (e->scan_host) hipHostFree(e->scan_host);
    if (e->results_host) hipHostFree(e->results_host);
    for (auto &ev : e->ev_res) if (ev) hipEventDestroy(ev);
    if (e->st_comm) { hipStreamSynchronize(e->st_comm); hipStreamDestroy(e->st_comm); }
    for (auto &ev : e->ev_kfx) if (ev) hipEventDestroy(ev);
    if (e->ev_pool) hipEventDestroy(e->ev_pool);
    for (auto &ev : e->ev_up_ring) if (ev) hipEventDestroy(ev);
    hipStreamSynchronize(ctx->stream2);
    hipStreamSynchronize(ctx->stream4);
    delete e;
    ctx->engine = nullptr;
    return ROAM_OK;
}

// the fused detection variant's tables and per-slot scratch, made when first needed (dead weight otherwise)
static int32_t fused_tables(roam_ctx *ctx, Engine *e)
{
    RtArgs &r = e->rt;
    if (r.fd_mapT) return ROAM_OK;
    const size_t npx = (size_t)e->W * e->W, R = (size_t)r.slots;
    uint32_t *mapT = nullptr, *boxtab = nullptr, *darktab = nullptr;
    bool ok = dalloc(ctx, e, &mapT, npx) && dalloc(ctx, e, &boxtab, retrack_fused_boxtab_words(e->W)) && dalloc(ctx, e, &darktab, retrack_darktab_words(e->W));
    ok = ok && dalloc(ctx, e, &r.fd_halo, R * 2 * (size_t)r.fd_halo_words) && dalloc(ctx, e, &r.fd_cc, R * 2048);
    if (!ok) return ROAM_E_HIP;
    HIP_TRY(ctx, launch_retrack_fused_tables(ctx->stream, e->warp_map, e->W, e->cfg.clip, mapT, boxtab, darktab));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    r.fd_boxtab = boxtab; r.fd_darktab = darktab;
    r.fd_mapT = mapT;                                   // (set last: the null check of launch_retrack_part is the "ready" test)
    return ROAM_OK;
}

int32_t roam_engine_create(roam_ctx *ctx, const roam_engine_cfg *cfg)
{
    if (!ctx) return ROAM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ARG_CHECK(ctx, cfg && cfg->lanes >= 1 && cfg->rows >= 1 && cfg->rows <= 1020 && cfg->clip >= 32 && cfg->clip <= 4094 && (cfg->clip / 2) % 2 == 0 &&
                       (int64_t)cfg->rows * cfg->stride < (1ll << 30) &&
                       cfg->payload_off >= 0 && cfg->stride >= cfg->payload_off + cfg->clip && cfg->pool_scans >= 1 &&
                       cfg->peaks_cap >= 1);
    roam_engine_destroy(ctx);
    Engine *e = new (std::nothrow) Engine();
    if (!e) return ROAM_E_HIP;
    ctx->engine = e;
    e->cfg = *cfg;
    if (!(e->cfg.keyframe_trans_m > 0.0)) e->cfg.keyframe_trans_m = 2.0;        // Mapping.py:14
    if (!(e->cfg.keyframe_rot_rad > 0.0)) e->cfg.keyframe_rot_rad = ROT_THR;      // Mapping.py:13
    const int B = e->B = cfg->lanes;
    e->lane_k.assign(B, 0);
    e->last_scan.assign(B, -1);
    e->W = 2 * (cfg->clip / 2);
    e->stage_cap = (cfg->clip + 1) / 2;
    pyr_desc_init(&e->pd, e->W, e->W);
    e->rec_bytes = (size_t)cfg->rows * cfg->stride;
    const int nw = KS / 64;
    bool ok = true;
    ok = ok && dalloc(ctx, e, &e->pool, e->rec_bytes * cfg->pool_scans + 64);      // + slack: the warp stages its boxes with dword loads
    ok = ok && dalloc(ctx, e, &e->pyr[0], (size_t)e->pd.lane_stride * B);
    ok = ok && dalloc(ctx, e, &e->pyr[1], (size_t)e->pd.lane_stride * B);
    ok = ok && dalloc(ctx, e, &e->pyr[2], (size_t)e->pd.lane_stride * B);
    ok = ok && dalloc(ctx, e, &e->pyr[3], (size_t)e->pd.lane_stride * B);
    ok = ok && dalloc(ctx, e, &e->pyr_dark, pyr_dark_words(e->W));
    ok = ok && dalloc(ctx, e, &e->warp_map, (size_t)e->W * e->W);
    ok = ok && dalloc(ctx, e, &e->row_stage, (size_t)B * cfg->rows * e->stage_cap);
    ok = ok && dalloc(ctx, e, &e->row_count, (size_t)B * cfg->rows);
    for (int i = 0; i < 3; i++) {
        ok = ok && dalloc(ctx, e, &e->peaks_out[i], (size_t)B * cfg->peaks_cap * 2);
        ok = ok && dalloc(ctx, e, &e->peaks_n[i], (size_t)B);
        ok = ok && dalloc(ctx, e, &e->scan_idx[i], 2 * (size_t)B);     // [0, B): pool scan of each lane; [B, 2B): 1 = the lane starts a new sequence
    }
    if (ok && hipHostMalloc(reinterpret_cast<void **>(&e->scan_host), sizeof(int32_t) * 6 * (size_t)B, hipHostMallocDefault) != hipSuccess) {
        ROAM_SET_ERR(ctx, "engine: hipHostMalloc failed"); ok = false;
    }
    ok = ok && dalloc(ctx, e, &e->feat, (size_t)B * KS * 2);
    ok = ok && dalloc(ctx, e, &e->feat_n, (size_t)B);
    ok = ok && dalloc(ctx, e, &e->klt_next, (size_t)B * KS * 2);
    ok = ok && dalloc(ctx, e, &e->klt_err, (size_t)B * KS);
    ok = ok && dalloc(ctx, e, &e->klt_status, (size_t)B * KS);
    ok = ok && dalloc(ctx, e, &e->good_old, (size_t)B * KS * 2);
    ok = ok && dalloc(ctx, e, &e->good_new, (size_t)B * KS * 2);
    ok = ok && dalloc(ctx, e, &e->good_idx, (size_t)B * KS);
    ok = ok && dalloc(ctx, e, &e->good_n, (size_t)B);
    ok = ok && dalloc(ctx, e, &e->adj, (size_t)B * KS * nw);
    ok = ok && dalloc(ctx, e, &e->cq_stack, (size_t)B * (KS + 2) * 2 * nw);
    ok = ok && dalloc(ctx, e, &e->cq_order, (size_t)B);
    ok = ok && dalloc(ctx, e, &e->cq_mask, (size_t)B * KS);
    ok = ok && dalloc(ctx, e, &e->cq_n, (size_t)B);
    ok = ok && dalloc(ctx, e, &e->cq_flags, (size_t)B);
    ok = ok && dalloc(ctx, e, &e->kab_src, (size_t)B * KS * 2);
    ok = ok && dalloc(ctx, e, &e->kab_tgt, (size_t)B * KS * 2);
    ok = ok && dalloc(ctx, e, &e->kab_out, (size_t)B * 6);
    ok = ok && dalloc(ctx, e, &e->in_n, (size_t)B);
    ok = ok && dalloc(ctx, e, &e->kf_pose, (size_t)B * 3);
    ok = ok && dalloc(ctx, e, &e->kf_vel, (size_t)B * 3) && dalloc(ctx, e, &e->kf_scan, (size_t)B) && dalloc(ctx, e, &e->kf_fresh, (size_t)B) &&
         dalloc(ctx, e, &e->kf_live, (size_t)B) && dalloc(ctx, e, &e->map_n, (size_t)B);
    ok = ok && dalloc(ctx, e, &e->kf_und, (size_t)B * KS * 2);
    ok = ok && dalloc(ctx, e, &e->kf_und_tmp, (size_t)B * KS * 2);
    ok = ok && dalloc(ctx, e, &e->pose, (size_t)B * 3);
    ok = ok && dalloc(ctx, e, &e->vel, (size_t)B * 3);
    ok = ok && dalloc(ctx, e, &e->T_wj0, (size_t)B * 9);
    ok = ok && dalloc(ctx, e, &e->T_init, (size_t)B * 9);
    ok = ok && dalloc(ctx, e, &e->p_w, (size_t)B * KS * 2);
    ok = ok && dalloc(ctx, e, &e->p_jt, (size_t)B * KS * 2);
    ok = ok && dalloc(ctx, e, &e->lm_work, (size_t)B * ((size_t)(2 * KS + 3) * 9 + KS));
    ok = ok && dalloc(ctx, e, &e->lm_out, (size_t)B * 6);
    ok = ok && dalloc(ctx, e, &e->lm_nfev, (size_t)B);
    ok = ok && dalloc(ctx, e, &e->lm_big, (size_t)2 * (1 + B));
    ok = ok && dalloc(ctx, e, &e->lm_info, (size_t)B);
    ok = ok && dalloc(ctx, e, &e->results, (size_t)B * RES_RING);
    if (ok && hipHostMalloc(reinterpret_cast<void **>(&e->results_host), sizeof(roam_lane_result) * (size_t)B * RES_RING, hipHostMallocDefault) != hipSuccess) {
        ROAM_SET_ERR(ctx, "engine: hipHostMalloc failed"); ok = false;
    }
    { const char *se = getenv("ROAM_STAGE_EVENTS"); e->stage_ev = !(se && se[0] == '0'); e->stage_ev_forced = se && (se[0] == '0' || se[0] == '1'); }
    e->rt_on = cfg->retrack_on_device != 0;
    if (ok && e->rt_on) {
        // device-side feature (re)detection: DEFAULT_FEATURE_PARAMS of getFeatures.py:13-18, i.e. sigma = linspace(0.01, 10, 3)
        RtArgs &r = e->rt;
        memset(&r, 0, sizeof(r));
        const double step = (10.0 - 0.01) / 2.0;                         // numpy.linspace: arange(num) * step + start, last = stop
        r.sigma1 = 1.0 * step + 0.01; r.sigma2 = 10.0; r.threshold = 0.0005;
        r.size1 = (int)(3 * r.sigma1); r.size2 = (int)(3 * r.sigma2);
        r.W = e->W; r.rows = cfg->rows; r.cols = cfg->clip; r.stride = cfg->stride; r.payload_off = cfg->payload_off;
        r.rec_bytes = (int64_t)e->rec_bytes;
        const int R = r.slots = std::max(1, std::min(B, cfg->retrack_slots > 0 ? cfg->retrack_slots : 2048));      // (round 6: 512 -> 2048, +2.5 % on the default workload)
        const size_t npx = (size_t)e->W * e->W;
        if (e->W > 2048) { ROAM_SET_ERR(ctx, "engine: device retrack needs a Cartesian image of at most 2048 x 2048"); roam_engine_destroy(ctx); return ROAM_E_ARG; }
        ok = ok && dalloc(ctx, e, &r.rt_n, 2) && dalloc(ctx, e, &r.rt_lane, (size_t)B) && dalloc(ctx, e, &r.rt_scan, (size_t)B);
        // rows of the integral image start on 128-byte lines: a wave's 512-byte store then fills four whole lines (with the natural
        // pitch of 2024 doubles HBM saw 1.44 x the bytes written)
        r.SP = (e->W + 15) & ~15;
        ok = ok && dalloc(ctx, e, &r.S, (size_t)r.SP * e->W * R);
        ok = ok && dalloc(ctx, e, const_cast<uint32_t **>(&r.boxtab), retrack_boxtab_words(e->W));
        ok = ok && dalloc(ctx, e, const_cast<uint32_t **>(&r.darktab), retrack_darktab_words(e->W));
        ok = ok && dalloc(ctx, e, const_cast<uint32_t **>(&r.phlist), retrack_phase_words(e->W));
        ok = ok && dalloc(ctx, e, &r.colT, (size_t)std::min(R, RT_TWO_PASS_SLOTS) * ((e->W + 63) / 64) * e->W);
        ok = ok && dalloc(ctx, e, &r.col_done, (size_t)std::min(R, RT_TWO_PASS_SLOTS) * 64);
        // the fused detection kernel (retrack_fused.inc: integral image + determinants + maxima without the float64 image in HBM) serves
        // chunks of >= RT_TWO_PASS_SLOTS detections when ROAM_FUSED_DETECT=1 asks for it.  It is bit-identical to the two-kernel form
        // (tests/test_gpu_fused_detect.py) and MEASURED SLOWER - 27 ms against 12.6 ms per 512 detections, DESIGN.md section 6e: the path is
        // bound by instruction issue, not by HBM - so the two-kernel form stays the default.  Its tables: transposed sampling map,
        // footprints and dark steps per band; per slot: two hand-off buffers + the column totals
        {
            // (its buffers - 1 MB of hand-off scratch per slot, 0.55 GB at 512 slots, and three tables - exist only when the variant is
            // asked for: by the environment here, by roam_engine_debug_detect / roam_engine_time_kernel("doh_fused") on first use)
            const char *fv = getenv("ROAM_FUSED_DETECT");
            r.fused = (fv && fv[0] == '1') ? 1 : 0;
            // Where the pyramid of the step after next runs (round 6).  Left alone it starts when its warp ends - beside the integral images
            // of this step, the one pairing on this path that is WORSE than running the two one after the other (integral 31 ms + pyramid
            // 15.5 ms in-step against 25 + 6.3 alone).  2: it waits for this step's determinants (1: for the integral images) and runs
            // beside the next step's back end: +2 % on the default workload.  Batches only - a single sequence lives on the overlap of
            // its front end with the previous pair's back end.  ROAM_PYR_AFTER_INTEGRAL = 0 / 1 / 2 overrides.
            const char *pv = getenv("ROAM_PYR_AFTER_INTEGRAL");
            e->pyr_after_int = pv ? atoi(pv) : (B >= 256 ? 2 : 0);
            // the polar peaks of the step after next (needed by that step's keyframe glue only) wait for the same event: they run beside the
            // one-wavefront-per-detection bookkeeping that ends this step instead of beside its back end: +2 % more (50.8 -> 51.8 k)
            const char *kv = getenv("ROAM_PEAKS_AFTER_INTEGRAL");
            e->peaks_after_int = kv ? atoi(kv) : (B >= 256 ? 1 : 0);
            const char *wv = getenv("ROAM_WARP_AFTER_INTEGRAL");
            e->warp_after_int = wv ? atoi(wv) : 0;
            // ... and not for the determinants' end itself but for the first two kernels of the bookkeeping behind them (candidate order,
            // longest-list-first ordering: 18 us + 0.6 ms alone): launched at the same moment as the pyramid's and the peaks' 130 000
            // workgroups, whichever bookkeeping kernel came first waited ~3.8 ms for slots.  +0.8 % (52.6-52.8 -> 53.1-53.3 k, same box;
            // the pyramid alone behind that event: nothing; the peaks alone: -1 %).  ROAM_PYR_AFTER_EMIT / ROAM_PEAKS_AFTER_EMIT = 0: as before
            e->pyr_after_emit = getenv("ROAM_PYR_AFTER_EMIT") ? atoi(getenv("ROAM_PYR_AFTER_EMIT")) : (B >= 256 ? 1 : 0);
            e->peaks_after_emit = getenv("ROAM_PEAKS_AFTER_EMIT") ? atoi(getenv("ROAM_PEAKS_AFTER_EMIT")) : (B >= 256 ? 1 : 0);
            // (experiment) the warp and the pyramid trade places: the warp of step N + 2 beside step N's bookkeeping, the pyramid of step
            // N + 1 beside step N's back end (after its tracker)
            e->swap_warp_pyr = getenv("ROAM_SWAP_WARP_PYR") ? atoi(getenv("ROAM_SWAP_WARP_PYR")) : 0;
            // the determinants of a chunk of 1 024 detections beside the next chunk's integral images (launch_retrack; +1.7 %); needs both
            // halves of a 2 048-slot scratch.  ROAM_DET_SIDE=0: one launch of each kernel per chunk of `slots`, as until late round 6
            e->det_side.chunk = getenv("ROAM_DET_SIDE") ? atoi(getenv("ROAM_DET_SIDE")) : ((B >= 2048 && r.slots >= 2048) ? 1024 : 0);
            e->lm_side = getenv("ROAM_LM_SIDE") ? atoi(getenv("ROAM_LM_SIDE")) : 0;      // (measured: nothing - 70.68 / 70.46 against 70.69 / 70.29 ms per step)
            if (e->warp_after_int && !e->pyr_after_int) e->pyr_after_int = 1;     // (the event is made for either)
            r.fd_halo_words = (int64_t)retrack_fused_halo_words(e->W);
        }
        // candidate lists and bookkeeping tables per DETECTION (0.9 MB each): K4-K7 run once per step over all of them
        const size_t D = (size_t)B;
        ok = ok && dalloc(ctx, e, &r.cand_rc, D * BP_MAX_PTS) && dalloc(ctx, e, &r.cand_val, D * BP_MAX_PTS) && dalloc(ctx, e, &r.cand_n, D);
        ok = ok && dalloc(ctx, e, &r.tasks, D * BP_MAX_TASKS) && dalloc(ctx, e, &r.pairs, D * (BP_MAX_PAIRS + 1));
        ok = ok && dalloc(ctx, e, &r.order, D * (BP_MAX_PAIRS + 1)) && dalloc(ctx, e, &r.ovbits, D * ((BP_MAX_PAIRS + 31) / 32 + 1));
        ok = ok && dalloc(ctx, e, &r.bigtab, D * 2 * 131072);
        ok = ok && dalloc(ctx, e, &r.kp, D * BP_MAX_PTS * 3) && dalloc(ctx, e, &r.kp_n, D) && dalloc(ctx, e, &r.slot_flags, D);
        ok = ok && dalloc(ctx, e, &r.ssc_work, D * 4 * BP_MAX_PTS) && dalloc(ctx, e, &r.sel, D * BP_MAX_PTS) && dalloc(ctx, e, &r.sel_n, D) && dalloc(ctx, e, &r.blob_order_buf, D);
    }
    if (!ok) { roam_engine_destroy(ctx); return ROAM_E_HIP; }
    if (e->rt_on) {
        if (hipError_t er = retrack_init(); er != hipSuccess) {
            ROAM_SET_ERR(ctx, "retrack_init failed: %s", hipGetErrorString(er));
            roam_engine_destroy(ctx);
            return ROAM_E_HIP;
        }
        RtArgs &r = e->rt;
        r.pool = e->pool; r.map = e->warp_map; r.feat = e->feat; r.feat_n = e->feat_n; r.vel = e->vel; r.kf_und = e->kf_und; r.res = nullptr;
    }
    for (auto &ev : e->ev_res)
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ROAM_SET_ERR(ctx, "hipEventCreate failed"); roam_engine_destroy(ctx); return ROAM_E_HIP; }
    e->slot_seq.assign((size_t)cfg->pool_scans, 0);
    for (auto &ev : e->ev_up_ring)
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ROAM_SET_ERR(ctx, "hipEventCreate failed"); roam_engine_destroy(ctx); return ROAM_E_HIP; }
    if (hipEventCreateWithFlags(&e->ev_pool, hipEventDisableTiming) != hipSuccess) {
        ROAM_SET_ERR(ctx, "hipEventCreate failed"); roam_engine_destroy(ctx); return ROAM_E_HIP;
    }
    for (auto &ev : e->ev) {
        if (hipEventCreate(&ev) != hipSuccess) { ROAM_SET_ERR(ctx, "hipEventCreate failed"); roam_engine_destroy(ctx); return ROAM_E_HIP; }
    }
    if (hipEventCreate(&e->ev_join) != hipSuccess ||
        hipEventCreate(&e->ev_pk0) != hipSuccess || hipEventCreate(&e->ev_pk1) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_warp, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e->ev_idx, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_peaks, hipEventDisableTiming) != hipSuccess) {
        ROAM_SET_ERR(ctx, "hipEventCreate failed"); roam_engine_destroy(ctx); return ROAM_E_HIP;
    }
    for (int i = 0; i < 4; i++)
        if (hipEventCreateWithFlags(&e->ev_klt[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e->ev_g4[i], hipEventDisableTiming) != hipSuccess) {
            ROAM_SET_ERR(ctx, "hipEventCreate failed"); roam_engine_destroy(ctx); return ROAM_E_HIP;
        }
    e->ev_ok = true;
    // (the 64 x (6 + 3 x RT_TRACE_CHUNKS) timestamp events of the per-step traces are created by the first step that records into a slot:
    // creating all of them here was ~5 ms of the 13.6 ms a single sequence's engine cost to set up and tear down, recorded or not)
    e->tr_ok = true;
    HIP_TRY(ctx, launch_warp_map(ctx->stream, cfg->rows, cfg->clip, e->warp_map));
    HIP_TRY(ctx, launch_pyr_dark(ctx->stream, e->warp_map, e->W, e->W, cfg->clip, e->pyr_dark));
    if (e->rt_on) HIP_TRY(ctx, launch_retrack_boxtab(ctx->stream, e->warp_map, e->W, cfg->clip, const_cast<uint32_t *>(e->rt.boxtab)));
    if (e->rt_on) HIP_TRY(ctx, launch_retrack_darktab(ctx->stream, e->warp_map, e->W, cfg->clip, const_cast<uint32_t *>(e->rt.darktab)));
    if (e->rt_on && e->rt.fused) { const int32_t rc_ = fused_tables(ctx, e); if (rc_ != ROAM_OK) { roam_engine_destroy(ctx); return rc_; } }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    e->rt_image_px = (int64_t)e->W * e->W;
    if (e->rt_on && e->W <= 2048 && B >= RT_TWO_PASS_SLOTS) {
        e->rt_image_px = 0;
        // (an engine of fewer lanes never launches the one-sweep kernel: a single sequence's engine skips these ~5 ms of its set-up)
        // the phases the one-sweep integral kernel walks: from the sampling map and the dark-step table just made (host code, once)
        // (through PINNED staging: a pageable hipMemcpy of this size leaves the runtime in a state in which the small device-to-device
        // copy of the keyframe exchange costs 80 us more per step - measured, round 6)
        const size_t npx = (size_t)e->W * e->W, ndt = retrack_darktab_words(e->W), nph = retrack_phase_words(e->W);
        uint32_t *stage = nullptr;
        HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&stage), (npx + ndt + nph) * 4, hipHostMallocDefault));
        uint32_t *mh = stage, *dh = stage + npx;
        std::vector<uint32_t> ph(nph, 0u);
        hipError_t ec = hipMemcpyAsync(mh, e->warp_map, npx * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (ec == hipSuccess) ec = hipMemcpyAsync(dh, e->rt.darktab, ndt * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (ec == hipSuccess) ec = hipStreamSynchronize(ctx->stream);
        bool okp = ec == hipSuccess && retrack_build_phases(mh, dh, e->W, cfg->clip, ph.data());
        if (okp) {
            memcpy(stage + npx + ndt, ph.data(), nph * 4);
            ec = hipMemcpyAsync(const_cast<uint32_t *>(e->rt.phlist), stage + npx + ndt, nph * 4, hipMemcpyHostToDevice, ctx->stream);
            if (ec == hipSuccess) ec = hipStreamSynchronize(ctx->stream);
        }
        (void)hipHostFree(stage);
        if (!okp || ec != hipSuccess) { ROAM_SET_ERR(ctx, "retrack: phase list (%s)", ec != hipSuccess ? hipGetErrorString(ec) : "image too large"); roam_engine_destroy(ctx); return ec != hipSuccess ? ROAM_E_HIP : ROAM_E_ARG; }
        for (uint32_t i = 0; i < ph[0]; i++) {
            const int br = retrack_band_rows(), band = (int)(ph[1 + i] & 255u), g = (int)((ph[1 + i] >> 8) & 15u), hrows = std::min(br, e->W - br * band);
            for (int w = 0; w < 4; w++)
                if ((ph[1 + i] >> (12 + w)) & 1u) e->rt_image_px += (int64_t)hrows * std::max(0, std::min(64, e->W - (g * 256 + 64 * w)));
        }
    }
    return ROAM_OK;
}

#define ENGINE()                                  \
    if (!ctx) return ROAM_E_ARG;                  \
    Engine *e = ctx->engine;                      \
    if (!e) { ROAM_SET_ERR(ctx, "engine not created"); return ROAM_E_STATE; } \
    HIP_TRY(ctx, hipSetDevice(ctx->device))

int32_t roam_engine_upload_scan(roam_ctx *ctx, int32_t pool_idx, const uint8_t *rec)
{
    ENGINE();
    ARG_CHECK(ctx, rec && pool_idx >= 0 && pool_idx < e->cfg.pool_scans);
    HIP_TRY(ctx, hipMemcpyAsync(e->pool + (size_t)pool_idx * e->rec_bytes, rec, e->rec_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ROAM_OK;
}

// f2 (raw-record ingest): records stream from PINNED host memory on the copy stream while the main
// stream computes.  Protocol for a double-buffered pool (halves A/B):
//   roam_engine_upload_scans_async(half B)   - copy stream; starts after the last roam_engine_fence()
//   roam_engine_step(scans of half A)        - main stream; first waits for the uploads of the slots it reads
//   roam_engine_fence()                      - uploads enqueued from now on wait for the steps enqueued so far
int32_t roam_engine_upload_scans_async(roam_ctx *ctx, int32_t pool_idx0, int32_t n, const uint8_t *host_records, int64_t host_stride)
{
    ENGINE();
    ARG_CHECK(ctx, host_records && n >= 1 && pool_idx0 >= 0 && pool_idx0 + n <= e->cfg.pool_scans && (host_stride == 0 || host_stride >= (int64_t)e->rec_bytes));
    // the copy kernel dereferences the records on the GPU: they must live in pinned (or registered) host memory - a pageable
    // buffer would fault on the device and abort the process, so it is refused here
    hipPointerAttribute_t at;
    const hipError_t pa = hipPointerGetAttributes(&at, host_records);
    if (pa != hipSuccess || (at.type != hipMemoryTypeHost && at.type != hipMemoryTypeManaged && at.type != hipMemoryTypeDevice)) {
        (void)hipGetLastError();
        ROAM_SET_ERR(ctx, "upload_scans_async: host_records is not pinned host memory (hipHostMalloc / hipHostRegister / roam_host_alloc)");
        return ROAM_E_ARG;
    }
    const uint8_t *dev_alias = at.devicePointer ? static_cast<const uint8_t *>(at.devicePointer) : host_records;
    // only the bytes the path reads cross PCIe: metadata + the clipped payload of every row (2 036 of 3 779 bytes for the
    // Oxford record at the 87.5 m clip).  A copy KERNEL on the copy stream reads the pinned host memory directly (it is
    // device-visible) with 16-byte loads - hipMemcpy2DAsync moves such rows one by one (270 scan pairs/s measured)
    const int width = e->cfg.payload_off + e->cfg.clip;
    hipLaunchKernelGGL(ingest_rows_kernel, dim3((e->cfg.rows + 3) / 4, n), dim3(256), 0, ctx->stream3, dev_alias, host_stride,
                       e->pool + (size_t)pool_idx0 * e->rec_bytes, (int64_t)e->rec_bytes, e->cfg.rows, e->cfg.stride, width);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_up, ctx->stream3));
    e->up_seq++;
    for (int i = 0; i < n; i++) e->slot_seq[(size_t)pool_idx0 + i] = e->up_seq;
    HIP_TRY(ctx, hipEventRecord(e->ev_up_ring[e->up_seq & 15], ctx->stream3));
    e->uploads_pending = true;
    return ROAM_OK;
}

int32_t roam_engine_fence(roam_ctx *ctx)
{
    ENGINE();
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fence, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream3, ctx->ev_fence, 0));
    return ROAM_OK;
}

int32_t roam_engine_copy_scan(roam_ctx *ctx, int32_t dst_idx, int32_t src_idx)
{
    ENGINE();
    ARG_CHECK(ctx, dst_idx >= 0 && dst_idx < e->cfg.pool_scans && src_idx >= 0 && src_idx < e->cfg.pool_scans);
    if (dst_idx != src_idx)
    {
        HIP_TRY(ctx, hipMemcpyAsync(e->pool + (size_t)dst_idx * e->rec_bytes, e->pool + (size_t)src_idx * e->rec_bytes,
                                    e->rec_bytes, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(e->ev_pool, ctx->stream));         // the front-end streams of the next step wait for it
        e->pool_dirty = true;
    }
    return ROAM_OK;
}

static WarpSrc pool_warp_src(Engine *e, const int32_t *lane_index)
{
    WarpSrc s = {e->pool, (int64_t)e->rec_bytes, (int64_t)e->cfg.stride, e->cfg.payload_off, 1, lane_index};
    return s;
}

static int32_t set_features_impl(roam_ctx *ctx, Engine *e, int32_t lane, const float *pts, int32_t K, int32_t kf_scan)
{
    float *f = e->feat + (size_t)lane * KS * 2;
    // f1: the keyframe this call replaces goes to the lane's map ring first (device-side decision, see the kernel)
    if (e->map_cap > 0) {
        hipLaunchKernelGGL(map_freeze_kernel, dim3(1), dim3(256), 0, ctx->stream, e->map_store, e->map_n, e->map_cap, lane,
                           e->kf_pose, e->kf_vel, e->feat_n, e->kf_scan, e->kf_und, e->kf_live, e->kf_fresh);
        HIP_TRY(ctx, hipGetLastError());
    }
    e->lane_k[lane] = K;
    if (K > 0) HIP_TRY(ctx, hipMemcpyAsync(f, pts, sizeof(float) * 2 * (size_t)K, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(e->feat_n + lane, &K, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    // the frame that triggers a retrack also adds a keyframe at the latest pose (RawROAMSystem.py:250-270):
    // kf pose = latest pose, kf locals = undistort(velocity, centred features)
    HIP_TRY(ctx, hipMemcpyAsync(e->kf_pose + 3 * (size_t)lane, e->pose + 3 * (size_t)lane, sizeof(double) * 3,
                                hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(e->kf_vel + 3 * (size_t)lane, e->vel + 3 * (size_t)lane, sizeof(double) * 3,
                                hipMemcpyDeviceToDevice, ctx->stream));
    const int32_t one = 1;
    HIP_TRY(ctx, hipMemcpyAsync(e->kf_live + lane, &one, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if (kf_scan >= 0) HIP_TRY(ctx, hipMemcpyAsync(e->kf_scan + lane, &kf_scan, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if (K > 0) {
        hipLaunchKernelGGL(lane_kf_reset_kernel, dim3((K + 255) / 256), dim3(256), 0, ctx->stream, f, K,
                           e->vel + 3 * (size_t)lane, e->kf_und + (size_t)lane * KS * 2);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ROAM_OK;
}

int32_t roam_engine_set_features(roam_ctx *ctx, int32_t lane, const float *pts, int32_t K)
{
    ENGINE();
    ARG_CHECK(ctx, lane >= 0 && lane < e->B && K >= 0 && K <= KS && (K == 0 || pts));
    return set_features_impl(ctx, e, lane, pts, K, e->last_scan[lane]);
}

int32_t roam_engine_map_reserve(roam_ctx *ctx, int32_t keyframes_per_lane)
{
    ENGINE();
    ARG_CHECK(ctx, keyframes_per_lane > 0 && keyframes_per_lane <= 4096);
    if (e->map_cap > 0) { ROAM_SET_ERR(ctx, "engine: the keyframe map is already reserved (%d per lane)", e->map_cap); return ROAM_E_STATE; }
    if (!dalloc(ctx, e, &e->map_store, (size_t)e->B * keyframes_per_lane * MAP_SLOT)) return ROAM_E_HIP;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    e->map_cap = keyframes_per_lane;
    return ROAM_OK;
}

int32_t roam_engine_map_count(roam_ctx *ctx, int32_t lane, int32_t *count)
{
    ENGINE();
    ARG_CHECK(ctx, lane >= 0 && lane < e->B && count);
    int32_t v[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(&v[0], e->map_n + lane, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&v[1], e->kf_live + lane, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *count = v[0] + (v[1] ? 1 : 0);                 // frozen keyframes + the live one
    return ROAM_OK;
}

int32_t roam_engine_map_get(roam_ctx *ctx, int32_t lane, int32_t index, double *pose3, double *vel3, double *locals_xy,
                            int32_t cap_pts, int32_t *n_out, int32_t *scan_out)
{
    ENGINE();
    ARG_CHECK(ctx, lane >= 0 && lane < e->B && index >= 0 && pose3 && vel3 && n_out && scan_out && cap_pts >= 0 && (cap_pts == 0 || locals_xy));
    int32_t v[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(&v[0], e->map_n + lane, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&v[1], e->kf_live + lane, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int frozen = v[0];
    if (index < frozen) {
        const double *src = e->map_store + ((size_t)lane * e->map_cap + index) * MAP_SLOT;
        double hdr[8];
        HIP_TRY(ctx, hipMemcpyAsync(hdr, src, sizeof(hdr), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const int n = (int)hdr[6];
        for (int i = 0; i < 3; i++) { pose3[i] = hdr[i]; vel3[i] = hdr[3 + i]; }
        *n_out = n; *scan_out = (int32_t)hdr[7];
        if (n > cap_pts) { ROAM_SET_ERR(ctx, "map_get: keyframe has %d points, capacity %d", n, cap_pts); return ROAM_E_CAPACITY; }
        if (n > 0) HIP_TRY(ctx, hipMemcpyAsync(locals_xy, src + 8, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    } else if (index == frozen && v[1]) {           // the live keyframe
        int32_t n = 0, sc = -1;
        HIP_TRY(ctx, hipMemcpyAsync(pose3, e->kf_pose + 3 * (size_t)lane, sizeof(double) * 3, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(vel3, e->kf_vel + 3 * (size_t)lane, sizeof(double) * 3, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&n, e->feat_n + lane, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&sc, e->kf_scan + lane, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        *n_out = n; *scan_out = sc;
        if (n > cap_pts) { ROAM_SET_ERR(ctx, "map_get: keyframe has %d points, capacity %d", n, cap_pts); return ROAM_E_CAPACITY; }
        if (n > 0) HIP_TRY(ctx, hipMemcpyAsync(locals_xy, e->kf_und + (size_t)lane * KS * 2, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        ROAM_SET_ERR(ctx, "map_get: lane %d holds %d keyframes, index %d", lane, frozen + (v[1] ? 1 : 0), index);
        return ROAM_E_ARG;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ROAM_OK;
}

int32_t roam_engine_init_lane(roam_ctx *ctx, int32_t lane, int32_t pool_idx, const float *pts, int32_t K,
                              const double *pose3)
{
    ENGINE();
    ARG_CHECK(ctx, lane >= 0 && lane < e->B && pool_idx >= 0 && pool_idx < e->cfg.pool_scans && pose3 && K >= 0 && K <= KS);
    if (e->uploads_pending) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_up, 0));     // asynchronous uploads land first
    // previous-image pyramid of this lane from the pool scan
    uint8_t *pyr = e->pyr[e->cur] + (size_t)lane * e->pd.lane_stride;
    WarpSrc ws = {e->pool + (size_t)pool_idx * e->rec_bytes, 0, (int64_t)e->cfg.stride, e->cfg.payload_off, 1, nullptr};
    HIP_TRY(ctx, launch_warp_gather(ctx->stream, e->warp_map, ws, 1, e->cfg.rows, e->cfg.clip, pyr, e->pd.lane_stride, e->warp_dark_zero));
    HIP_TRY(ctx, launch_build_pyramid(ctx->stream, pyr, e->pd, 1, e->pyr_dark));
    double zero[3] = {0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(e->pose + 3 * (size_t)lane, pose3, sizeof(double) * 3, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(e->vel + 3 * (size_t)lane, zero, sizeof(double) * 3, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return set_features_impl(ctx, e, lane, pts, K, pool_idx);
}

int32_t roam_engine_init_lane_detect(roam_ctx *ctx, int32_t lane, int32_t pool_idx, const double *pose3)
{
    ENGINE();
    ARG_CHECK(ctx, lane >= 0 && lane < e->B && pool_idx >= 0 && pool_idx < e->cfg.pool_scans && pose3);
    if (!e->rt_on) { ROAM_SET_ERR(ctx, "engine created without retrack_on_device"); return ROAM_E_STATE; }
    int32_t rc = roam_engine_init_lane(ctx, lane, pool_idx, nullptr, 0, pose3);      // pyramid, pose, zero velocity, empty keyframe
    if (rc != ROAM_OK) return rc;
    // first-frame appendNewFeatures(prevImgCart, empty) (RawROAMSystem.py:150): the retrack path for this one lane
    const int32_t one = 1;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(e->rt.rt_n, &one, sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(e->rt.rt_lane, &lane, sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(e->rt.rt_scan, &pool_idx, sizeof(int32_t), hipMemcpyHostToDevice, st));
    e->rt.res = nullptr;
    HIP_TRY(ctx, launch_retrack(st, e->rt, 1));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    e->lane_k[lane] = 320;
    return ROAM_OK;
}

int32_t roam_bcast_keyframe(roam_ctx *ctx, int32_t root, int32_t lane, roam_keyframe_hdr *hdr_out, double *locals_xy,
                            int32_t cap_pts, int32_t *peaks, int64_t peaks_cap)
{
    ENGINE();
    // (every rank validates the lane BEFORE the collective: a root that bailed out alone would leave the others in ncclBroadcast)
    ARG_CHECK(ctx, hdr_out && lane >= 0 && lane < e->B && cap_pts >= 0 && peaks_cap >= 0);
    static_assert(sizeof(roam_keyframe_hdr) == KFB_HDR, "header layout");
    const size_t total = KFB_PEAKS_OFF + (size_t)e->cfg.peaks_cap * 8;
    if (!e->kfb && !dalloc(ctx, e, &e->kfb, total)) return ROAM_E_HIP;
    hipStream_t st = ctx->stream;
    if (roam_comm_rank(ctx) == root) {
        hipLaunchKernelGGL(kf_pack_kernel, dim3(16), dim3(256), 0, st, e->kfb, lane, e->kf_pose, e->kf_vel, e->feat_n, e->kf_scan,
                           e->kf_und, e->peaks_n[e->pk], e->peaks_out[e->pk], e->cfg.peaks_cap);
        HIP_TRY(ctx, hipGetLastError());
    }
    // header + features (fixed 16.4 KB), then the peak list sized by the header: two latency-bound broadcasts
    int32_t rc = roam_comm_bcast_bytes(ctx, e->kfb, KFB_PEAKS_OFF, root);
    if (rc != ROAM_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(hdr_out, e->kfb, sizeof(roam_keyframe_hdr), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const int n = hdr_out->n_features, P = hdr_out->n_peaks;
    if (n < 0 || n > KS || P < 0 || P > e->cfg.peaks_cap) { ROAM_SET_ERR(ctx, "bcast_keyframe: corrupt header (n=%d, P=%d)", n, P); return ROAM_E_STATE; }
    if (P > 0) {
        rc = roam_comm_bcast_bytes(ctx, e->kfb + KFB_PEAKS_OFF, (size_t)P * 8, root);
        if (rc != ROAM_OK) return rc;
    }
    if (e->rmap_cap > 0) {
        // (refused BEFORE anything is written: the exchange owns the ring slots and their count once it has run)
        if (e->kfx_calls) { ROAM_SET_ERR(ctx, "remote map: roam_bcast_keyframe and roam_keyframe_exchange were mixed on one engine"); return ROAM_E_STATE; }
        // Map.addKeyframe on this rank: the payload stays in HBM (device-to-device), header + features, then the peaks
        const size_t slot_bytes = KFB_PEAKS_OFF + (size_t)e->cfg.peaks_cap * 8;
        uint8_t *dst = e->rmap + (size_t)(e->rmap_n % e->rmap_cap) * slot_bytes;
        HIP_TRY(ctx, hipMemcpyAsync(dst, e->kfb, KFB_HDR + sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToDevice, st));
        if (P > 0) HIP_TRY(ctx, hipMemcpyAsync(dst + KFB_PEAKS_OFF, e->kfb + KFB_PEAKS_OFF, (size_t)P * 8, hipMemcpyDeviceToDevice, st));
        e->rmap_root[e->rmap_n % e->rmap_cap] = root;
        e->rmap_n++;
        e->rmap_n_bcast++;
    }
    if (locals_xy) {
        if (n > cap_pts) { ROAM_SET_ERR(ctx, "bcast_keyframe: %d features, capacity %d", n, cap_pts); return ROAM_E_CAPACITY; }
        if (n > 0) HIP_TRY(ctx, hipMemcpyAsync(locals_xy, e->kfb + KFB_LOCALS_OFF, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost, st));
    }
    if (peaks) {
        if (P > peaks_cap) { ROAM_SET_ERR(ctx, "bcast_keyframe: %d peaks, capacity %lld", P, (long long)peaks_cap); return ROAM_E_CAPACITY; }
        if (P > 0) HIP_TRY(ctx, hipMemcpyAsync(peaks, e->kfb + KFB_PEAKS_OFF, sizeof(int32_t) * 2 * (size_t)P, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return ROAM_OK;
}

// the exchange's resources, made once for `world` ranks (a failed set-up is retried, never half used)
static int32_t kfx_setup(roam_ctx *ctx, Engine *e, int world)
{
    if (e->kfx_ready) {
        if (world > e->kfx_world) { ROAM_SET_ERR(ctx, "keyframe exchange: set up for %d ranks, asked for %d", e->kfx_world, world); return ROAM_E_STATE; }
        return ROAM_OK;
    }
    e->kfx_peaks = std::min(e->cfg.peaks_cap, 32768);
    e->kfx_rec = KFB_PEAKS_OFF + (size_t)e->kfx_peaks * 8;
    if (!e->st_comm) HIP_TRY(ctx, hipStreamCreateWithFlags(&e->st_comm, hipStreamNonBlocking));
    for (auto &ev : e->ev_kfx) if (!ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (!e->kfx_send && !dalloc(ctx, e, &e->kfx_send, e->kfx_rec)) return ROAM_E_HIP;
    if (!e->kfx_recv && !dalloc(ctx, e, &e->kfx_recv, e->kfx_rec * (size_t)world)) return ROAM_E_HIP;
    if (!e->rmap_n_dev && !dalloc(ctx, e, &e->rmap_n_dev, 1)) return ROAM_E_HIP;
    if (!e->rmap_root_dev && !dalloc(ctx, e, &e->rmap_root_dev, (size_t)e->rmap_cap)) return ROAM_E_HIP;
    HIP_TRY(ctx, hipMemsetAsync(e->rmap_n_dev, 0, sizeof(int64_t), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    e->kfx_world = world;
    e->kfx_ready = true;
    return ROAM_OK;
}

// test / debug: the RECEIVE half of roam_keyframe_exchange on one GPU.  `recv` (host) holds `world` records as the ncclAllGather of a
// `world`-rank job leaves them in every rank's receive buffer - record r = rank r's, *rec_bytes apart (header at 0, locals at
// *locals_off, peaks at *peaks_off) - and goes through the SAME kfx_append_kernel, on the exchange stream, into this rank's remote map.
// With recv == NULL only the layout is returned.  No communicator is needed; the map must be reserved with at least `world` slots.
int32_t roam_debug_keyframe_append(roam_ctx *ctx, const uint8_t *recv, int32_t world, int64_t *rec_bytes, int32_t *locals_off,
                                   int32_t *peaks_off, int32_t *max_peaks)
{
    ENGINE();
    ARG_CHECK(ctx, world >= 1 && world <= 4096);
    if (e->rmap_cap <= 0) { ROAM_SET_ERR(ctx, "debug_keyframe_append: reserve the remote map first"); return ROAM_E_STATE; }
    if (e->rmap_cap < world) { ROAM_SET_ERR(ctx, "debug_keyframe_append: remote map of %d slots for %d ranks", e->rmap_cap, world); return ROAM_E_CAPACITY; }
    if (e->rmap_n_bcast) { ROAM_SET_ERR(ctx, "remote map: roam_bcast_keyframe and roam_keyframe_exchange were mixed on one engine"); return ROAM_E_STATE; }
    { const int32_t rc_ = kfx_setup(ctx, e, world); if (rc_ != ROAM_OK) return rc_; }
    if (rec_bytes) *rec_bytes = (int64_t)e->kfx_rec;
    if (locals_off) *locals_off = KFB_LOCALS_OFF;
    if (peaks_off) *peaks_off = KFB_PEAKS_OFF;
    if (max_peaks) *max_peaks = e->kfx_peaks;
    if (!recv) return ROAM_OK;
    hipStream_t st = e->st_comm;
    HIP_TRY(ctx, hipMemcpyAsync(e->kfx_recv, recv, e->kfx_rec * (size_t)world, hipMemcpyHostToDevice, st));
    const size_t slot_bytes = KFB_PEAKS_OFF + (size_t)e->cfg.peaks_cap * 8;
    hipLaunchKernelGGL(kfx_append_kernel, dim3(1), dim3(256), 0, st, e->kfx_recv, world, e->kfx_rec, e->rmap, slot_bytes, e->rmap_cap,
                       e->rmap_n_dev, e->rmap_root_dev, KS, e->kfx_peaks);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));                            // (the host buffer is the caller's again)
    e->kfx_calls++;
    return ROAM_OK;
}

// collective, NON-BLOCKING: every rank calls it once after each roam_engine_step with its own lane.  The rank's record (the lane's
// new keyframe if the step made one, else empty) is packed on the device, one ncclAllGather of the fixed-size records runs on the
// exchange stream behind the step's last kernel, and every non-empty record is appended to this rank's remote map on the device.
// No host synchronisation: the step pipeline keeps running; roam_remote_map_count / _get wait for the exchange stream.
int32_t roam_keyframe_exchange(roam_ctx *ctx, int32_t lane)
{
    ENGINE();
    ARG_CHECK(ctx, lane >= 0 && lane < e->B);
    if (!ctx->comm) { ROAM_SET_ERR(ctx, "keyframe_exchange: communicator not initialised"); return ROAM_E_STATE; }
    if (e->rmap_cap <= 0) { ROAM_SET_ERR(ctx, "keyframe_exchange: reserve the remote map first"); return ROAM_E_STATE; }
    if (e->nstep == 0) { ROAM_SET_ERR(ctx, "keyframe_exchange: no step enqueued"); return ROAM_E_STATE; }
    const int world = roam_comm_world(ctx);
    // (the append kernel writes the records of one gather into consecutive ring slots: fewer slots than ranks would tear them)
    if (e->rmap_cap < world) { ROAM_SET_ERR(ctx, "keyframe_exchange: remote map of %d slots for %d ranks", e->rmap_cap, world); return ROAM_E_CAPACITY; }
    if (e->rmap_n_bcast) { ROAM_SET_ERR(ctx, "remote map: roam_bcast_keyframe and roam_keyframe_exchange were mixed on one engine"); return ROAM_E_STATE; }
    { const int32_t rc_ = kfx_setup(ctx, e, world); if (rc_ != ROAM_OK) return rc_; }
    hipStream_t st = e->st_comm;
    const int rs = (int)((e->nstep - 1) % RES_RING);
    HIP_TRY(ctx, hipStreamWaitEvent(st, e->ev_res[rs], 0));            // the step's records (and with them its keyframe state) are final
    hipLaunchKernelGGL(kfx_pack_kernel, dim3(16), dim3(256), 0, st, e->kfx_send, lane, e->results + (size_t)rs * e->B, e->kf_pose, e->kf_vel,
                       e->feat_n, e->kf_scan, e->kf_und, e->peaks_n[e->pk], e->peaks_out[e->pk], e->cfg.peaks_cap, e->kfx_peaks);
    HIP_TRY(ctx, hipGetLastError());
    // the following steps overwrite what the pack kernel reads (keyframe state: the compute stream; the peak list of this ring slot:
    // the peak stream, three steps on): their streams wait for the PACK, not for the collective
    HIP_TRY(ctx, hipEventRecord(e->ev_kfx[e->pk], st));
    e->kfx_last = e->pk;
    int32_t rc = roam_comm_allgather_bytes(ctx, e->kfx_send, e->kfx_recv, e->kfx_rec, st);
    if (rc != ROAM_OK) return rc;
    const size_t slot_bytes = KFB_PEAKS_OFF + (size_t)e->cfg.peaks_cap * 8;
    hipLaunchKernelGGL(kfx_append_kernel, dim3(1), dim3(256), 0, st, e->kfx_recv, world, e->kfx_rec, e->rmap, slot_bytes, e->rmap_cap,
                       e->rmap_n_dev, e->rmap_root_dev, KS, e->kfx_peaks);
    HIP_TRY(ctx, hipGetLastError());
    e->kfx_calls++;
    return ROAM_OK;
}

// the exchange stream has drained: pull the device-side count and senders over to the host's bookkeeping
static int32_t kfx_settle(roam_ctx *ctx, Engine *e)
{
    if (!e->kfx_ready) return ROAM_OK;
    HIP_TRY(ctx, hipStreamSynchronize(e->st_comm));
    int64_t n = 0;
    HIP_TRY(ctx, hipMemcpy(&n, e->rmap_n_dev, sizeof(n), hipMemcpyDeviceToHost));
    // (the map may be polled while the loop runs: the device-side count only grows; the two producers refuse each other at call time)
    if (n > 0) {
        e->rmap_n = n;
        HIP_TRY(ctx, hipMemcpy(e->rmap_root.data(), e->rmap_root_dev, sizeof(int32_t) * (size_t)e->rmap_cap, hipMemcpyDeviceToHost));
    }
    return ROAM_OK;
}

int32_t roam_remote_map_reserve(roam_ctx *ctx, int32_t keyframes)
{
    ENGINE();
    ARG_CHECK(ctx, keyframes > 0 && keyframes <= 65536);
    if (e->rmap_cap > 0) { ROAM_SET_ERR(ctx, "engine: the remote map is already reserved (%d keyframes)", e->rmap_cap); return ROAM_E_STATE; }
    const size_t slot_bytes = KFB_PEAKS_OFF + (size_t)e->cfg.peaks_cap * 8;
    if (!dalloc(ctx, e, &e->rmap, slot_bytes * (size_t)keyframes)) return ROAM_E_HIP;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    e->rmap_root.assign((size_t)keyframes, -1);
    e->rmap_cap = keyframes;
    return ROAM_OK;
}

int32_t roam_remote_map_count(roam_ctx *ctx, int64_t *received, int32_t *resident)
{
    ENGINE();
    ARG_CHECK(ctx, received && resident);
    { const int32_t rc_ = kfx_settle(ctx, e); if (rc_ != ROAM_OK) return rc_; }
    *received = e->rmap_n;
    *resident = (int32_t)std::min<int64_t>(e->rmap_n, e->rmap_cap);
    return ROAM_OK;
}

int32_t roam_remote_map_get(roam_ctx *ctx, int32_t index, roam_keyframe_hdr *hdr_out, int32_t *root_out, double *locals_xy,
                            int32_t cap_pts, int32_t *peaks, int64_t peaks_cap)
{
    ENGINE();
    { const int32_t rc_ = kfx_settle(ctx, e); if (rc_ != ROAM_OK) return rc_; }
    const int resident = (int)std::min<int64_t>(e->rmap_n, e->rmap_cap);
    ARG_CHECK(ctx, hdr_out && index >= 0 && index < resident && cap_pts >= 0 && peaks_cap >= 0);
    // index 0 = the oldest keyframe still resident
    const int64_t abs_i = e->rmap_n - resident + index;
    const size_t slot_bytes = KFB_PEAKS_OFF + (size_t)e->cfg.peaks_cap * 8;
    const uint8_t *src = e->rmap + (size_t)(abs_i % e->rmap_cap) * slot_bytes;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(hdr_out, src, sizeof(roam_keyframe_hdr), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (root_out) *root_out = e->rmap_root[abs_i % e->rmap_cap];
    const int n = hdr_out->n_features, P = hdr_out->n_peaks;
    if (locals_xy) {
        if (n > cap_pts) { ROAM_SET_ERR(ctx, "remote_map_get: %d features, capacity %d", n, cap_pts); return ROAM_E_CAPACITY; }
        if (n > 0) HIP_TRY(ctx, hipMemcpyAsync(locals_xy, src + KFB_LOCALS_OFF, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost, st));
    }
    if (peaks) {
        if (P > peaks_cap) { ROAM_SET_ERR(ctx, "remote_map_get: %d peaks, capacity %lld", P, (long long)peaks_cap); return ROAM_E_CAPACITY; }
        if (P > 0) HIP_TRY(ctx, hipMemcpyAsync(peaks, src + KFB_PEAKS_OFF, sizeof(int32_t) * 2 * (size_t)P, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return ROAM_OK;
}

// batched roam_engine_init_lane_detect for lanes lane0 .. lane0 + n - 1: one warp / pyramid launch over the n lanes and ONE
// detection pass (chunks of `retrack_slots`) instead of n single-lane passes of 2.4 ms each (4096 lanes: 10 s -> 0.3 s)
int32_t roam_engine_init_lanes_detect(roam_ctx *ctx, int32_t lane0, int32_t n, const int32_t *pool_idx, const double *poses3)
{
    ENGINE();
    ARG_CHECK(ctx, n >= 1 && lane0 >= 0 && lane0 + n <= e->B && pool_idx && poses3);
    if (!e->rt_on) { ROAM_SET_ERR(ctx, "engine created without retrack_on_device"); return ROAM_E_STATE; }
    for (int i = 0; i < n; i++) ARG_CHECK(ctx, pool_idx[i] >= 0 && pool_idx[i] < e->cfg.pool_scans);
    if (e->map_cap > 0) {                                   // the keyframe map freezes the replaced keyframes lane by lane
        for (int i = 0; i < n; i++) {
            const int32_t rc = roam_engine_init_lane_detect(ctx, lane0 + i, pool_idx[i], poses3 + 3 * (size_t)i);
            if (rc != ROAM_OK) return rc;
        }
        return ROAM_OK;
    }
    hipStream_t st = ctx->stream;
    if (e->uploads_pending) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_up, 0));
    std::vector<int32_t> lanes((size_t)n);
    for (int i = 0; i < n; i++) lanes[i] = lane0 + i;
    HIP_TRY(ctx, hipMemcpyAsync(e->rt.rt_n, &n, sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(e->rt.rt_lane, lanes.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(e->rt.rt_scan, pool_idx, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
    // previous-image pyramids from the pool scans (lane i of the launch reads record rt_scan[i])
    uint8_t *pyr = e->pyr[e->cur] + (size_t)lane0 * e->pd.lane_stride;
    HIP_TRY(ctx, launch_warp_gather(st, e->warp_map, pool_warp_src(e, e->rt.rt_scan), n, e->cfg.rows, e->cfg.clip, pyr, e->pd.lane_stride, e->warp_dark_zero));
    HIP_TRY(ctx, launch_build_pyramid(st, pyr, e->pd, n, e->pyr_dark));
    // pose, zero velocity, empty feature set, keyframe at the pose created on the scan (set_features_impl with K = 0)
    HIP_TRY(ctx, hipMemcpyAsync(e->pose + 3 * (size_t)lane0, poses3, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(e->kf_pose + 3 * (size_t)lane0, poses3, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(e->vel + 3 * (size_t)lane0, 0, sizeof(double) * 3 * (size_t)n, st));
    HIP_TRY(ctx, hipMemsetAsync(e->kf_vel + 3 * (size_t)lane0, 0, sizeof(double) * 3 * (size_t)n, st));
    HIP_TRY(ctx, hipMemsetAsync(e->feat_n + lane0, 0, sizeof(int32_t) * (size_t)n, st));
    HIP_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(e->kf_live + lane0), 1, (size_t)n, st));
    HIP_TRY(ctx, hipMemcpyAsync(e->kf_scan + lane0, pool_idx, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
    // first-frame appendNewFeatures(prevImgCart, empty) (RawROAMSystem.py:150) for all n lanes
    e->rt.res = nullptr;
    HIP_TRY(ctx, launch_retrack(st, e->rt, n));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int i = 0; i < n; i++) e->lane_k[lane0 + i] = 320;
    return ROAM_OK;
}

int32_t roam_engine_step(roam_ctx *ctx, const int32_t *scan_idx)
{
    ENGINE();
    ARG_CHECK(ctx, scan_idx);
    const int B = e->B;
    for (int b = 0; b < B; b++) ARG_CHECK(ctx, scan_idx[b] >= 0 && (scan_idx[b] & ~ROAM_STEP_NEW_SEQUENCE) < e->cfg.pool_scans);
    for (int b = 0; b < B; b++) e->last_scan[b] = scan_idx[b] & ~ROAM_STEP_NEW_SEQUENCE;
    bool any_new = false;
    for (int b = 0; b < B; b++) any_new = any_new || (scan_idx[b] & ROAM_STEP_NEW_SEQUENCE);
    if (any_new && !(e->rt_on && e->rt_mode)) { ROAM_SET_ERR(ctx, "ROAM_STEP_NEW_SEQUENCE needs device-side detection (retrack_on_device, mode >= 1)"); return ROAM_E_STATE; }
    hipStream_t st = ctx->stream;
    const roam_engine_cfg &c = e->cfg;
    const int nw = KS / 64;
    const int KM = e->kmax();          // host-known bound: feature counts only shrink between (re)seeds
    // Three-stage pipeline across steps:
    //   stage A (stream2): warp (+ polar peaks on stream5) - depend only on the raw scan; issue-bound (VALU / LDS)
    //   stage B (stream4): pyramid of the warped image - HBM-bound, little arithmetic
    //   stage C (stream):  KLT ... LM, g4              - needs stage B of its own step and stage C of the previous one
    // When steps are enqueued back to back, A(N+2), B(N+1) and C(N) run concurrently: the HBM-bound pyramid and the
    // latency-bound LM solve fill the gaps of the issue-bound kernels.  Four pyramid buffers (previous / current /
    // in stage B / in stage A) and three peak / scan-index buffers keep the stages apart:
    //   A(N) waits for KLT(N-3)  - the pyramid it overwrites was that tracker's "previous" image
    //   A(N) waits for g4(N-3)   - that kernel reads the peak counts / scan indices of the same ring slot
    //   B(N) waits for A(N), KLT(N) waits for B(N)
    hipStream_t sA = ctx->stream2, sB = ctx->stream4;
    if (e->kfx_calls) {
        // a keyframe exchange reads the previous step's keyframe state (the compute stream overwrites it: wait for the latest PACK) and the peak
        // list of ITS ring slot, which the peak stream overwrites three steps later: stream5 waits for the pack that read the slot it is about
        // to fill - not for the latest one, which would take the peak kernel out of the stage overlap
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, e->ev_kfx[e->kfx_last], 0));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream5, e->ev_kfx[(int)(e->nstep % 3)], 0));
    }
    const int rs = (int)(e->nstep % RES_RING);            // ring slot of this step's result records
    if (e->nstep >= RES_RING) HIP_TRY(ctx, hipEventSynchronize(e->ev_res[rs]));   // its previous copy (8 steps ago) has long landed
    roam_lane_result *res_slot = e->results + (size_t)rs * B;
    const int pb = (int)(e->nstep % 3);                 // ring slot of the peak / scan-index buffers
    const int k4 = (int)(e->nstep & 3), w4 = (int)((e->nstep + 1) & 3);   // event slot of this step / of step N-3
    uint8_t *prev = e->pyr[e->cur], *next = e->pyr[(e->cur + 1) & 3];
    HIP_TRY(ctx, hipStreamWaitEvent(sA, e->ev_klt[w4], 0));
    HIP_TRY(ctx, hipStreamWaitEvent(sA, e->ev_g4[w4], 0));
    // (lane initialisation, retracks and synchronous uploads finish on the host before a step is enqueued)
    if (e->uploads_pending) {
        uint64_t need = 0;
        for (int b = 0; b < B; b++) need = std::max(need, e->slot_seq[(size_t)(scan_idx[b] & ~ROAM_STEP_NEW_SEQUENCE)]);
        if (need > e->up_waited) {
            if (e->up_seq - need < 16) HIP_TRY(ctx, hipStreamWaitEvent(sA, e->ev_up_ring[need & 15], 0));
            else { HIP_TRY(ctx, hipStreamWaitEvent(sA, ctx->ev_up, 0)); need = e->up_seq; }      // (its event has been reused: the newest covers it)
            e->up_waited = need;
        }
        if (e->up_waited == e->up_seq) e->uploads_pending = false;
    }
    if (e->pool_dirty) { HIP_TRY(ctx, hipStreamWaitEvent(sA, e->ev_pool, 0)); e->pool_dirty = false; }   // device-to-device record copies
    int32_t *hs = e->scan_host + (size_t)pb * 2 * B;
    if (e->nstep >= 3) HIP_TRY(ctx, hipEventSynchronize(e->ev_g4[w4]));   // the staging slot's last copy has long been consumed
    for (int b = 0; b < B; b++) { hs[b] = scan_idx[b] & ~ROAM_STEP_NEW_SEQUENCE; hs[B + b] = (scan_idx[b] & ROAM_STEP_NEW_SEQUENCE) ? 1 : 0; }
    HIP_TRY(ctx, hipMemcpyAsync(e->scan_idx[pb], hs, sizeof(int32_t) * 2 * (size_t)B, hipMemcpyHostToDevice, sA));
    hipEvent_t *tr = e->tr_ev[e->nstep & 63];
    if (e->stage_ev) {
        for (auto &ev : e->tr_ev[e->nstep & 63]) if (!ev) HIP_TRY(ctx, hipEventCreate(&ev));
        if (e->rt_on) for (auto &ev : e->rt_ev[e->nstep & 63]) if (!ev) HIP_TRY(ctx, hipEventCreate(&ev));
    }
    e->tr_ev_ok[e->nstep & 63] = e->stage_ev;       // (the switch may be flipped between steps: every step remembers what it recorded)
    if (e->stage_ev) e->stage_ev_step = e->nstep;
    // the peak kernel gets its own stream: it only needs the scan indices (event ev_idx) and is joined before g4
    hipStream_t sP = ctx->stream5;
    HIP_TRY(ctx, hipEventRecord(e->ev_idx, sA));
    HIP_TRY(ctx, hipStreamWaitEvent(sP, e->ev_idx, 0));
    if (e->peaks_after_int && e->pyr_after_int && e->ev_int_valid) HIP_TRY(ctx, hipStreamWaitEvent(sP, (e->swap_warp_pyr && e->ev_emit_last) ? e->ev_emit_last : (e->peaks_after_emit && e->ev_emit) ? e->ev_emit : e->ev_int, 0));
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(e->ev[ST_PEAKS], sP));
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(e->ev_pk0, sP));
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(tr[0], sP));
    PeakSrc ps = {e->pool, (int64_t)e->rec_bytes, (int64_t)c.stride, c.payload_off, 1, e->scan_idx[pb]};
    HIP_TRY(ctx, launch_peaks(sP, ps, B, c.rows, c.clip, e->row_stage, e->stage_cap, e->row_count, e->peaks_out[pb], c.peaks_cap, e->peaks_n[pb]));
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(e->ev_pk1, sP));
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(tr[5], sP));
    HIP_TRY(ctx, hipEventRecord(e->ev_peaks, sP));
    if (e->warp_after_int && e->ev_int_valid) HIP_TRY(ctx, hipStreamWaitEvent(sA, e->ev_int, 0));
    if (e->swap_warp_pyr && e->nstep >= 2 && e->ev_emit2[e->nstep & 1] && e->ev_int_valid) HIP_TRY(ctx, hipStreamWaitEvent(sA, e->ev_emit2[e->nstep & 1], 0));
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(tr[1], sA));
    HIP_TRY(ctx, launch_warp_gather(sA, e->warp_map, pool_warp_src(e, e->scan_idx[pb]), B, c.rows, c.clip, next, e->pd.lane_stride, e->warp_dark_zero));
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(tr[2], sA));
    HIP_TRY(ctx, hipEventRecord(e->ev_warp, sA));                         // end of stage A
    HIP_TRY(ctx, hipStreamWaitEvent(sB, e->ev_warp, 0));
    if (e->swap_warp_pyr) {
        if (e->nstep >= 1) HIP_TRY(ctx, hipStreamWaitEvent(sB, e->ev_klt[(e->nstep + 3) & 3], 0));      // the tracker of the step before this one
    } else
    if (e->pyr_after_int && e->ev_int_valid) HIP_TRY(ctx, hipStreamWaitEvent(sB, (e->pyr_after_emit && e->ev_emit) ? e->ev_emit : e->ev_int, 0));
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(tr[3], sB));
    HIP_TRY(ctx, launch_build_pyramid(sB, next, e->pd, B, e->pyr_dark));
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(tr[4], sB));
    HIP_TRY(ctx, hipEventRecord(e->ev_join, sB));                         // end of stage B
    HIP_TRY(ctx, hipStreamWaitEvent(st, e->ev_join, 0));
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(e->ev[ST_KLT], st));
    if (any_new) {
        // lanes that start a NEW sequence on this scan drop their features: nothing is tracked, the pose stays, and the retrack
        // branch detects the sequence's first features on this scan (appendNewFeatures(prevImgCart, empty), RawROAMSystem.py:150)
        hipLaunchKernelGGL(new_sequence_kernel, dim3((B + 255) / 256), dim3(256), 0, st, e->feat_n, e->scan_idx[pb] + B, B);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, launch_klt(st, prev, next, e->pd, e->feat, e->feat_n, KM, KS, B, e->klt_next, e->klt_status, e->klt_err));
    HIP_TRY(ctx, hipEventRecord(e->ev_klt[k4], st));
    hipLaunchKernelGGL(g1_good_kernel, dim3(B), dim3(256), 0, st, e->feat, e->feat_n, e->klt_next, e->klt_status, e->klt_err,
                       e->good_old, e->good_new, e->good_idx, e->good_n, KM);
    HIP_TRY(ctx, hipGetLastError());
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(e->ev[ST_GRAPH], st));
    if (c.reject_outliers) {
        HIP_TRY(ctx, launch_consistency_graph(st, e->good_old, e->good_new, e->good_n, KM, KS, B, 0.5 / M_PER_PX, e->adj, nw));
        if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(e->ev[ST_CLIQUE], st));
        HIP_TRY(ctx, launch_max_clique(st, e->adj, e->good_n, KM, KS, nw, B, c.clique_node_limit, e->cq_stack, e->cq_mask, e->cq_n, e->cq_flags, e->cq_order));
    } else {
        if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(e->ev[ST_CLIQUE], st));
        hipLaunchKernelGGL(fill_mask_kernel, dim3(B), dim3(256), 0, st, e->cq_mask, e->good_n, e->cq_n, e->cq_flags);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(e->ev[ST_KABSCH], st));
    hipLaunchKernelGGL(g2_inliers_kernel, dim3(B), dim3(256), 0, st, e->good_old, e->good_new, e->good_idx, e->good_n, e->cq_mask,
                       e->kf_pose, e->kf_und, e->kf_und_tmp, e->kab_src, e->kab_tgt, e->p_w, e->p_jt, e->feat, e->in_n, e->kab_out, e->pose,
                       c.motion_distortion ? e->T_wj0 : nullptr, e->T_init);                 // (T_wj0 / T_init feed the LM solve only)
    HIP_TRY(ctx, hipGetLastError());
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(e->ev[ST_LM], st));
    if (c.motion_distortion) {
        MdsProblemDesc P;
        P.T_wj0 = e->T_wj0; P.T_init = e->T_init; P.p_w = e->p_w; P.p_jt = e->p_jt; P.count = e->in_n;
        P.N = KM; P.nstride = KS; P.nmax = KM; P.B = B; P.period = 0.25;
        for (int i = 0; i < 5; i++) P.sigma5[i] = c.sigma5[i];
        P.big = e->lm_big; P.big_slot = e->lm_big_slot; e->lm_big_slot ^= 1;
        if (e->lm_side) {                                                    // batches: the workgroup form beside the wave form (the peaks' stream is idle here)
            if (!e->ev_lm[0]) { HIP_TRY(ctx, hipEventCreateWithFlags(&e->ev_lm[0], hipEventDisableTiming)); HIP_TRY(ctx, hipEventCreateWithFlags(&e->ev_lm[1], hipEventDisableTiming)); }
            P.side = ctx->stream5; P.ev_fork = e->ev_lm[0]; P.ev_join = e->ev_lm[1];
        }
        HIP_TRY(ctx, launch_mds_solve(st, P, e->lm_work, e->lm_out, e->lm_nfev, e->lm_info, nullptr, nullptr));
    }
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(e->ev[ST_GLUE], st));
    HIP_TRY(ctx, hipStreamWaitEvent(st, e->ev_peaks, 0));
    hipLaunchKernelGGL(g4_update_kernel, dim3(B), dim3(256), 0, st, c, e->lm_out, e->lm_nfev, e->lm_info, e->kab_out, e->pose,
                       e->vel, e->kf_pose, e->kf_und, e->kf_und_tmp, e->p_jt, e->in_n, e->good_n, e->feat_n, e->peaks_n[pb],
                       e->cq_flags, res_slot, e->scan_idx[pb], e->kf_vel, e->kf_scan, e->kf_fresh, e->kf_live, e->map_store, e->map_n,
                       e->map_cap, (e->rt_on && e->rt_mode) ? (e->rt_mode == 2 ? 2 : 1) : 0, e->rt.rt_lane, e->rt.rt_scan, e->rt.rt_n);
    HIP_TRY(ctx, hipGetLastError());
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(e->ev[ST_RETRACK], st));
    if (e->rt_on && e->rt_mode) {
        // lanes that ran out of features (flag bit 2; listed by g4_update_kernel's last block): appendNewFeatures on the current scan +
        // keyframe refresh, on the device
        e->rt.res = res_slot;
        if (e->pyr_after_int && !e->ev_int) HIP_TRY(ctx, hipEventCreateWithFlags(&e->ev_int, hipEventDisableTiming));
        if ((e->pyr_after_emit || e->peaks_after_emit) && e->pyr_after_int && !e->ev_emit && B >= 256) HIP_TRY(ctx, hipEventCreateWithFlags(&e->ev_emit, hipEventDisableTiming));
        if (e->swap_warp_pyr && e->pyr_after_int && B >= 256) {
            // (swap experiment: the "first bookkeeping kernels are out" event alternates between two objects, so that a wait enqueued two
            // steps later still finds this step's record; ev_emit is made to point at the one just recorded for the peaks' wait)
            if (!e->ev_emit2[e->nstep & 1]) HIP_TRY(ctx, hipEventCreateWithFlags(&e->ev_emit2[e->nstep & 1], hipEventDisableTiming));
        }
        hipEvent_t emit_ev = (e->swap_warp_pyr && e->ev_emit2[e->nstep & 1]) ? e->ev_emit2[e->nstep & 1] : e->ev_emit;
        if (e->det_side.chunk > 0 && !e->det_side.ev_i[0]) {
            e->det_side.st = sB;
            for (int i = 0; i < 4; i++) { HIP_TRY(ctx, hipEventCreateWithFlags(&e->det_side.ev_i[i], hipEventDisableTiming)); HIP_TRY(ctx, hipEventCreateWithFlags(&e->det_side.ev_d[i], hipEventDisableTiming)); }
        }
        HIP_TRY(ctx, launch_retrack(st, e->rt, B, e->stage_ev ? e->rt_ev[e->nstep & 63] : nullptr, RT_TRACE_CHUNKS, e->pyr_after_int ? e->ev_int : nullptr, e->pyr_after_int - 1, emit_ev,
                                    e->det_side.chunk > 0 ? &e->det_side : nullptr));
        if (e->swap_warp_pyr) e->ev_emit_last = emit_ev;
        if (e->pyr_after_int) e->ev_int_valid = true;
        if (e->rt_mode == 2) e->rt_floor = std::min(KS, e->kmax() + 256);
    }
    e->rt_ev_ok[e->nstep & 63] = e->rt_on && e->rt_mode && e->stage_ev;
    if (e->stage_ev) HIP_TRY(ctx, hipEventRecord(e->ev[ST_COUNT], st));
    HIP_TRY(ctx, hipEventRecord(e->ev_g4[k4], st));
    // per-step result record -> pinned host ring: roam_engine_step_results(step) waits for THIS copy only
    // (on the compute stream itself: a side stream waiting on an event here cost 11 % of the step rate - the extra stream
    // shares a hardware queue with one of the pipeline's streams and serialises it; the 0.5 MB copy takes ~20 us)
    HIP_TRY(ctx, hipMemcpyAsync(e->results_host + (size_t)rs * B, res_slot, sizeof(roam_lane_result) * (size_t)B, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipEventRecord(e->ev_res[rs], st));
    e->cur = (e->cur + 1) & 3;
    e->pk = pb;
    e->nstep++;
    e->stepped = true;
    return ROAM_OK;
}

int32_t roam_engine_results(roam_ctx *ctx, roam_lane_result *out, int32_t n)
{
    ENGINE();
    ARG_CHECK(ctx, out && n >= 1 && n <= e->B);
    if (e->nstep == 0) { ROAM_SET_ERR(ctx, "no step enqueued"); return ROAM_E_STATE; }
    return roam_engine_step_results(ctx, e->nstep - 1, out, n);
}

int32_t roam_engine_step_results(roam_ctx *ctx, int64_t step, roam_lane_result *out, int32_t n)
{
    ENGINE();
    ARG_CHECK(ctx, out && n >= 1 && n <= e->B);
    if (step < 0 || step >= e->nstep || step < e->nstep - RES_RING) {
        ROAM_SET_ERR(ctx, "step %lld is not in the result ring (steps %lld..%lld)", (long long)step,
                     (long long)std::max<int64_t>(0, e->nstep - RES_RING), (long long)e->nstep - 1);
        return ROAM_E_STATE;
    }
    const int rs = (int)(step % RES_RING);
    HIP_TRY(ctx, hipEventSynchronize(e->ev_res[rs]));      // waits for that step's records only; later steps keep running
    memcpy(out, e->results_host + (size_t)rs * e->B, sizeof(roam_lane_result) * (size_t)n);
    if (e->rt_on && e->rt_floor > 320 && step == e->nstep - 1 && e->rt_mode != 2) {
        // a forced (mode 2) step inflated the launch width of every later KLT / graph launch; the records of the LATEST step say
        // what every lane really holds, and lanes only shrink (or re-detect to <= 60 + 256) from here on
        const roam_lane_result *r = e->results_host + (size_t)rs * e->B;
        int m = 320;
        for (int b = 0; b < e->B; b++) m = std::max(m, (int)((r[b].flags & 8) ? r[b].n_after_retrack : r[b].n_inliers));
        e->rt_floor = std::min(e->rt_floor, (m + 63) & ~63);
    }
    return ROAM_OK;
}

int32_t roam_engine_set_retrack(roam_ctx *ctx, int32_t mode)
{
    ENGINE();
    ARG_CHECK(ctx, mode >= 0 && mode <= 2);
    if (!e->rt_on) { ROAM_SET_ERR(ctx, "engine created without retrack_on_device"); return ROAM_E_STATE; }
    e->rt_mode = mode;
    return ROAM_OK;
}

int32_t roam_engine_steps_enqueued(roam_ctx *ctx, int64_t *nstep)
{
    ENGINE();
    ARG_CHECK(ctx, nstep);
    *nstep = e->nstep;
    return ROAM_OK;
}

int32_t roam_engine_lane_features(roam_ctx *ctx, int32_t lane, float *pts, int32_t cap, int32_t *K)
{
    ENGINE();
    ARG_CHECK(ctx, lane >= 0 && lane < e->B && pts && K && cap >= 0);
    int32_t n = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n, e->feat_n + lane, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *K = n;
    const int m = n < cap ? n : cap;
    if (m > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(pts, e->feat + (size_t)lane * KS * 2, sizeof(float) * 2 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return n > cap ? ROAM_E_CAPACITY : ROAM_OK;
}

int32_t roam_engine_lane_peaks(roam_ctx *ctx, int32_t lane, int32_t *out, int64_t cap, int64_t *n_out)
{
    ENGINE();
    ARG_CHECK(ctx, lane >= 0 && lane < e->B && out && n_out && cap >= 0);
    int32_t n = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n, e->peaks_n[e->pk] + lane, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n;
    int64_t m = n < cap ? n : cap;
    if (m > e->cfg.peaks_cap) m = e->cfg.peaks_cap;
    if (m > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(out, e->peaks_out[e->pk] + (size_t)lane * e->cfg.peaks_cap * 2, sizeof(int32_t) * 2 * (size_t)m,
                                    hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return (n > cap || n > e->cfg.peaks_cap) ? ROAM_E_CAPACITY : ROAM_OK;
}

int32_t roam_engine_doh_maxima(roam_ctx *ctx, int32_t pool_idx, const double *sigmas, int32_t num_sigma, double threshold,
                               int32_t *out_rcs, double *out_val, int32_t cap, int32_t *n_out)
{
    ENGINE();
    ARG_CHECK(ctx, pool_idx >= 0 && pool_idx < e->cfg.pool_scans);
    return roam_doh_maxima_record_device(ctx, e->pool + (size_t)pool_idx * e->rec_bytes, e->cfg.rows, e->cfg.stride,
                                         e->cfg.payload_off, e->cfg.clip, sigmas, num_sigma, threshold, out_rcs, out_val,
                                         cap, n_out);
}

int32_t roam_engine_lane_image(roam_ctx *ctx, int32_t lane, int32_t level, uint8_t *out, int64_t cap)
{
    ENGINE();
    ARG_CHECK(ctx, lane >= 0 && lane < e->B && level >= 0 && level < ROAM_PYR_LEVELS && out);
    const size_t n = (size_t)e->pd.w[level] * e->pd.h[level];
    ARG_CHECK(ctx, cap >= (int64_t)n);
    HIP_TRY(ctx, hipMemcpyAsync(out, e->pyr[e->cur] + (size_t)lane * e->pd.lane_stride + e->pd.off[level], n,
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ROAM_OK;
}

int32_t roam_engine_set_stage_events(roam_ctx *ctx, int32_t on)
{
    ENGINE();
    if (!e->stage_ev_forced) e->stage_ev = on != 0;
    return ROAM_OK;
}

int32_t roam_engine_stage_times(roam_ctx *ctx, float *ms_out, const char **names_out, int32_t cap, int32_t *n)
{
    ENGINE();
    ARG_CHECK(ctx, ms_out && n && cap >= ST_COUNT);
    if (!e->stepped) { ROAM_SET_ERR(ctx, "no step recorded"); return ROAM_E_STATE; }
    if (!e->stage_ev) { ROAM_SET_ERR(ctx, "stage events are off (roam_engine_set_stage_events)"); return ROAM_E_STATE; }
    if (e->stage_ev_step != e->nstep - 1) { ROAM_SET_ERR(ctx, "the last step was enqueued with stage events off: run a step first"); return ROAM_E_STATE; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < ST_COUNT; i++) {
        float ms = 0;
        hipEvent_t *tl = e->tr_ev[(e->nstep - 1) & 63];                                     // stage A / B boundaries of the last step
        if (i == ST_PEAKS) HIP_TRY(ctx, hipEventElapsedTime(&ms, tl[0], tl[5]));
        else if (i == ST_WARP) HIP_TRY(ctx, hipEventElapsedTime(&ms, tl[1], tl[2]));
        else if (i == ST_PYR) HIP_TRY(ctx, hipEventElapsedTime(&ms, tl[3], tl[4]));
        else HIP_TRY(ctx, hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]));
        ms_out[i] = ms;
        if (names_out) names_out[i] = kStageNames[i];
    }
    *n = ST_COUNT;
    return ROAM_OK;
}

// average launch time of one front-end kernel over the last `last_steps` steps (at most 64), from the event
// pairs recorded on the stream the kernel ran on - the live, in-step duration (other kernels may share the GPU)
int32_t roam_engine_kernel_avg(roam_ctx *ctx, const char *name, int32_t last_steps, float *avg_ms, int32_t *n_used)
{
    ENGINE();
    ARG_CHECK(ctx, name && last_steps >= 1 && avg_ms && n_used);
    if (!e->stage_ev) { ROAM_SET_ERR(ctx, "stage events are off (roam_engine_set_stage_events)"); return ROAM_E_STATE; }
    int k = !strcmp(name, "ingest_peaks") ? 0 : (!strcmp(name, "warp_quantise") ? 1 : (!strcmp(name, "pyramid") ? 2 : -1));
    // the two image-scale kernels of the detection: the FIRST chunk of every step (min(retrack_slots, lanes flagged in that step)
    // detections; steps without device-side detection do not count) - roam_engine_kernel_chunk_ms has every chunk
    const int kd = !strcmp(name, "doh_integral") ? 0 : (!strcmp(name, "doh_det_maxima") ? 1 : -1);
    if (k < 0 && kd < 0) { ROAM_SET_ERR(ctx, "unknown kernel '%s'", name); return ROAM_E_ARG; }
    if (kd >= 0 && !e->rt_on) { ROAM_SET_ERR(ctx, "engine created without retrack_on_device"); return ROAM_E_STATE; }
    if (!e->stepped) { ROAM_SET_ERR(ctx, "run a step first"); return ROAM_E_STATE; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int64_t n = std::min<int64_t>(std::min<int64_t>(last_steps, e->nstep), 64);
    double sum = 0;
    if (kd >= 0) {
        int64_t used = 0;
        for (int64_t i = e->nstep - n; i < e->nstep; i++) {
            if (!e->rt_ev_ok[i & 63]) continue;
            float ms = 0;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, e->rt_ev[i & 63][kd], e->rt_ev[i & 63][kd + 1]));
            sum += ms; used++;
        }
        *avg_ms = used ? (float)(sum / (double)used) : 0.f;
        *n_used = (int32_t)used;
        return ROAM_OK;
    }
    int64_t used = 0;
    for (int64_t i = e->nstep - n; i < e->nstep; i++) {
        if (!e->tr_ev_ok[i & 63]) continue;                                   // enqueued while the events were switched off
        float ms = 0;
        const int a0 = k == 2 ? 3 : (k == 1 ? 1 : 0), a1 = k == 2 ? 4 : (k == 1 ? 2 : 5);   // peaks and pyramid: their own streams' pairs
        HIP_TRY(ctx, hipEventElapsedTime(&ms, e->tr_ev[i & 63][a0], e->tr_ev[i & 63][a1]));
        sum += ms; used++;
    }
    *avg_ms = used ? (float)(sum / (double)used) : 0.f;
    *n_used = (int32_t)used;
    return ROAM_OK;
}

// detections per launch of a detection kernel inside a step (retrack_slots, or the chunk of the two-stream form)
int32_t roam_engine_detect_chunk(roam_ctx *ctx, int32_t *chunk)
{
    ENGINE();
    ARG_CHECK(ctx, chunk);
    if (!e->rt_on) { ROAM_SET_ERR(ctx, "engine created without retrack_on_device"); return ROAM_E_STATE; }
    *chunk = e->det_chunk();
    return ROAM_OK;
}

// launch durations of a detection kernel, chunk by chunk: ms_out[s * chunks + c] = chunk c of the s-th of the last `steps_out`
// steps (oldest first; -1 for a step without device-side detection).  A step launches `chunks` = ceil(lanes / retrack_slots) chunks
// (at most RT_TRACE_CHUNKS are traced) whatever the number of lanes that re-detect - only the device knows it - and chunk c holds
// clamp(n - c * retrack_slots, 0, retrack_slots) of the step's n detections: the caller has n in the step's result records
int32_t roam_engine_kernel_chunk_ms(roam_ctx *ctx, const char *name, int32_t last_steps, float *ms_out, int32_t cap, int32_t *chunks,
                                    int32_t *steps_out)
{
    ENGINE();
    ARG_CHECK(ctx, name && last_steps >= 1 && ms_out && chunks && steps_out);
    const int kd = !strcmp(name, "doh_integral") ? 0 : (!strcmp(name, "doh_det_maxima") ? 1 : -1);
    if (kd < 0) { ROAM_SET_ERR(ctx, "unknown kernel '%s'", name); return ROAM_E_ARG; }
    if (!e->rt_on) { ROAM_SET_ERR(ctx, "engine created without retrack_on_device"); return ROAM_E_STATE; }
    if (!e->stepped) { ROAM_SET_ERR(ctx, "run a step first"); return ROAM_E_STATE; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t n = std::min<int64_t>(std::min<int64_t>(last_steps, e->nstep), 64);
    const int nchunk = std::min((e->B + e->det_chunk() - 1) / e->det_chunk(), RT_TRACE_CHUNKS);
    ARG_CHECK(ctx, (int64_t)cap >= n * nchunk);
    for (int64_t i = e->nstep - n, s = 0; i < e->nstep; i++, s++)
        for (int c = 0; c < nchunk; c++) {
            float ms = -1.f;
            if (e->rt_ev_ok[i & 63]) HIP_TRY(ctx, hipEventElapsedTime(&ms, e->rt_ev[i & 63][3 * c + kd], e->rt_ev[i & 63][3 * c + kd + 1]));
            ms_out[s * nchunk + c] = ms;
        }
    *chunks = nchunk;
    *steps_out = (int32_t)n;
    return ROAM_OK;
}

// re-launch one streaming kernel of the step `reps` times over all lanes (inputs resident, outputs
// overwritten in the scratch "next" pyramid / peak buffers) and time it with HIP events on the
// context stream.  algo_bytes = algorithmic HBM bytes per launch.
int32_t roam_engine_time_kernel(roam_ctx *ctx, const char *name, int32_t reps, float *avg_ms, double *algo_bytes)
{
    ENGINE();
    ARG_CHECK(ctx, name && reps >= 1 && avg_ms);
    if (!e->stepped) { ROAM_SET_ERR(ctx, "run a step first"); return ROAM_E_STATE; }
    hipStream_t st = ctx->stream;
    const roam_engine_cfg &c = e->cfg;
    const int B = e->B;
    uint8_t *next = e->pyr[(e->cur + 1) & 3];       // not the live "previous" pyramid
    double bytes = 0;
    hipEvent_t a, b;
    HIP_TRY(ctx, hipEventCreate(&a));
    HIP_TRY(ctx, hipEventCreate(&b));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipEventRecord(a, st));
    for (int r = 0; r < reps; r++) {
        if (!strcmp(name, "warp_quantise")) {
            HIP_TRY(ctx, launch_warp_gather(st, e->warp_map, pool_warp_src(e, e->scan_idx[e->pk]), B, c.rows, c.clip, next, e->pd.lane_stride, e->warp_dark_zero));
            bytes = (double)B * ((double)c.rows * c.clip + (double)e->W * e->W);
        } else if (!strcmp(name, "ingest_peaks")) {
            PeakSrc ps = {e->pool, (int64_t)e->rec_bytes, (int64_t)c.stride, c.payload_off, 1, e->scan_idx[e->pk]};
            hipError_t er = launch_peaks(st, ps, B, c.rows, c.clip, e->row_stage, e->stage_cap, e->row_count, e->peaks_out[(e->pk + 1) % 3], c.peaks_cap, e->peaks_n[(e->pk + 1) % 3]);
            HIP_TRY(ctx, er);
            bytes = (double)B * ((double)c.rows * c.clip);
        } else if (!strcmp(name, "doh_integral") || !strcmp(name, "doh_det_maxima") || !strncmp(name, "doh_fused", 9)) {
            // the image-scale kernels of the device-side retrack over all scratch slots (per launch: `slots` detections)
            if (!e->rt_on) { hipEventDestroy(a); hipEventDestroy(b); ROAM_SET_ERR(ctx, "engine created without retrack_on_device"); return ROAM_E_STATE; }
            // ("doh_fused:<d>": with diagnostics word d - ablations of retrack_fused.inc, measurement only)
            const int P = e->rt.slots, which = !strcmp(name, "doh_integral") ? 0 : (!strcmp(name, "doh_det_maxima") ? 1 : 2 + (name[9] == ':' ? atoi(name + 10) : 0));
            if (which >= 2) { const int32_t rc_ = fused_tables(ctx, e); if (rc_ != ROAM_OK) return rc_; }       // (made on first use)
            if (r == 0) {
                std::vector<int32_t> sc(P);
                for (int i = 0; i < P; i++) sc[i] = e->last_scan[i % B] >= 0 ? e->last_scan[i % B] : 0;
                HIP_TRY(ctx, hipMemcpy(e->rt.rt_scan, sc.data(), sizeof(int32_t) * (size_t)P, hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMemcpy(e->rt.rt_n, &P, sizeof(int32_t), hipMemcpyHostToDevice));
                if (which == 1) HIP_TRY(ctx, launch_retrack_part(st, e->rt, P, 0));      // valid integral images to work on
                HIP_TRY(ctx, hipStreamSynchronize(st));
                HIP_TRY(ctx, hipEventRecord(a, st));
            }
            HIP_TRY(ctx, launch_retrack_part(st, e->rt, P, which));
            if (which >= 1) HIP_TRY(ctx, hipMemsetAsync(e->rt.cand_n, 0, sizeof(int32_t) * (size_t)P, st));   // (no bookkeeping follows that would clear the counts)
            const double npx = (double)e->W * e->W;
            // algorithmic bytes per detection (strict): integral image = polar payload read + float64 image written once
            // (the 4-byte sampling-map word per pixel it also reads is the same geometry table for every detection and mostly comes
            // out of L2: not input data, not counted since round 4); determinants + maxima = float64 image read once (the
            // candidates it writes are a few KB)
            // the fused kernel: the polar payload is all a detection reads; the float64 image never leaves the CU
            // round 6: the part of the image that exists - the tiles the determinant kernel reads, the only ones the integral kernel
            // writes (retrack_build_phases): 87.7 % of a 2024 x 2024 image; rounds 2-5 counted the whole image for both kernels
            const double ipx = e->rt_image_px > 0 ? (double)e->rt_image_px : npx;
            bytes = (double)P * (which == 0 ? ((double)c.rows * c.clip + ipx * 8.0) : (which == 1 ? ipx * 8.0 : (double)c.rows * c.clip));
        } else if (!strcmp(name, "pyramid")) {
            HIP_TRY(ctx, launch_build_pyramid(st, next, e->pd, B, e->pyr_dark));
            double rd = 0, wr = 0;
            for (int l = 0; l + 1 < ROAM_PYR_LEVELS; l++) { rd += (double)e->pd.w[l] * e->pd.h[l]; wr += (double)e->pd.w[l + 1] * e->pd.h[l + 1]; }
            bytes = (double)B * (rd + wr);
        } else {
            hipEventDestroy(a); hipEventDestroy(b);
            ROAM_SET_ERR(ctx, "unknown kernel '%s'", name);
            return ROAM_E_ARG;
        }
    }
    HIP_TRY(ctx, hipEventRecord(b, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a); hipEventDestroy(b);
    *avg_ms = ms / reps;
    if (algo_bytes) *algo_bytes = bytes;
    return ROAM_OK;
}

// diagnostics: the image-scale detection kernels of `n_slots` detections (scratch slot i works on the scan lane i % lanes saw last) in the
// two-kernel form (fused = 0: rt_integral_kernel + rt_det_strip_kernel) or as rt_fused_kernel (fused = 1), candidates sorted into
// (row, column, layer) order.  Out: per slot the count (cand_n, may exceed cap_per_slot) and the first cap_per_slot candidates; S_out
// (optional, W x W doubles): the integral image of slot `s_slot` as that form computed it.  The parity tests hold the two forms equal.
int32_t roam_engine_debug_detect(roam_ctx *ctx, int32_t fused, int32_t n_slots, int32_t cap_per_slot, uint32_t *rc_out, double *val_out,
                                 int32_t *n_out, int32_t s_slot, double *S_out)
{
    ENGINE();
    ARG_CHECK(ctx, n_slots >= 1 && cap_per_slot >= 0 && n_out);
    if (!e->rt_on || !e->stepped) { ROAM_SET_ERR(ctx, "debug_detect: needs retrack_on_device and one step"); return ROAM_E_STATE; }
    ARG_CHECK(ctx, n_slots <= e->rt.slots && n_slots <= e->B && (!S_out || (s_slot >= 0 && s_slot < n_slots)));
    hipStream_t st = ctx->stream;
    const int P = n_slots, B = e->B;
    HIP_TRY(ctx, hipStreamSynchronize(st));
    std::vector<int32_t> sc(P);
    for (int i = 0; i < P; i++) sc[i] = e->last_scan[i % B] >= 0 ? e->last_scan[i % B] : 0;
    HIP_TRY(ctx, hipMemcpy(e->rt.rt_scan, sc.data(), sizeof(int32_t) * (size_t)P, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(e->rt.rt_n, &P, sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemsetAsync(e->rt.cand_n, 0, sizeof(int32_t) * (size_t)P, st));
    if (fused) { const int32_t rc_ = fused_tables(ctx, e); if (rc_ != ROAM_OK) return rc_; }
    if (fused) HIP_TRY(ctx, launch_retrack_part(st, e->rt, P, S_out ? 3 : 2));
    else { HIP_TRY(ctx, launch_retrack_part(st, e->rt, P, 0)); HIP_TRY(ctx, launch_retrack_part(st, e->rt, P, 1)); }
    HIP_TRY(ctx, launch_retrack_emit(st, e->rt, P));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipMemcpy(n_out, e->rt.cand_n, sizeof(int32_t) * (size_t)P, hipMemcpyDeviceToHost));
    const int keep = std::min(cap_per_slot, BP_MAX_PTS);
    for (int i = 0; i < P && keep > 0; i++) {
        if (rc_out) HIP_TRY(ctx, hipMemcpy(rc_out + (size_t)i * cap_per_slot, e->rt.cand_rc + (size_t)i * BP_MAX_PTS, sizeof(uint32_t) * (size_t)keep, hipMemcpyDeviceToHost));
        if (val_out) HIP_TRY(ctx, hipMemcpy(val_out + (size_t)i * cap_per_slot, e->rt.cand_val + (size_t)i * BP_MAX_PTS, sizeof(double) * (size_t)keep, hipMemcpyDeviceToHost));
    }
    if (S_out)
        HIP_TRY(ctx, hipMemcpy2D(S_out, sizeof(double) * (size_t)e->W, e->rt.S + (size_t)s_slot * e->rt.SP * e->W, sizeof(double) * (size_t)e->rt.SP,
                                 sizeof(double) * (size_t)e->W, (size_t)e->W, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemset(e->rt.cand_n, 0, sizeof(int32_t) * (size_t)P));     // the lists are clean for the next detection
    return ROAM_OK;
}

}  // extern "C"
