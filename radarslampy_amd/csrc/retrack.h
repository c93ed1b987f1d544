// retrack.h - arguments of the device-side feature (re)detection (retrack.hip), filled by the engine
#pragma once
#include "roam_internal.h"
#include "blobprune.h"

#define RT_TWO_PASS_SLOTS 200     // = RI_MIN_DETECTIONS of retrack.hip: chunks of at least that many detections take the one-sweep kernel
enum { RT_F_CAND_OVERFLOW = 1, RT_F_TREE_OVERFLOW = 2, RT_F_PAIR_OVERFLOW = 4, RT_F_FEAT_OVERFLOW = 8 };

struct RtArgs {
    // geometry / parameters
    int W, rows, cols, stride, payload_off, slots, size1, size2;
    int64_t rec_bytes;
    double sigma1, sigma2, threshold;
    // engine state
    const uint8_t *pool;
    const uint32_t *map;
    const uint32_t *phlist;         // the phases the one-sweep integral kernel walks (retrack_build_phases): [0] = count, then the phases
    const uint32_t *boxtab;         // per (band, group, column wave) of the one-sweep integral kernel: the polar footprint of the patch,
                                    // {min ix | max ix << 16, min iy | max iy << 16} (max ix = 0xffff: nothing inside the maximum range)
    const uint32_t *darktab;        // per column strip of the determinant kernel: which steps see nothing but pixels beyond the maximum range
                                    // (retrack_darktab_words: 3 x 8 words per strip - step is dark / its load / its LDS fill can be skipped)
    float *feat;
    int32_t *feat_n;
    const double *vel;
    double *kf_und;
    roam_lane_result *res;          // result records of the step being amended (may be null)
    // flagged lanes (device-built): rt_n lanes, lane / pool scan of each
    int32_t *rt_n, *rt_lane, *rt_scan;
    // per-slot scratch (slots entries): the image-scale kernels work on chunks of `slots` detections
    double *S;                      // W rows x SP float64: integral image, rows padded to whole 128-byte lines
    int32_t *col_done;              // two-pass integral image: finished bands per (slot, column group of 64): RT_TWO_PASS_SLOTS x 64, zero between launches
    double *colT;                   // two-pass integral image (small chunks): ceil(W / 64) band totals x W columns, RT_TWO_PASS_SLOTS slots
    int SP;                         // row pitch of S in elements (a multiple of 16, >= W)
    // per-detection scratch (one entry per lane: the bookkeeping kernels run once over all detections of a step)
    uint32_t *cand_rc;              // BP_MAX_PTS: row << 16 | col << 2 | layer (appended by the determinant kernel, sorted by rt_emit_kernel)
    double *cand_val;               // BP_MAX_PTS
    int32_t *cand_n;                // maxima found (may exceed BP_MAX_PTS: RT_F_CAND_OVERFLOW, the kept subset is then arbitrary)
    BpTask *tasks;                  // BP_MAX_TASKS
    uint32_t *pairs;                // BP_MAX_PAIRS + 1
    uint16_t *order;                // BP_MAX_PAIRS + 1
    uint32_t *ovbits;               // (BP_MAX_PAIRS + 31) / 32 + 1
    uint16_t *bigtab;               // 2 x 131072 (set tables of the rare > 4914-pair case)
    double *kp;                     // BP_MAX_PTS x 3 keypoints in adaptiveNMS priority order
    int32_t *kp_n, *slot_flags;
    int32_t *ssc_work;              // 4 x BP_MAX_PTS
    int32_t *sel, *sel_n;           // BP_MAX_PTS, 1
    int32_t *blob_order_buf;        // one entry per lane: scratch of the order below
    const int32_t *blob_order;      // the detections by falling number of candidates (set by launch_retrack for its bookkeeping kernels: the
                                    // longest lists start first; null = slot order)
    // fused detection kernel (retrack_fused.inc: integral image + determinants + maxima in one kernel, chunks of >= RT_TWO_PASS_SLOTS detections)
    int fused;                      // 1: rt_fused_kernel serves those chunks, rt_integral_kernel / rt_det_strip_kernel return at once for them
    const uint32_t *fd_mapT;        // the sampling map transposed (W x W)
    const uint32_t *fd_boxtab;      // polar footprint per (band + 1, block): retrack_fused_boxtab_words(W) words
    const uint32_t *fd_darktab;     // per band: the dark steps (the format of darktab, made from the transposed map)
    double *fd_halo;                // per slot: two hand-off buffers of fd_halo_words doubles (the 32 rows a band leaves to the band below)
    int64_t fd_halo_words;
    double *fd_cc;                  // per slot: 2048 column totals of the bands above
};

hipError_t launch_retrack_collect(hipStream_t st, const roam_lane_result *res, const int32_t *scan_idx, int B, int force_all, const RtArgs &a);
// runs the detection for the rt_n flagged lanes (device count, at most B), in chunks of a.slots
// trace (optional): three events per chunk (the first ntrace chunks), recorded around its kernels - before the integral image, between
// it and the determinants, after the determinants (live kernel durations for bench.py's roofline)
#define RT_TRACE_CHUNKS 16
// (ROAM_DET_SIDE) the determinants of chunk c on a second stream beside the integral images of chunk c + 1: chunks of `chunk`
// detections, the integral images alternating between two banks of the scratch (slots >= 2 * chunk)
struct RtSide { hipStream_t st; hipEvent_t ev_i[4], ev_d[4]; int chunk; };
bool retrack_sided(const RtArgs &a, int B, const RtSide *side);      // does launch_retrack take that form for this engine?
hipError_t launch_retrack(hipStream_t st, const RtArgs &a, int B, hipEvent_t *trace = nullptr, int ntrace = 0, hipEvent_t after_integral = nullptr, int after_det = 0, hipEvent_t after_emit = nullptr, const RtSide *side = nullptr);
hipError_t launch_ssc_batch(hipStream_t st, const double *kp, int64_t kp_stride, const int32_t *count, int kp_cap, int P,
                            int num_ret, double tol, int cols, int rows, int32_t *work, int32_t *sel, int32_t *n_sel,
                            const int32_t *n_active, int first);
// which: 0 = integral image, 1 = determinants + maxima (both: the two-kernel form, whatever a.fused says), 2 + d = the fused kernel with
// diagnostics word d (0 = none; bit 0: its integral image is written to a.S as well; higher bits: ablations, retrack_fused.inc)
hipError_t launch_retrack_part(hipStream_t st, const RtArgs &a, int P, int which);
// candidates of the first P slots sorted into (row, column, layer) order (diagnostics: the lists stay as they are)
hipError_t launch_retrack_emit(hipStream_t st, const RtArgs &a, int P);
hipError_t retrack_init();
// fills boxtab (retrack_boxtab_words(W) uint32 words) from the sampling map: geometry only, once per engine
size_t retrack_boxtab_words(int W);
// the phase list of the one-sweep integral kernel from the sampling map and the determinant kernel's dark-step table (both on the HOST):
// out = retrack_phase_words(W) uint32.  false: the image is too large for the one-sweep kernel (it is not used then)
size_t retrack_phase_words(int W);
int retrack_band_rows();              // rows of a band of the one-sweep integral kernel (a phase-list entry's tile height)
bool retrack_build_phases(const uint32_t *map_host, const uint32_t *darktab_host, int W, int cols, uint32_t *out);
hipError_t launch_retrack_boxtab(hipStream_t st, const uint32_t *map, int W, int cols, uint32_t *boxtab);
// fills darktab (retrack_darktab_words(W) zero-initialised uint32 words) from the sampling map: geometry only, once per engine
size_t retrack_darktab_words(int W);
// tables of the fused kernel: transposed map (W * W words), footprints, dark steps per band
size_t retrack_fused_boxtab_words(int W);
size_t retrack_fused_halo_words(int W);
hipError_t launch_retrack_fused_tables(hipStream_t st, const uint32_t *map, int W, int cols, uint32_t *mapT, uint32_t *boxtab, uint32_t *darktab);
hipError_t launch_retrack_darktab(hipStream_t st, const uint32_t *map, int W, int cols, uint32_t *darktab);
