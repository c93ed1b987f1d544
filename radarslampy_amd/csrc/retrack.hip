// Device-side feature (re)detection of the engine: appendNewFeatures(currImgCart, good_new) of the reference's loop
// (RawROAMSystem.py:250-271, getFeatures.py:74-118) for every lane whose step ran out of features, inside
// roam_engine_step, without a host round trip.  Per flagged lane ("slot" = one detection):
//   K1+K2 rt_integral    the float64 integral image in ONE sweep: float32 Cartesian pixel from the polar record through the engine's
//                        sampling map (the arithmetic of warp.hip, never written to memory; the polar footprint of a wave's patch
//                        staged in LDS), column cumsum in registers, row cumsum through double-buffered LDS tiles - both in NumPy's
//                        sequential order, image written once.  Chunks of fewer than RI_MIN_DETECTIONS detections (and a lane's
//                        first detection) take rt_integ_cols + rt_integ_rows instead: thousands of threads per detection
//   K3 rt_det_strip      box-filter Hessian determinants of both live layers (sigma 5.005 / 10, sizes 15 / 30; the sigma 0.01 layer
//                        is all-NaN in scikit-image and ignored): workgroups march down 62-column strips of the integral image with
//                        its rows in an LDS ring (each byte fetched ~1.4 times), over the steps of the strip that lie inside the
//                        maximum range (a table, geometry only); dxy boxes only where dxx*dyy can pass the threshold; 3x3x3 maxima above the threshold are appended to the detection's candidate list
//   K4 rt_emit           the candidates sorted into C (row, col, layer) order
//   K5 rt_blobs          one wavefront per lane: response order, scikit-image's _prune_blobs in ITS pair order (blobprune.h:
//                        cKDTree emission order + CPython set order; the tree is built level by level with one lane per node, the
//                        traversal and the set order run on lane 0 out of LDS), NumPy-1.22 argsort of the sigmas -> keypoints in
//                        adaptiveNMS's priority order
//   K6 ssc_batch         ANMS.ssc (ssc.hip)
//   K7 rt_append         [x, y] flip, vstack + drop exact duplicates keeping the first (getFeatures.py:109-112), keyframe
//                        refresh (Mapping.py:59-66 with the frame's velocity), feature count
// The image-scale kernels (K1-K3) run in chunks of `slots` detections (scratch: a 33 MB integral image per slot); K3 appends to
// per-DETECTION candidate lists, and K4-K7 run once over all detections of the step.  Every kernel exits at once for
// detections beyond the number of flagged lanes, which only the device knows.
#include "roam_internal.h"
#include "doh_common.h"
#include "blobprune.h"
#include "retrack.h"
#include <algorithm>

#define KS ROAM_MAX_FEATURES
#define CART_CENTER 1012.0
#define M_PER_PX 0.0864
#define TWO_PI 6.283185307179586476925286766559

// ------------------------------------------------------------------------------------------------ K0: flagged lanes
__global__ __launch_bounds__(256) void rt_collect_kernel(const roam_lane_result *__restrict__ res, const int32_t *__restrict__ scan_idx,
                                                         int B, int force_all, int32_t *__restrict__ rt_lane,
                                                         int32_t *__restrict__ rt_scan, int32_t *__restrict__ rt_n)
{
    __shared__ int sh[8];
    __shared__ int base_s;
    const int t = threadIdx.x;
    if (t == 0) base_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int b = b0 + t;
        const int f = (b < B && (force_all || (res[b].flags & 4))) ? 1 : 0;
        // block exclusive scan
        const int lane = t & 63, w = t >> 6;
        int inc = f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int n = __shfl_up(inc, d); if (lane >= d) inc += n; }
        if (lane == 63) sh[w] = inc;
        __syncthreads();
        int off = base_s, tot = 0;
        for (int i = 0; i < 4; i++) { if (i < w) off += sh[i]; tot += sh[i]; }
        if (f) { rt_lane[off + inc - 1] = b; rt_scan[off + inc - 1] = scan_idx[b]; }
        __syncthreads();
        if (t == 0) base_s += tot;
        __syncthreads();
    }
    if (t == 0) *rt_n = base_s;
}

// ------------------------------------------------------------------------------------------------ K1 / K2: integral image
__device__ __forceinline__ float rt_code_to_f32(uint32_t k) { return (float)__dmul_rn((double)k, 1.0 / 255.0); }
// (round 5: the float32-only decode of warp.hip - fma(k, head, k * tail) - in place of the table read was tried in the one-sweep kernel's
// taps, with byte reads and with 16-bit reads: 9.1 ms per 512 detections against 6.6 with the table; the table stays)

// Two ways to the integral image, chosen on the device by the number of detections of the chunk (only the device knows it):
//   * rt_integral_kernel (below): one workgroup per detection, the image written once - 12.7 us per detection at 512, but a chain of
//     1016 dependent phases per detection: 6.5 ms per chunk whatever its size
//   * rt_integ_cols_kernel (+ the band fix-up by its last workgroups) + rt_integ_rows_kernel: thousands of threads per detection, three times the
//     traffic, 0.26 ms for one alone
// Both are launched; the one whose regime it is not returns at once.
#define RI_MIN_DETECTIONS RT_TWO_PASS_SLOTS
__device__ __forceinline__ bool rt_one_sweep(const RtArgs &a, int first) { return a.W <= 2048 && *a.rt_n - first >= RI_MIN_DETECTIONS; }

struct __attribute__((packed)) RtU16 { uint16_t v; };
struct __attribute__((packed)) RtU32 { uint32_t v; };
// one Cartesian pixel out of the polar record (the arithmetic of warp_gather_kernel's direct path = warp_pixel); lut[k] = rt_code_to_f32(k)
__device__ __forceinline__ float rt_pixel(uint32_t m, const uint8_t *__restrict__ p, int rows, int cols, int stride, const float *lut)
{
    const int ix = m & 4095, iy = (m >> 12) & 1023;
    if (ix >= cols) return 0.f;
    const float wx1 = __fmul_rn((float)((m >> 22) & 31), 1.f / 32.f), wx0 = __fsub_rn(1.f, wx1);
    const float wy1 = __fmul_rn((float)(m >> 27), 1.f / 32.f), wy0 = __fsub_rn(1.f, wy1);
    int r0 = iy - 1, r1 = iy;
    if (r0 < 0) r0 += rows; else if (r0 >= rows) r0 -= rows;
    if (r1 >= rows) r1 -= rows;
    const uint8_t *q0 = p + r0 * stride + ix, *q1 = p + r1 * stride + ix;
    const bool i1 = ix + 1 < cols;
    // in the last column the pair is read one byte to the left and shifted, so that no load leaves the row
    const int back = i1 ? 0 : 1, sh = back * 8;
    const uint32_t w0 = reinterpret_cast<const RtU16 *>(q0 - back)->v >> sh, w1 = reinterpret_cast<const RtU16 *>(q1 - back)->v >> sh;
    const float s00 = lut[w0 & 255], s01 = i1 ? lut[w0 >> 8] : 0.f;        // lut[k] = rt_code_to_f32(k)
    const float s10 = lut[w1 & 255], s11 = i1 ? lut[w1 >> 8] : 0.f;
    float v = __fmul_rn(s00, __fmul_rn(wy0, wx0));
    v = __fadd_rn(v, __fmul_rn(s01, __fmul_rn(wy0, wx1)));
    v = __fadd_rn(v, __fmul_rn(s10, __fmul_rn(wy1, wx0)));
    v = __fadd_rn(v, __fmul_rn(s11, __fmul_rn(wy1, wx1)));
    return v;
}

// Column pass, parallel over bands of RC_BAND rows.  A pixel is a float32 of at least 2^-18 (code / 255 times a weight product that is a
// multiple of 2^-10) or zero, i.e. a multiple of 2^-41, and a column of at most 4094 of them sums to less than 2^12: every partial
// sum is EXACT in float64, in any order.  So the 64 rows of a band's column are four threads of 16 rows each (round 6; one thread per
// band column until then: 16 dependent pairs of round trips - map word, then taps - were 48 of a lone detection's 54 us in this pass):
// all 16 map words leave at once, then all 64 taps, the quarters' totals meet in LDS; the workgroup writes the band-local sums and the
// band's total; the last workgroup of a column group turns the totals into what lies above each band, and the row pass adds that in as
// it loads (all exact = NumPy's values).
#define RT_TWO_PASS_Z 8                      // detections of a chunk the two-pass kernels work on at a time (grid z / y)
#define RC_BAND 64
#define RC_Q 16                              // rows per thread: RC_BAND / 4
__global__ __launch_bounds__(256) void rt_integ_cols_kernel(RtArgs a, int first, int P)
{
    // (the detections of the chunk are walked gridDim.z at a time: a grid of one slab per possible detection was 204 800 workgroups per
    // chunk that only returned whenever the chunk belonged to the one-sweep kernel or was empty - 1 to 9 ms of dispatch beside other kernels)
    const int nls = rt_one_sweep(a, first) ? 0 : min(min(RT_TWO_PASS_SLOTS, P), *a.rt_n - first);      // (P: the chunk's scratch slots)
    if ((int)blockIdx.z >= nls) return;
    __shared__ double tot[4][64];
    __shared__ float lut[256];
    __shared__ int last_s;
    lut[threadIdx.x] = rt_code_to_f32(threadIdx.x);
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, W = a.W, band = blockIdx.y, nb = (W + RC_BAND - 1) / RC_BAND;
    for (int ls = blockIdx.z; ls < nls; ls += (int)gridDim.z) {
    const int slot = first + ls;
    const uint8_t *p = a.pool + (int64_t)a.rt_scan[slot] * a.rec_bytes + a.payload_off;
    double *S = a.S + (int64_t)ls * a.SP * W;
    const int rows = a.rows, cols = a.cols, stride = a.stride;
    const int r0 = band * RC_BAND + q * RC_Q;
    uint32_t m[RC_Q];
#pragma unroll
    for (int k = 0; k < RC_Q; k++) m[k] = (c < W && r0 + k < W) ? a.map[(int64_t)(r0 + k) * W + c] : 0xfffu;     // (bin 4095: beyond any scan, a zero pixel)
    __syncthreads();                                   // (the table; the map words are on their way)
    float v[RC_Q];
#pragma unroll
    for (int k = 0; k < RC_Q; k++) v[k] = rt_pixel(m[k], p, rows, cols, stride, lut);      // (= warp_pixel: two 16-bit loads, table decode)
    double s[RC_Q], acc = 0;
#pragma unroll
    for (int k = 0; k < RC_Q; k++) { acc = __dadd_rn(acc, (double)v[k]); s[k] = acc; }
    tot[q][lane] = acc;
    __syncthreads();
    double base = 0;
    for (int j = 0; j < q; j++) base = __dadd_rn(base, tot[j][lane]);
    if (c < W) {
#pragma unroll
        for (int k = 0; k < RC_Q; k++)
            if (r0 + k < W) S[(int64_t)(r0 + k) * a.SP + c] = __dadd_rn(base, s[k]);
        if (q == 3) __hip_atomic_store(a.colT + ((int64_t)ls * nb + band) * W + c, __dadd_rn(base, acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // band totals -> sum of the bands above (exclusive prefix per column; exact, see above), by the workgroup of this column group that
    // finishes LAST (a kernel of its own until round 6: one launch more in the chain every step enqueues); the totals are loaded 32 at a
    // time (one round trip instead of one per band).  col_done[detection][column group] counts the finished bands and is left at zero.
    // The totals cross between workgroups - between XCDs, each with an L2 of its own - as device-scope atomic stores and loads, the
    // counter after them: a __threadfence() here writes the XCD's whole L2 back, 32 MB of integral image included (151 us instead of 18).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last_s = atomicAdd(a.col_done + ls * 64 + (int)blockIdx.x, 1) == nb - 1;
    __syncthreads();
    if (last_s) {
    if (q == 0 && c < W) {
        double *T = a.colT + (int64_t)ls * nb * W + c;
        double run = 0.0;
        for (int b0 = 0; b0 < nb; b0 += 32) {
            double tv[32];
#pragma unroll
            for (int j = 0; j < 32; j++) tv[j] = b0 + j < nb ? __hip_atomic_load(T + (int64_t)(b0 + j) * W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
            for (int j = 0; j < 32; j++)
                if (b0 + j < nb) { T[(int64_t)(b0 + j) * W] = run; run = __dadd_rn(run, tv[j]); }
        }
    }
    if (threadIdx.x == 0) a.col_done[ls * 64 + (int)blockIdx.x] = 0;
    }
    __syncthreads();                                   // (tot / last_s are the next detection's)
    }
}

// Row pass: a workgroup per 16 rows; 16 lanes of wave 0, lane = row, walk sequentially along the row (the reference's summation order);
// the image streams through LDS in 16 x 256 tiles so that global accesses stay coalesced (all four waves load and store, 2 KB per row).
// Round 6, for the lone detection of a single sequence (197 us until then, for a chain of 2 024 additions):
//   * the tiles of the next RR_DEPTH steps are in flight in registers (the loads of tile i + 1 used to leave one 32-column chain ahead
//     of their use: 64 tiles x one HBM round trip);
//   * 16 rows per workgroup instead of 64: 127 workgroups instead of 32 - a CU moves ~36 GB/s of this pattern, and 32 of them needed
//     57 us for the 65 MB whatever the chain cost (ablation: profiles/r06_detection_experiments.txt);
//   * the next 16 values of the chain are read from LDS while the current 16 are added.
#define RR_DEPTH 4
#define RR_ROWS 16
#define RR_COLS 256
__global__ __launch_bounds__(256) void rt_integ_rows_kernel(RtArgs a, int first, int P)
{
    static_assert(RC_BAND % RR_ROWS == 0, "the rows of a workgroup lie in one band of the column pass");
    // (one buffer: a thread stores, and then overwrites, its own elements only; a row pitch of 258 doubles: every access below is a 16-byte one,
    // and the 16 chain lanes - 4 banks each, 4 apart - cover the 64 banks exactly)
    __shared__ __align__(16) double tile[RR_ROWS][RR_COLS + 2];
    const int nls = rt_one_sweep(a, first) ? 0 : min(min(RT_TWO_PASS_SLOTS, P), *a.rt_n - first);      // (P: the chunk's scratch slots)      // (as in the column pass: gridDim.y detections at a time)
    if ((int)blockIdx.y >= nls) return;
    const int W = a.W, H = a.W, nb = (W + RC_BAND - 1) / RC_BAND;
    const int r0 = blockIdx.x * RR_ROWS;
    for (int ls = blockIdx.y; ls < nls; ls += (int)gridDim.y) {
    double *S = a.S + (int64_t)ls * a.SP * W;          // (rows are 128-byte aligned: SP is a multiple of 16)
    const double *T = a.colT + ((int64_t)ls * nb + r0 / RC_BAND) * W;     // column sums of the bands above this one
    // a lane moves PAIRS of columns (16-byte loads and stores: 18 memory instructions per tile and lane, so that RR_DEPTH - 1 tiles in
    // flight stay below the 64 a wave can have outstanding): 128 lanes per row, 2 rows per instruction of the workgroup
    const int lc = 2 * (threadIdx.x & 127), lr = threadIdx.x >> 7;
    const int ntiles = (W + RR_COLS - 1) / RR_COLS;
    double2 reg[RR_DEPTH][8], off[RR_DEPTH];
    // (loads are unconditional, from clamped addresses, and what lies outside the image is zeroed when the tile is USED: a conditional load
    // is a branch whose join waits for the data - the prefetch would be gone)
    const int ce = (W - 1) & ~1;                                          // the last even column: ce + 1 < SP
    auto fetch = [&](double2 (&x)[8], double2 &o, int i) {
        const int c = i * RR_COLS + lc;
        o.x = T[min(c, W - 1)];
        o.y = T[min(c + 1, W - 1)];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = *reinterpret_cast<const double2 *>(S + (int64_t)min(r0 + 2 * k + lr, H - 1) * a.SP + min(c, ce));
    };
#pragma unroll
    for (int d = 0; d < RR_DEPTH; d++) fetch(reg[d], off[d], min(d, ntiles - 1));
    double acc = 0;
    // tile i out of its registers, the tile RR_DEPTH steps later into them.  Nothing between two uses of a register set is conditional:
    // the compiler's wait for "these loads" counts the memory instructions that are CERTAIN to have followed them - behind `if`s, none
    auto step = [&](double2 (&x)[8], double2 &o, int i) {
        const int c = i * RR_COLS + lc;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool rin = r0 + 2 * k + lr < H;
            *reinterpret_cast<double2 *>(&tile[2 * k + lr][lc]) =
                make_double2(rin && c < W ? __dadd_rn(x[k].x, o.x) : 0.0, rin && c + 1 < W ? __dadd_rn(x[k].y, o.y) : 0.0);
        }
        fetch(x, o, min(i + RR_DEPTH, ntiles - 1));                       // (past the end: the last tile again, never used)
        __syncthreads();
#ifndef RR_NOCHAIN
        if (threadIdx.x < RR_ROWS) {
            // (columns past the image hold zeros: the chain runs through them, nothing of it is stored)
            double2 *row = reinterpret_cast<double2 *>(tile[threadIdx.x]);
            double2 y[8], z[8];
#pragma unroll
            for (int j = 0; j < 8; j++) y[j] = row[j];
            for (int j0 = 0; j0 < RR_COLS / 2; j0 += 16) {                // (two chunks of 16 columns per turn: y and z swap roles, no copies)
#pragma unroll
                for (int j = 0; j < 8; j++) z[j] = row[j0 + 8 + j];
#pragma unroll
                for (int j = 0; j < 8; j++) { acc = __dadd_rn(acc, y[j].x); y[j].x = acc; acc = __dadd_rn(acc, y[j].y); y[j].y = acc; }
#pragma unroll
                for (int j = 0; j < 8; j++) row[j0 + j] = y[j];
                const int jn = min(j0 + 16, RR_COLS / 2 - 8);
#pragma unroll
                for (int j = 0; j < 8; j++) y[j] = row[jn + j];
#pragma unroll
                for (int j = 0; j < 8; j++) { acc = __dadd_rn(acc, z[j].x); z[j].x = acc; acc = __dadd_rn(acc, z[j].y); z[j].y = acc; }
#pragma unroll
                for (int j = 0; j < 8; j++) row[j0 + 8 + j] = z[j];
            }
        }
#endif
        __syncthreads();
#ifndef RR_NOSTORE
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int r = r0 + 2 * k + lr;
            const double2 t2 = *reinterpret_cast<const double2 *>(&tile[2 * k + lr][lc]);
            if (r < H && c + 1 < W) *reinterpret_cast<double2 *>(S + (int64_t)r * a.SP + c) = t2;
            else if (r < H && c < W) S[(int64_t)r * a.SP + c] = t2.x;
        }
#endif
    };
    int i0 = 0;
    for (; i0 + RR_DEPTH <= ntiles; i0 += RR_DEPTH) {
#pragma unroll
        for (int d = 0; d < RR_DEPTH; d++) step(reg[d], off[d], i0 + d);
    }
#pragma unroll
    for (int d = 0; d < RR_DEPTH - 1; d++)
        if (i0 + d < ntiles) step(reg[d], off[d], i0 + d);
    __syncthreads();                                   // (the tile is the next detection's)
    }
}

// ------------------------------------------------------------------------------------------------ K1+K2 fused: one sweep
// One workgroup per detection walks the image in bands of RI_ROWS rows and, inside a band, in groups of RI_WAVES 64-column tiles.
// For every (band, group) "phase" i:
//   A(i)  column waves, lane = column: the band's RI_ROWS pixels of the column (computed from the polar record, see rt_pixel) are
//         added one after the other to the column's running sum (a register, carried from band to band) -> tile[row][column]
//   B(i)  the row wave, lane = row: the row's running sum walks through the group's tiles column by column (carried in a register
//         from group to group) - the second cumsum, in place
//   C(i)  every column wave writes its tile to the integral image
// The tiles are double-buffered, so that between two barriers the column waves run C(i-1) and A(i+1) while the row wave runs B(i):
// A is bound by load latency, B by the latency of 64 x RI_WAVES dependent float64 additions, and they hide each other.
// Both cumulative sums keep NumPy's sequential order; the float64 image is written ONCE (32.8 MB per detection instead of the
// 98.6 MB moved by the two-pass kernels above, which stay for small chunks - see rt_one_sweep - and for image sizes above 2048).
#ifndef RI_ROWS
#define RI_ROWS 16                          // (32 with ONE tile buffer: see RI_SINGLE_BUF)
#endif
#ifndef RI_WAVES
#define RI_WAVES 4
#endif
#define RI_GROUPS (2048 / (64 * RI_WAVES))
#ifndef RI_LDS_PAD
#define RI_LDS_PAD 0                        // (occupancy experiments: profiles/build_variant.py)
#endif
// (round 6, measured and dropped: ONE tile buffer with the taps of A(i + 1) beside B(i) - half the LDS, so four workgroups per CU - is
// bit-identical and slower: 6.90 ms per 512 detections against 6.58 at two workgroups per CU, 16.9 us per detection at three; eight column
// waves per workgroup on one buffer: 10.2 ms.  More waves per CU make this kernel slower, not faster - profiles/r06_detection_experiments.txt)
#ifndef RI_PHASE_SKIP
#define RI_PHASE_SKIP 1                     // walk the list of phases that matter (retrack_build_phases) instead of all bands x groups
#endif
#define RI_PHL_MAX 1024                     // phases of the largest image of the one-sweep kernel (128 bands x 8 groups)
#ifndef RI_BD
#define RI_BD 4                             // batches of eight columns the row wave reads ahead of its chain
#endif
#define RI_TP 66                            // tile pitch in doubles: even, so that the row wave moves two columns per LDS instruction (round 6)
// ONE tile buffer and bands of 32 rows (-DRI_SINGLE_BUF=1 -DRI_ROWS=32; round 6, late; bit-identical, NOT the default).  The row wave is
// what a phase waits for, and an LDS instruction costs it ~20 cycles whatever its active lanes (profiles/r06_detection_experiments.txt
// items 10, 11): with 32 rows per band it carries twice the pixels per instruction.  The tiles of a 32-row band take the LDS of two
// buffers of 16 rows, so there is one buffer: A2(i) (column sums -> tile) | barrier | B(i) beside A1(i + 1) (the taps of the next phase:
// registers only) | barrier | C(i) (tile -> HBM).  512 detections launched together (one round, the two workgroups of every CU in step):
// 6.46 -> 5.5 ms; 1 024: 12.5 -> 14.0; 1 900: 23.7 -> 22.9; 2 048: 24.8 -> 25.4; in the step (1 700-2 100 detections per launch): 23.44-23.50
// against 23.36-23.48 ms per launch, 70.4-70.7 against 70.4-70.6 ms per step - once the workgroups of a CU are out of step, nothing.
// (One buffer with 16-row bands: 6.75 ms per 512, slower than two.)
#ifndef RI_SINGLE_BUF
#define RI_SINGLE_BUF 0
#endif
#define RI_NBUF (RI_SINGLE_BUF ? 1 : 2)
#ifndef RI_ROUND
#define RI_ROUND 0                          // > 0: launch_retrack launches a chunk's integral images in rounds of that many workgroups
#endif
#define RI_LDS_BYTES (RI_NBUF * RI_WAVES * RI_ROWS * RI_TP * 8 + RI_LDS_PAD)
#ifndef RI_BOX
#define RI_BOX 2560                         // (round 6: 1536 -> 2560, the LDS that is left at two workgroups per CU: fewer patches on the gather path, -2 %)
#endif
#ifndef RI_PREFETCH
#define RI_PREFETCH 1                       // the box of the next phase is loaded one phase ahead (rt_integral_kernel)
#endif
//                        // bytes of polar footprint a column wave may stage (rt_integral_kernel)                      // two neighbouring codes in one (unaligned) load
// the polar footprint of every (band, group, wave) patch of the sweep depends on the sampling map only: computed once per engine
// (one wave per patch, the reduction the integral kernel used to redo for every detection and phase: 24 cross-lane exchanges)
__global__ __launch_bounds__(64) void rt_boxtab_kernel(const uint32_t *__restrict__ map, int W, int cols, uint32_t *__restrict__ boxtab)
{
    const int H = W, lane = threadIdx.x;
    const int idx = blockIdx.x, wave = idx % RI_WAVES, g = (idx / RI_WAVES) % RI_GROUPS, band = idx / (RI_WAVES * RI_GROUPS);
    const int c = min(g * 64 * RI_WAVES + 64 * wave + lane, W - 1);
    int mnx = 0x7fffffff, mxx = -1, mny = 0x7fffffff, mxy = -1;
    for (int k = 0; k < RI_ROWS; k++) {
        const uint32_t m = map[(int64_t)min(band * RI_ROWS + k, H - 1) * W + c];
        const int ix = m & 4095, iy = (m >> 12) & 1023;
        if (ix < cols) { mnx = min(mnx, ix); mxx = max(mxx, ix); mny = min(mny, iy); mxy = max(mxy, iy); }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        mnx = min(mnx, __shfl_xor(mnx, d)); mxx = max(mxx, __shfl_xor(mxx, d));
        mny = min(mny, __shfl_xor(mny, d)); mxy = max(mxy, __shfl_xor(mxy, d));
    }
    if (lane == 0) {
        boxtab[2 * idx] = mxx < 0 ? 0xffff0000u : ((uint32_t)mnx | ((uint32_t)mxx << 16));
        boxtab[2 * idx + 1] = mxx < 0 ? 0u : ((uint32_t)mny | ((uint32_t)mxy << 16));
    }
}

#ifdef RI_PROF
__device__ unsigned long long ri_prof[16];
extern "C" int roam_debug_integral_prof(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ri_prof), sizeof(ri_prof)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ri_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define RI_P(k) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); rip_[k] += tn_ - rit_; rit_ = tn_; }
#else
#define RI_P(k)
#endif
__global__ __launch_bounds__(64 * (RI_WAVES + 1)) void rt_integral_kernel(RtArgs a, int first, int ls0)
{
#ifdef RI_PROF
    unsigned long long rip_[8] = {0}, rit_ = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __align__(16) double ri_lds[];
    const int ls = ls0 + (int)blockIdx.x, slot = first + ls;      // (ls0: rounds of a chunk launched one after the other, RI_ROUND)
    if (slot >= *a.rt_n || !rt_one_sweep(a, first) || a.fused) return;
    typedef double Tile[RI_ROWS][RI_TP];
    Tile *tiles = reinterpret_cast<Tile *>(ri_lds);                        // [2][RI_WAVES]
    const int W = a.W, H = a.W, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int nbands = (H + RI_ROWS - 1) / RI_ROWS;
    __shared__ float lut[256];
    __shared__ __align__(4) uint8_t box[RI_WAVES][RI_BOX];
    // the phases to walk: band | group << 8 | (tiles the determinant kernel reads, one bit per column wave) << 12, in sweep order
    // (retrack_build_phases: a phase is left out when nothing in it is lit and its row sums are either still zero or never read again)
    __shared__ uint16_t phl[RI_PHL_MAX];
    if (t < 256) lut[t] = rt_code_to_f32(t);
#if RI_PHASE_SKIP
    const int nph = (int)a.phlist[0];
    for (int i = t; i < nph; i += (int)blockDim.x) phl[i] = (uint16_t)a.phlist[1 + i];
#else
    const int nph = nbands * RI_GROUPS;
    for (int i = t; i < nph; i += (int)blockDim.x) phl[i] = (uint16_t)((i / RI_GROUPS) | ((i % RI_GROUPS) << 8) | (((1u << RI_WAVES) - 1u) << 12));
#endif
    __syncthreads();
    // (an entry is read once - one LDS read a phase, a phase ahead - and passed on as a scalar)
    auto ph_ent = [&](int i) { return i < nph ? (int)__builtin_amdgcn_readfirstlane((int)phl[i]) : 0; };
    auto ph_band = [](int e) { return e & 255; };
    auto ph_group = [](int e) { return (e >> 8) & 15; };
    if (wave < RI_WAVES) {
        // ------------------------------------------------------------------------------------ column waves: C(i-1), A(i+1)
        const uint8_t *p = a.pool + (int64_t)a.rt_scan[slot] * a.rec_bytes + a.payload_off;
        double *S = a.S + (int64_t)ls * a.SP * W;
        const int SP = a.SP;
        const int rows = a.rows, cols = a.cols, stride = a.stride;
        double acc[RI_GROUPS];                                             // the running sums of this thread's columns
#pragma unroll
        for (int g = 0; g < RI_GROUPS; g++) acc[g] = 0.0;
        uint32_t m[RI_ROWS];                                               // the map words of the NEXT A, in flight
        uint32_t ext0 = 0, ext1 = 0;                                       // the polar footprint (boxtab) of the patch of the A after the next, in flight
        uint32_t cur0 = 0xffff0000u, cur1 = 0;                             // ... of the next A (wave-uniform)
        uint32_t raw[8];                                                   // the first eight pieces of the next A's box, in flight (RI_PREFETCH)
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        auto fetch = [&](int band, int g) {
            const int c = min(g * 64 * RI_WAVES + 64 * wave + lane, W - 1);
#pragma unroll
            for (int k = 0; k < RI_ROWS; k++) m[k] = a.map[(int64_t)min(band * RI_ROWS + k, H - 1) * W + c];
        };
        auto fetch_ext = [&](int i) {
            const uint32_t *bt = a.boxtab + 2 * (i * RI_WAVES + wave_u);
            ext0 = bt[0]; ext1 = bt[1];
        };
        // geometry of a patch's polar box out of its table entry (all wave-uniform)
        struct Box { int mnx, mxx, mny, mxy, bh, bp, nrg, ncb; bool staged, inside; };
        auto geom = [&](uint32_t e0, uint32_t e1) {
            Box b;
            b.mnx = e0 & 0xffff; b.mxx = (e0 >> 16) == 0xffff ? -1 : (int)(e0 >> 16); b.mny = e1 & 0xffff; b.mxy = e1 >> 16;
            const int bw = b.mxx - b.mnx + 2;
            b.bh = b.mxy - b.mny + 2; b.bp = (bw + 3) & ~3;
            b.nrg = (b.bh + 3) >> 2; b.ncb = (b.bp + 63) >> 6;             // pieces of 4 polar rows x 64 bytes, one load instruction each
            b.staged = b.mxx >= 0 && b.bp * b.bh <= RI_BOX && cols >= 4;
            // the box lies inside the scan with a margin: no azimuth wrap, no bin past the last one
            b.inside = b.staged && b.mny >= 1 && b.mny - 1 + 4 * b.nrg <= rows && b.mnx + 64 * b.ncb <= cols;
            return b;
        };
        const int sub = lane >> 4, c4 = (lane & 15) * 4;
        // pieces q0 .. q0 + 7 of an `inside` box: a piece is "uniform base + the lane's own offset" for the load and for the LDS store
        auto load_pieces = [&](const Box &b, int q0) {
            const uint8_t *pl = p + (sub * stride + c4);
            const int npiece = b.nrg * b.ncb;
            int kg = (q0 / b.ncb) * 4, cb = (q0 % b.ncb) * 64;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (q0 + u < npiece) raw[u] = reinterpret_cast<const RtU32 *>(pl + ((b.mny - 1 + kg) * stride + b.mnx + cb))->v;
                cb += 64;
                if (cb >= b.bp) { cb = 0; kg += 4; }
            }
        };
        auto store_pieces = [&](const Box &b, int q0, uint8_t *bx) {
            uint8_t *bl = bx + (sub * b.bp + c4);
            const int npiece = b.nrg * b.ncb;
            int kg = (q0 / b.ncb) * 4, cb = (q0 % b.ncb) * 64;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (q0 + u < npiece && sub < b.bh - kg && c4 < b.bp - cb) *reinterpret_cast<uint32_t *>(bl + (kg * b.bp + cb)) = raw[u];
                cb += 64;
                if (cb >= b.bp) { cb = 0; kg += 4; }
            }
        };
        // The first eight pieces of the NEXT A's box leave one phase ahead (its extent came out of the table a phase before that): their
        // HBM / L2 round trip - 37 % of a column wave's time when the loads were issued inside the phase that needs them (s_memtime) -
        // passes while the row wave works on the tile in between
        auto prefetch_box = [&]() {
            cur0 = __builtin_amdgcn_readfirstlane(ext0); cur1 = __builtin_amdgcn_readfirstlane(ext1);
#if RI_PREFETCH
            const Box b = geom(cur0, cur1);
            if (b.inside) load_pieces(b, 0);
#endif
        };
        float v[RI_ROWS];                                                  // the pixels of the phase A1 has prepared for A2
        auto A1 = [&](int i, int e_n1, int e_n2) {                         // phase i of the list: the patch's pixels -> v[]; e_n1 / e_n2: the entries of the next two phases
            // The polar footprint of the wave's 64 x 16 pixel patch is a small box (range span x azimuth span, a few hundred bytes):
            // it is copied into LDS with a handful of coalesced row loads (16 lanes per polar row, four rows per instruction) and
            // the 4 taps per pixel become LDS byte reads; per-lane byte gathers from global memory (two 16-bit loads per pixel,
            // 32 wave-level gathers per phase) kept the texture addresser busy for 8 of this kernel's 19 us.  Patches whose box
            // does not fit (next to the image centre, across the 0 / 2 pi seam) gather as before.
            // extent of the patch's polar footprint: out of the table (the same for every detection), as wave-uniform values
            const Box b = geom(cur0, cur1);
            const int mnx = b.mnx, mny = b.mny, bh = b.bh, bp = b.bp;
            if (b.mxx < 0) {
#pragma unroll
                for (int k = 0; k < RI_ROWS; k++) v[k] = 0.f;              // beyond the maximum range
            } else if (b.staged) {
                uint8_t *bx = box[wave];
                if (b.inside) {
                    // (the general form below spends ~20 instructions per piece on clamps and addresses: a sixth of this kernel)
                    const int npiece = b.nrg * b.ncb;
                    for (int q0 = 0; q0 < npiece; q0 += 8) {
                        if (q0 > 0 || !RI_PREFETCH) load_pieces(b, q0);
                        store_pieces(b, q0, bx);
                    }
                } else
                for (int kg = 0; kg < bh; kg += 4) {
                    const int kk = kg + sub;
                    int r = mny + kk - 1;
                    if (r < 0) r += rows; else if (r >= rows) r -= rows;
                    for (int cb = 0; cb < bp; cb += 64) {
                        const int cc = cb + c4;
                        if (kk < bh && cc < bp) {
                            // bytes beyond the scan's last range bin read as zero: the load is moved back inside the row and shifted
                            const int x0 = mnx + cc, xl = min(x0, cols - 4), sh = 8 * (x0 - xl);
                            uint32_t rw = 0;
                            if (sh < 32) rw = reinterpret_cast<const RtU32 *>(p + r * stride + xl)->v >> sh;
                            *reinterpret_cast<uint32_t *>(bx + kk * bp + cc) = rw;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < RI_ROWS; k++) {
                    const uint32_t mk = m[k];
                    const int ix = mk & 4095, iy = (mk >> 12) & 1023;
                    float r_ = 0.f;
                    if (ix < cols) {
                        const float wx1 = __fmul_rn((float)((mk >> 22) & 31), 1.f / 32.f), wx0 = __fsub_rn(1.f, wx1);
                        const float wy1 = __fmul_rn((float)(mk >> 27), 1.f / 32.f), wy0 = __fsub_rn(1.f, wy1);
                        const uint8_t *q = bx + (iy - mny) * bp + (ix - mnx);
#ifdef RI_TAP_VALU
                        // (experiment: the float32-only decode of warp.hip instead of the table - three VALU instructions against one LDS read)
                        auto dec = [](uint32_t k) { const float kf = (float)k; return __fmaf_rn(kf, 0x1.0101020000000p-8f, __fmul_rn(kf, -0x1.fdfdfe0000000p-33f)); };
                        const float s00 = dec(q[0]), s01 = dec(q[1]), s10 = dec(q[bp]), s11 = dec(q[bp + 1]);
#elif defined(RI_TAP16)
                        // (experiment: the two neighbouring codes of a polar row in ONE 16-bit LDS read at any alignment)
                        const uint32_t w0 = reinterpret_cast<const RtU16 *>(q)->v, w1 = reinterpret_cast<const RtU16 *>(q + bp)->v;
                        const float s00 = lut[w0 & 255], s01 = lut[w0 >> 8], s10 = lut[w1 & 255], s11 = lut[w1 >> 8];
#else
                        const float s00 = lut[q[0]], s01 = lut[q[1]], s10 = lut[q[bp]], s11 = lut[q[bp + 1]];   // lut[0] = 0: bins past the scan
#endif
                        r_ = __fmul_rn(s00, __fmul_rn(wy0, wx0));
                        r_ = __fadd_rn(r_, __fmul_rn(s01, __fmul_rn(wy0, wx1)));
                        r_ = __fadd_rn(r_, __fmul_rn(s10, __fmul_rn(wy1, wx0)));
                        r_ = __fadd_rn(r_, __fmul_rn(s11, __fmul_rn(wy1, wx1)));
                    }
                    v[k] = r_;
                }
            } else {
#pragma unroll
                for (int k = 0; k < RI_ROWS; k++) v[k] = rt_pixel(m[k], p, rows, cols, stride, lut);
            }
            // the next phase's map words leave now; they land while the row wave works
            if (i + 1 < nph) {
                fetch(ph_band(e_n1), ph_group(e_n1));
                prefetch_box();
                if (i + 2 < nph) fetch_ext(ph_band(e_n2) * RI_GROUPS + ph_group(e_n2));
            }
        };
        int gcur = 0;                                                      // the group whose running sums sit in acc[0]
        auto A2 = [&](int i, int band, int g) {
            const int c = g * 64 * RI_WAVES + 64 * wave + lane;
            Tile &tl = tiles[(RI_SINGLE_BUF ? 0 : (i & 1)) * RI_WAVES + wave];
            {
                // acc[0] is always the running sum of the CURRENT group's column: the array is rotated by one after every phase (8
                // register moves; a group-indexed array was kept in scratch memory by the compiler: 16 MB of extra HBM writes per
                // detection) - and by as many groups as the phase list leaves out in between (their pixels are dark: sums unchanged)
                while (gcur != g) {
                    const double s0 = acc[0];
#pragma unroll
                    for (int q = 0; q + 1 < RI_GROUPS; q++) acc[q] = acc[q + 1];
                    acc[RI_GROUPS - 1] = s0;
                    gcur = (gcur + 1) % RI_GROUPS;
                }
                double s = acc[0];
                if (c < W) {
#pragma unroll
                    for (int k = 0; k < RI_ROWS; k++) {
                        if (band * RI_ROWS + k < H) s = __dadd_rn(s, (double)v[k]);
                        tl[k][lane] = s;
                    }
                }
#pragma unroll
                for (int q = 0; q + 1 < RI_GROUPS; q++) acc[q] = acc[q + 1];
                acc[RI_GROUPS - 1] = s;
                gcur = (gcur + 1) % RI_GROUPS;
            }
        };
        auto C = [&](int i, int e) {
            const int band = ph_band(e), g = ph_group(e);
            const int c = g * 64 * RI_WAVES + 64 * wave + lane;
            const Tile &tl = tiles[(RI_SINGLE_BUF ? 0 : (i & 1)) * RI_WAVES + wave];
            const bool wanted = ((e >> (12 + wave_u)) & 1) != 0;              // does anything read this tile?
            if (c < W && wanted) {
                double *q = S + (int64_t)band * RI_ROWS * SP + c;
                const int nk = min(RI_ROWS, H - band * RI_ROWS);
                if (nk == RI_ROWS) {
#pragma unroll
                    for (int k = 0; k < RI_ROWS; k++) q[(int64_t)k * SP] = tl[k][lane];
                } else
                    for (int k = 0; k < nk; k++) q[(int64_t)k * SP] = tl[k][lane];
            }
        };
        // entries of phases i - 1 .. i + 3 as scalars
        int e_m1 = 0, e_0 = ph_ent(0), e_1 = ph_ent(1), e_2 = ph_ent(2), e_3 = ph_ent(3);
        if (nph > 0) {
            fetch(ph_band(e_0), ph_group(e_0));
            fetch_ext(ph_band(e_0) * RI_GROUPS + ph_group(e_0));
            prefetch_box();
            if (nph > 1) fetch_ext(ph_band(e_1) * RI_GROUPS + ph_group(e_1));
            A1(0, e_1, e_2);
#if RI_SINGLE_BUF
#pragma unroll 1
            for (int i = 0; i < nph; i++) {
                const int e_4 = ph_ent(i + 4);
                A2(i, ph_band(e_0), ph_group(e_0));
                __syncthreads();                                           // tile i is complete: B(i) runs ..
                if (i + 1 < nph) A1(i + 1, e_2, e_3);                      // .. beside the taps of phase i + 1 (registers only)
                __syncthreads();                                           // B(i) is complete
                C(i, e_0);
                e_m1 = e_0; e_0 = e_1; e_1 = e_2; e_2 = e_3; e_3 = e_4;
            }
        }
#else
            A2(0, ph_band(e_0), ph_group(e_0));
#pragma unroll 1
            for (int i = 0; i < nph; i++) {
                const int e_4 = ph_ent(i + 4);                             // (lands while this phase runs)
                RI_P(0)
                __syncthreads();                                           // A(i) and B(i-1) are complete
                RI_P(1)
                if (i > 0) C(i - 1, e_m1);
                RI_P(2)
                if (i + 1 < nph) { A1(i + 1, e_2, e_3); RI_P(3) A2(i + 1, ph_band(e_1), ph_group(e_1)); }
                RI_P(4)
                e_m1 = e_0; e_0 = e_1; e_1 = e_2; e_2 = e_3; e_3 = e_4;
            }
        }
        __syncthreads();
        if (nph > 0) C(nph - 1, e_m1);
#endif
    } else {
        // ------------------------------------------------------------------------------------ the row wave: B(i)
        // (s_setprio 3 for this wave - the chain a phase waits for - moves the wait from the column waves' barrier to their taps: the row wave
        // busy 80 -> 70 % of a phase, the column waves' A1 55 -> 67 %, the kernel 6.44 -> 6.43 ms per 512: dropped)
        double carry = 0.0;                                                // running sum of row band * RI_ROWS + lane
        int pband = -1;
        int e_nx = ph_ent(0);
        for (int i = 0; i < nph; i++) {
            const int band = ph_band(e_nx), g = ph_group(e_nx);
            e_nx = ph_ent(i + 1);
            if (band != pband) { carry = 0.0; pband = band; }              // (the phases a band leaves out on its left have sums of zero)
            const bool live = lane < RI_ROWS && band * RI_ROWS + lane < H;
            RI_P(5)
            __syncthreads();
            RI_P(6)
            if (live) {
                const int C0 = g * 64 * RI_WAVES, ncols = min(64 * RI_WAVES, W - C0);
                Tile *tg = tiles + (RI_SINGLE_BUF ? 0 : (i & 1)) * RI_WAVES;
                int j = 0;
                if (ncols >= 16) {
                    // batches of eight columns (eight dependent float64 additions), 16-byte LDS accesses (two columns per instruction)
                    auto rd = [&](double(&x)[8], int jj) {                 // (a batch never straddles two tiles: 64 = 8 x 8)
                        const double2 *q = reinterpret_cast<const double2 *>(&tg[jj >> 6][lane][jj & 63]);
#pragma unroll
                        for (int u = 0; u < 4; u++) { const double2 v2 = q[u]; x[2 * u] = v2.x; x[2 * u + 1] = v2.y; }
                    };
                    auto chain_wr = [&](double(&x)[8], int jj) {
#pragma unroll
                        for (int u = 0; u < 8; u++) { carry = __dadd_rn(carry, x[u]); x[u] = carry; }
                        double2 *q = reinterpret_cast<double2 *>(&tg[jj >> 6][lane][jj & 63]);
#pragma unroll
                        for (int u = 0; u < 4; u++) q[u] = make_double2(x[2 * u], x[2 * u + 1]);
                    };
                    // RI_BD batches of eight columns are in flight ahead of the chain: beside eight column waves' tap reads an LDS read
                    // takes several hundred cycles to come back, and one batch ahead the chain waited for it at every batch
                    // (56 cycles per column, the row wave busy 87 % of a phase: profiles/ri_prof.py)
                    double xr[RI_BD][8];
#pragma unroll
                    for (int u = 0; u < RI_BD; u++)
                        if (8 * u + 8 <= ncols) rd(xr[u], 8 * u);
                    for (; j + 8 * RI_BD <= ncols; j += 8 * RI_BD) {
#pragma unroll
                        for (int u = 0; u < RI_BD; u++) {
                            chain_wr(xr[u], j + 8 * u);
                            if (j + 8 * (RI_BD + u) + 8 <= ncols) rd(xr[u], j + 8 * (RI_BD + u));
                        }
                    }
                    // (what is left of the group - fewer than RI_BD batches - is in the registers already)
#pragma unroll
                    for (int u = 0; u < RI_BD; u++)
                        if (j + 8 <= ncols) { chain_wr(xr[u], j); j += 8; }
                }
                for (; j < ncols; j++) {
                    double *q = &tg[j >> 6][lane][j & 63];
                    carry = __dadd_rn(carry, *q);
                    *q = carry;
                }
            }
#if RI_SINGLE_BUF
            __syncthreads();                                               // (outside the lanes' branch: one barrier per wave)
#endif
        }
#if !RI_SINGLE_BUF
        __syncthreads();
#endif
    }
#ifdef RI_PROF
    if (lane == 0 && (wave == 0 || wave == RI_WAVES)) for (int k = 0; k < 8; k++) atomicAdd(&ri_prof[k + (wave == 0 ? 0 : 8)], rip_[k]);
#endif
}

// A maximum goes straight onto the detection's candidate list, in whatever order the workgroups get there; rt_emit_kernel sorts the
// list into the reference's (row, column, layer) order.  (Round 2's first version wrote a 1-byte mask per pixel plus row counts
// and had the emission kernel scan the mask and compute every candidate's determinant again from global memory: 4 MB more
// traffic per detection and 1.1-1.7 ms of latency per chunk.)
__device__ __forceinline__ void rt_push_maxima(const RtArgs &a, int ls, int r, int c, uint32_t bits, double v0, double v1)
{
#pragma unroll
    for (int l = 0; l < 2; l++)
        if ((bits >> l) & 1u) {
            const int o = atomicAdd(&a.cand_n[ls], 1);
            if (o < BP_MAX_PTS) {
                a.cand_rc[(int64_t)ls * BP_MAX_PTS + o] = ((uint32_t)r << 16) | ((uint32_t)c << 2) | (uint32_t)(l + 1);
                a.cand_val[(int64_t)ls * BP_MAX_PTS + o] = l ? v1 : v0;
            }
        }
}

// ------------------------------------------------------------------------------------------------ K3: determinants + maxima (strip march)
// Box sizes (15, 30) = int(3 sigma) of the engine's detector parameters, compile-time.  Round 2's kernel staged a 62 x 94 block of the
// integral image per 30 x 62 outputs: every byte of the image was fetched 3.1 times and the L2 captured none of it (PMC: 53.7 GB per
// 512 detections against 16.8 GB algorithmic).  Here a workgroup of 8 waves owns a column STRIP of 62 outputs (SD_HALVES = 2: 16
// waves, 126 outputs) and marches DOWN the image: the rows of the integral image live in an LDS ring of 64 rows x 94 columns, every
// step loads the SD_T NEW rows only (prefetched into registers three steps ahead) and computes SD_T x 64 determinant positions, so
// a byte is fetched 94 / 62 = 1.5 times before L2 and 1.36 times from HBM (PMC; SD_HALVES = 2: 1.25 / 1.06, but one workgroup per
// CU and 10 % slower) and the vertical halos (box rows and the 3 x 3 x 3 maxima) cost nothing: the maxima of a step's last row are
// decided one step later from a two-row seam kept with the per-position maxima.
//   * ONE barrier per step: the rows step t + 1 needs are written into ring slots that step t does not read (64 slots, 46 live
//     rows, 16 new ones), the per-position maxima are double-buffered - staging, the maxima of step t - 1 and the determinants of
//     step t run between the same two barriers, on different waves at different times.
//   * ring addressing without arithmetic: the step loop is unrolled by four, so the ring row of (step phase, position row, box
//     offset) is a compile-time constant; the wave's own row term (0..7) sits in the base register, and rows 0..7 of the ring are
//     stored twice (also as rows 64..71) so that "constant + wave" never wraps.  DS offsets are 16 bits: two base registers per
//     column (ring rows 0..35 / 36..71).
//   * columns are clipped like skimage's _integ ONCE per thread: the twelve clipped corner columns of the dxx / dyy boxes are byte
//     offsets in registers, so the strips along the left / right image border run the same code as interior ones; rows need clipping
//     in the first and the last steps only (a variant with computed row offsets).  The dxy boxes (where dxx * dyy can pass the
//     threshold: ~4 % of the wave-rows) compute their addresses on the fly.
//   * the 16 boxes of a thread and step are ONE stream of 8 pairs, the reads of pair i + 1 issued before the arithmetic of pair i
//     (sd_tile_fast); max(0, box) is the clamp modifier of the box's last subtraction (the ring holds the image scaled by 2^-10).
//   * workgroup -> strip mapping is XCD-aware: the strips of a detection are consecutive workgroups of ONE XCD, started together
//     and marching in step, so that the cache lines neighbouring strips share come from that XCD's L2 (hit rate 25 %).
//   * a step whose window lies beyond the maximum range - the corners of the image - skips its boxes, and the blocks only such steps
//     would read are neither loaded nor staged (rt_darktab_kernel: a bit table per strip, geometry only).
#define SD_T 16                             // position rows per step
#define SD_RING 64
#define SD_DUP 8
#ifndef SD_HALVES
#define SD_HALVES 1                         // 64-column groups per workgroup (8 waves each); 1: two workgroups per CU
#endif
#define SD_PC (64 * SD_HALVES)              // position columns per strip
#define SD_OUT (SD_PC - 2)
#define SD_HL 14                            // lowest / highest box offset of size 30
#define SD_HR 16
#define SD_BC (SD_PC + SD_HL + SD_HR)       // staged columns (158 | 94)
#define SD_BP ((SD_BC + 15) / 16 * 16)      // ring pitch in doubles (160 | 96)
#define SD_PITCHB (SD_BP * 8)
#define SD_SPLIT 36
#define SD_THREADS (512 * SD_HALVES)
#define SD_RING_BYTES ((SD_RING + SD_DUP) * SD_PITCHB)
#define SD_M2_ROWS (SD_T + 2)
#ifndef SD_LDS_PAD
#define SD_LDS_PAD 0
#endif
#define SD_LDS_BYTES (SD_RING_BYTES + 2 * SD_M2_ROWS * SD_PC * 8 + SD_LDS_PAD)
#define SD_NST ((SD_T * SD_BP + SD_THREADS - 1) / SD_THREADS)   // staged elements per thread and step (3)
extern __shared__ __align__(16) char sd_smem[];

// The ring holds the integral image scaled by 2^-10 (an exact operation that commutes with every rounding below), so that a box sum -
// at most 30 x 30 pixels of at most 1.0 - stays below 1 and skimage's max(0, sum) is the CLAMP output modifier of the box's last
// subtraction instead of a v_max_f64 of its own (8 of a position's 46 vector instructions); the 1 / size^2 factors carry the 2^10.
#define SD_SCALE 0.0009765625
#define SD_UNSCALE 1024.0
__device__ __forceinline__ double sd_box(double a, double d, double b, double c)
{
    const double t = __dsub_rn(__dadd_rn(a, d), b);
    double r;
    asm("v_add_f64 %0, %1, -%2 clamp" : "=v"(r) : "v"(t), "v"(c));
    return r;
}
// max of two determinants (never NaN): one v_max_f64 - fmax() canonicalises both operands first (three instructions)
__device__ __forceinline__ double sd_max(double x, double y)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
template <int SIZE> __device__ __forceinline__ double sd_wi() { return __dmul_rn(__ddiv_rn(__ddiv_rn(1.0, (double)SIZE), (double)SIZE), SD_UNSCALE); }

__device__ __forceinline__ double sd_ldr(const uint32_t (&ca)[12], const uint32_t (&cb)[12], int row, int j)
{
    const int rr = ((row % SD_RING) + SD_RING) % SD_RING;                  // (constants after unrolling)
    return rr < SD_SPLIT ? *reinterpret_cast<const double *>(sd_smem + (ca[j] + rr * SD_PITCHB))
                         : *reinterpret_cast<const double *>(sd_smem + (cb[j] + (rr - SD_SPLIT) * SD_PITCHB));
}

// dxx * dyy (hessian_det_pruned's first product) of a thread's two positions in a step whose box rows need no clipping; PH = step & 3.
// The 16 boxes are ONE stream of 8 pairs (the same box of both positions): the eight corner reads of pair i + 1 are issued before the
// arithmetic of pair i (the compiler's own order waited for every box's reads before it issued the next four), and the two
// positions' dependent float64 chains alternate instruction by instruction, so that a wave that is alone on its SIMD - the tail of
// every step: the hardware favours the oldest wave, the youngest finish last - still issues back to back.
template <int PH>
__device__ __forceinline__ void sd_tile_fast(const uint32_t (&ca)[12], const uint32_t (&cb)[12], double (&d0)[SD_T / 8], double (&d1)[SD_T / 8])
{
    static_assert(SD_T == 16, "two positions per thread and step");
    double v[2][2][4];                                                     // [pair slot][position][corner]
    auto issue = [&](int i, double(&o)[2][4]) {
        const int L = (i >> 2) & 1, q = i & 3;                             // q: xx-mid, xx-side, yy-mid, yy-side (hessian_box 4..7)
        const int SIZE = L ? 30 : 15, s2 = (SIZE - 1) / 2, s3 = SIZE / 3;
        const int ra = q < 2 ? -s3 + 1 : (q == 2 ? -s2 : -(s3 / 2)), rb = ra + (q < 2 ? 2 * s3 - 1 : (q == 2 ? SIZE : s3));
        const int ja = 6 * L + (q == 0 ? 0 : (q == 1 ? 2 : 4)), jb = ja + 1;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int R0 = SD_T * PH + 8 * k;
            o[k][0] = sd_ldr(ca, cb, R0 + ra, ja); o[k][1] = sd_ldr(ca, cb, R0 + rb, jb);
            o[k][2] = sd_ldr(ca, cb, R0 + ra, jb); o[k][3] = sd_ldr(ca, cb, R0 + rb, ja);
        }
    };
    issue(0, v[0]);
    double mid[2] = {0.0, 0.0}, dxx[2] = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i + 1 < 8) issue(i + 1, v[(i + 1) & 1]);
        const int L = (i >> 2) & 1, q = i & 3;
        const double(&x)[2][4] = v[i & 1];
        double t0 = __dadd_rn(x[0][0], x[0][1]), t1 = __dadd_rn(x[1][0], x[1][1]);
        t0 = __dsub_rn(t0, x[0][2]); t1 = __dsub_rn(t1, x[1][2]);
        double b0, b1;
        asm("v_add_f64 %0, %1, -%2 clamp" : "=v"(b0) : "v"(t0), "v"(x[0][3]));
        asm("v_add_f64 %0, %1, -%2 clamp" : "=v"(b1) : "v"(t1), "v"(x[1][3]));
        if (q == 0 || q == 2) { mid[0] = b0; mid[1] = b1; }
        else {
            const double w_i = L ? sd_wi<30>() : sd_wi<15>();
            double m0 = __dmul_rn(3.0, b0), m1 = __dmul_rn(3.0, b1);
            double e0 = __dsub_rn(mid[0], m0), e1 = __dsub_rn(mid[1], m1);
            e0 = __dmul_rn(-e0, w_i); e1 = __dmul_rn(-e1, w_i);
            if (q == 1) { dxx[0] = e0; dxx[1] = e1; }
            else if (L) { d1[0] = __dmul_rn(dxx[0], e0); d1[1] = __dmul_rn(dxx[1], e1); }
            else { d0[0] = __dmul_rn(dxx[0], e0); d0[1] = __dmul_rn(dxx[1], e1); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the same product with the box rows clipped to the image (first / last steps of a strip): r = image row of the position (wave-uniform)
template <int SIZE, int L>
__device__ __forceinline__ double sd_det_rows(const uint32_t (&ca)[12], int w8, int r, int H)
{
    constexpr int s2 = (SIZE - 1) / 2, s3 = SIZE / 3, w = SIZE, J = 6 * L;
    const double w_i = sd_wi<SIZE>();
    auto ld = [&](int row, int j) { return *reinterpret_cast<const double *>(sd_smem + (ca[j] + (uint32_t)(((row & (SD_RING - 1)) - w8) * SD_PITCHB))); };
    const int xa = clipi(r - s3 + 1, 0, H - 1), xb = clipi(xa + 2 * s3 - 1, 0, H - 1);
    double mid = sd_box(ld(xa, J + 0), ld(xb, J + 1), ld(xa, J + 1), ld(xb, J + 0));
    double side = sd_box(ld(xa, J + 2), ld(xb, J + 3), ld(xa, J + 3), ld(xb, J + 2));
    double dxx = __dsub_rn(mid, __dmul_rn(3.0, side));
    dxx = __dmul_rn(-dxx, w_i);
    const int ya = clipi(r - s2, 0, H - 1), yb = clipi(ya + w, 0, H - 1), za = clipi(r - s3 / 2, 0, H - 1), zb = clipi(za + s3, 0, H - 1);
    mid = sd_box(ld(ya, J + 4), ld(yb, J + 5), ld(ya, J + 5), ld(yb, J + 4));
    side = sd_box(ld(za, J + 4), ld(zb, J + 5), ld(za, J + 5), ld(zb, J + 4));
    double dyy = __dsub_rn(mid, __dmul_rn(3.0, side));
    dyy = __dmul_rn(-dyy, w_i);
    return __dmul_rn(dxx, dyy);
}

// the dxy term of a position whose dxx * dyy passes the threshold (rare): addresses computed on the fly, clipping included
template <int SIZE>
__device__ __forceinline__ double sd_dxy(double det, int r, int c, int H, int W, int cbase)
{
    constexpr int s3 = SIZE / 3;
    const double w_i = sd_wi<SIZE>();
    const int r0 = clipi(r - s3, 0, H - 1), r1 = clipi(r0 + s3, 0, H - 1), r2 = clipi(r + 1, 0, H - 1), r3 = clipi(r2 + s3, 0, H - 1);
    const int c0 = clipi(c - s3, 0, W - 1), c1 = clipi(c0 + s3, 0, W - 1), c2 = clipi(c + 1, 0, W - 1), c3 = clipi(c2 + s3, 0, W - 1);
    auto at = [&](int rr, int cc) { return *reinterpret_cast<const double *>(sd_smem + (((rr & (SD_RING - 1)) * SD_BP + (cc - cbase)) * 8)); };
    const double tl = sd_box(at(r0, c0), at(r1, c1), at(r0, c1), at(r1, c0));
    const double br = sd_box(at(r2, c2), at(r3, c3), at(r2, c3), at(r3, c2));
    const double bl = sd_box(at(r0, c2), at(r1, c3), at(r0, c3), at(r1, c2));
    const double tr = sd_box(at(r2, c0), at(r3, c1), at(r2, c1), at(r3, c0));
    double dxy = __dsub_rn(__dsub_rn(__dadd_rn(bl, tr), tl), br);
    dxy = __dmul_rn(-dxy, w_i);
    return __dsub_rn(det, __dmul_rn(0.81, __dmul_rn(dxy, dxy)));
}

// ---- which steps of a strip see nothing: geometry only, once per engine.
// The Cartesian pixels beyond the maximum range (sampling-map word with ix >= cols: 21 % of the image, its four corners) are zero whatever
// the scan holds.  A step whose whole window - position rows 16 t .. 16 t + 15 with their box rows -14 .. +16, the strip's position columns
// with their box columns -14 .. +16 - lies there has box sums of exactly nothing (up to the rounding of the integral image's cumulative
// sums, ~1e-9, against a threshold of 5e-4): its determinants can neither pass the threshold nor exceed a passing neighbour, they count as
// 0.  Such a step skips the boxes; a 16-row block of the integral image that only such steps would read (steps j - 1, j, j + 1 for
// block j) is not loaded.  The lit steps of a strip are ONE run (the range limit is a circle): the march starts just above it and stops
// just below - the dark steps outside cost a barrier and a ring fill each, 40 % of a lit step.  Per strip SD_DT_WORDS words: [0, 8) bit
// t = step t is dark, [8, 16) bit t = the block loaded AT step t (block t + 4) can be skipped, [16] / [17] = first / last lit step
// (nt / -1: none).
// (Round 3's first version tested the block's sum out of the ring in every wave and step - four LDS reads and a wait in front of every
// step's boxes - and loaded every block.)
#define SD_DT_WORDS 24
__global__ __launch_bounds__(256) void rt_darktab_kernel(const uint32_t *__restrict__ map, int W, int cols, uint32_t *__restrict__ tab)
{
    const int H = W, strip = blockIdx.x, t = blockIdx.y, c0 = strip * SD_OUT;
    const int ra = max(SD_T * t - SD_HL, 0), rb = min(SD_T * t + SD_T - 1 + SD_HR, H - 1);
    const int xa = max(c0 - 1 - SD_HL, 0), xb = min(c0 - 1 + SD_PC - 1 + SD_HR, W - 1);
    const int nx = xb - xa + 1, n = max(0, rb - ra + 1) * max(0, nx);
    int lit = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int r = ra + i / nx, c = xa + i % nx;
        if ((int)(map[(int64_t)r * W + c] & 4095u) < cols) lit = 1;
    }
    lit = __syncthreads_or(lit);
    if (threadIdx.x == 0 && !lit) atomicOr(&tab[strip * SD_DT_WORDS + (t >> 5)], 1u << (t & 31));
}
__global__ __launch_bounds__(256) void rt_darkskip_kernel(int nt, uint32_t *__restrict__ tab)
{
    uint32_t *T = tab + blockIdx.x * SD_DT_WORDS;
    const int t = threadIdx.x;
    auto dark = [&](int q) { return q < 0 || q >= nt || ((T[q >> 5] >> (q & 31)) & 1u); };     // steps that do not exist read nothing
    auto skip = [&](int j) { return dark(j - 1) && dark(j) && dark(j + 1); };
    if (skip(t + 4)) atomicOr(&T[8 + (t >> 5)], 1u << (t & 31));
    if (t == 0) {
        int first = nt, last = -1;
        for (int q = 0; q < nt; q++)
            if (!dark(q)) { if (first == nt) first = q; last = q; }
        T[16] = (uint32_t)first; T[17] = (uint32_t)last;
    }
}

template <int P> struct SdTag { static constexpr int value = P; };

__global__ __launch_bounds__(SD_THREADS) void rt_det_strip_kernel(RtArgs a, int first, int P, int nstrips)
{
    const int nact = min(P, max(0, *a.rt_n - first));
    if (a.fused && rt_one_sweep(a, first)) return;                          // rt_fused_kernel's chunk
    // XCD-aware order: workgroup id -> (XCD = id % 8, j = id / 8); XCD x owns the x-th contiguous eighth of the (detection, strip) list
    const int total = nact * nstrips, per = (total + 7) >> 3;
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int work = xcd * per + jq;
    if (jq >= per || work >= total) return;
    const int ls = work / nstrips, strip = work - ls * nstrips;
    const int W = a.W, H = a.W, SP = a.SP;
    const double thr = a.threshold;
    typedef const uint32_t __attribute__((address_space(4))) *SdConstPtr;     // constant address space + uniform address = scalar load
    const SdConstPtr dtab = (SdConstPtr)(uintptr_t)(a.darktab + strip * SD_DT_WORDS);
    uint32_t dk_w = 0, skf_w = 0;                                           // the words of the current 32 steps: dark / skip the load
    const double *__restrict__ S = a.S + (int64_t)ls * SP * W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), w8 = wave & 7, half = wave >> 3;   // half < SD_HALVES
    const int c0 = strip * SD_OUT, cbase = c0 - 1 - SD_HL;
    const int pc = half * 64 + lane, c = c0 - 1 + pc;
    const bool cvalid = c >= 0 && c < W;
    const bool wave_live = c0 - 1 + half * 64 < W;                          // some column of this wave lies inside the image
    // per-position maxima max(layer 15, layer 30), two buffers by step parity, SD_M2_ROWS x SD_PC each: rows 0, 1 = the seam (rows 14, 15
    // of the step before), row 2 + r = position row r.  mb = this thread's column in row 0 of buffer 0 (byte offset)
    constexpr int M2ROW = SD_PC * 8, M2BUF = SD_M2_ROWS * M2ROW;
    const uint32_t mb = (uint32_t)(SD_RING_BYTES + w8 * M2ROW + pc * 8);
    auto m2at = [&](int buf, int row, int dc) -> double & { return *reinterpret_cast<double *>(sd_smem + (mb + (uint32_t)(buf * M2BUF + (row - w8) * M2ROW + dc * 8))); };
    // the twelve clipped corner columns (skimage _integ: c' = clip(c + off), c'' = clip(c' + width)) as ring byte offsets
    uint32_t ca[12], cb[12];
    {
        constexpr int off[6][2] = {{-7, 15}, {-2, 5}, {-4, 9}, {-14, 30}, {-5, 10}, {-9, 19}};
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const int x0 = clipi(c + off[q][0], 0, W - 1), x1 = clipi(x0 + off[q][1], 0, W - 1);
            ca[2 * q] = (uint32_t)((w8 * SD_BP + (x0 - cbase)) * 8);
            ca[2 * q + 1] = (uint32_t)((w8 * SD_BP + (x1 - cbase)) * 8);
        }
#pragma unroll
        for (int q = 0; q < 12; q++) cb[q] = ca[q] + SD_SPLIT * SD_PITCHB;
    }
    // staging: the SD_T x SD_BP new elements of a step in linear order over the threads (element e = tid + SD_THREADS j: a wave reads runs
    // of 512 contiguous bytes); a thread's (row, column) pairs are fixed, the image base moves: loads are "uniform base + lane offset".
    // The rows of step t are loaded three steps ahead into one of two register sets (HBM latency is longer than a step)
    const bool strip_inside = cbase >= 0 && cbase + SD_BP <= W;
    uint32_t goff[SD_NST], loff[SD_NST];                                   // byte offsets: image (from row 16 t + 16, column cbase) / ring (from slot 16 q)
    int srow[SD_NST], scol[SD_NST];
#pragma unroll
    for (int j = 0; j < SD_NST; j++) {
        const int e = tid + SD_THREADS * j;
        srow[j] = e / SD_BP; scol[j] = e - srow[j] * SD_BP;
        goff[j] = (uint32_t)((srow[j] * SP + scol[j]) * 8);
        loff[j] = (uint32_t)(srow[j] * SD_PITCHB + scol[j] * 8);
    }
    double stage[2][SD_NST];
    const int nt = H / SD_T + 1;                                           // the last step's last row lies outside the image
    // the march covers the strip's lit steps [first, last] only: it starts on the multiple of four at or below first - 1 (a dark step, or
    // step 0: nothing above it is read; the per-position maxima of "the step before" start as zeros, which is what dark steps hold) and
    // ends with step last + 1, whose zeros close the maxima of step last
#ifdef SD_NO_TRIM
    const int t_first = 0, t_last = nt - 1;
#else
    const int t_first = (int)dtab[16], t_last = (int)dtab[17];
#endif
    if (t_first >= nt) return;                                             // nothing but the corners: no candidates (uniform: before any barrier)
    const int tb = t_first >= 1 ? ((t_first - 1) & ~3) : 0, te = min(nt, t_last + 2);
    int frow = SD_T * tb;                                                  // first image row of the next load: 16 t + 16, t = tb - 1, tb, ..
    // (skip: only dark steps would read the block - its loads are pointed at the strip's first block instead, lines this workgroup has
    // in its L1 since the prologue: the same instructions on both paths keep the compiler's vmcnt bookkeeping exact, a branch around
    // the loads made every later wait a wait for ALL outstanding loads and cost more than the traffic it saved)
    auto fetch = [&](double(&st)[SD_NST], bool skip = false) {
        const bool whole = frow >= 0 && frow + SD_T <= H && strip_inside;
        const char *base = reinterpret_cast<const char *>(S + (int64_t)((skip && whole) ? 0 : frow) * SP + cbase);       // (wave-uniform)
        if (whole) {
#pragma unroll
            for (int j = 0; j < SD_NST; j++)
                if (SD_THREADS * (j + 1) <= SD_T * SD_BP || tid + SD_THREADS * j < SD_T * SD_BP) st[j] = *reinterpret_cast<const double *>(base + goff[j]);
        } else {
#pragma unroll
            for (int j = 0; j < SD_NST; j++) {
                const int gr = frow + srow[j], col = cbase + scol[j];
                if (srow[j] < SD_T && gr >= 0 && gr < H && col >= 0 && col < W) st[j] = *reinterpret_cast<const double *>(base + goff[j]);
            }
        }
        frow += SD_T;
    };
    // ring slot of row 16 t + 16 + srow = 16 ((t + 1) & 3) + srow; slots 0..7 are mirrored at 64..71
    auto put = [&](int quarter, const double(&st)[SD_NST]) {
#pragma unroll
        for (int j = 0; j < SD_NST; j++)
            if (SD_THREADS * (j + 1) <= SD_T * SD_BP || tid + SD_THREADS * j < SD_T * SD_BP) {
                double *d = reinterpret_cast<double *>(sd_smem + (loff[j] + (uint32_t)(quarter * SD_T * SD_PITCHB)));
                const double v = __dmul_rn(st[j], SD_SCALE);
                *d = v;
                if (quarter == 0 && srow[j] < SD_DUP) d[SD_RING * SD_BP] = v;
            }
    };
    double d0[SD_T / 8], d1[SD_T / 8], p0 = 0.0, p1 = 0.0;                 // this step's determinants; the last row of the step before
    uint32_t cand = 0, pcand = 0;                                          // bit k: position k of the step holds a determinant above the threshold
#pragma unroll
    for (int k = 0; k < SD_T / 8; k++) { d0[k] = 0.0; d1[k] = 0.0; }
    // 3 x 3 x 3 maxima out of buffer `buf`: rr2 = row in the buffer (2 + position row; 1 = the seam row), r = image row
    auto decide = [&](int buf, int rr2, int r, double v0, double v1) {
        if (pc < 1 || pc > SD_OUT || c >= W || r >= H) return;
        double mx = m2at(buf, rr2, 0);
#pragma unroll
        for (int dr = -1; dr <= 1; dr++)
#pragma unroll
            for (int dc = -1; dc <= 1; dc++) { const double u = m2at(buf, rr2 + dr, dc); mx = u > mx ? u : mx; }
        const uint32_t bits = ((v0 > thr && !(mx > v0)) ? 1u : 0u) | ((v1 > thr && !(mx > v1)) ? 2u : 0u);
        if (bits) rt_push_maxima(a, first + ls, r, c, bits, v0, v1);
    };
    // maxima of step pt (rows 0 .. SD_T - 2) and of the last row of the step before it; everything they need is in step pt's buffer
    auto maxima = [&](int pt, int buf) {
        if (w8 == 7) {
            if (pcand) decide(buf, 1, SD_T * pt - 1, p0, p1);
            p0 = d0[SD_T / 8 - 1]; p1 = d1[SD_T / 8 - 1]; pcand = cand >> (SD_T / 8 - 1);
        }
        if (cand) {
#pragma unroll
            for (int k = 0; k < SD_T / 8; k++) {
                const int rr = 8 * k + w8;
                if (((cand >> k) & 1u) && rr < SD_T - 1) decide(buf, 2 + rr, SD_T * pt + rr, d0[k], d1[k]);
            }
        }
    };
    // prologue: per-position maxima cleared (rows above the image count as zero), rows 0..31 of the image, loads of steps 1 and 2 in flight
    for (int i = tid; i < 2 * SD_M2_ROWS * SD_PC; i += SD_THREADS) reinterpret_cast<double *>(sd_smem + SD_RING_BYTES)[i] = 0.0;
    fetch(stage[1]);
    fetch(stage[0]);
    put(0, stage[1]);
    put(1, stage[0]);
    fetch(stage[1]);
    fetch(stage[0]);
    __syncthreads();
    auto step = [&](auto tag, int t) {
        constexpr int PH = decltype(tag)::value, WB = PH & 1;
        // rows of step t + 1 (loaded two steps ago) into slots that step t does not read; then the loads of step t + 3 - unless only
        // dark steps would read them (rt_darktab_kernel)
        const uint32_t tb = 1u << (t & 31);
        put((PH + 2) & 3, stage[(PH + 1) & 1]);
        fetch(stage[(PH + 1) & 1], (skf_w & tb) != 0);
        if (t >= 1) maxima(t - 1, WB ^ 1);
        const int rbase = SD_T * t;
        const bool fast = t >= 1 && rbase + SD_T - 1 + SD_HR <= H - 1;
        const bool dark = (dk_w & tb) != 0;                                 // every box of every position of this step lies beyond the maximum range
        if (!wave_live || dark) {
#pragma unroll
            for (int k = 0; k < SD_T / 8; k++) { d0[k] = 0.0; d1[k] = 0.0; }
        } else if (fast) {
            sd_tile_fast<PH>(ca, cb, d0, d1);
        } else {
#pragma unroll
            for (int k = 0; k < SD_T / 8; k++) {
                const int r = rbase + 8 * k + w8;
                d0[k] = r < H ? sd_det_rows<15, 0>(ca, w8, r, H) : 0.0;
                d1[k] = r < H ? sd_det_rows<30, 1>(ca, w8, r, H) : 0.0;
            }
        }
        cand = 0;
#pragma unroll
        for (int k = 0; k < SD_T / 8; k++) {
            const int rr = 8 * k + w8, r = rbase + rr;
            double mx = sd_max(d0[k], d1[k]);
            if (mx > thr) {
                // (rare) the dxy term, hessian_det_pruned's second half.  A product that stays at or below the threshold can neither
                // pass nor exceed a passing neighbour, so it goes to the maxima buffer as it is
                if (d0[k] > thr) d0[k] = sd_dxy<15>(d0[k], r, c, H, W, cbase);
                if (d1[k] > thr) d1[k] = sd_dxy<30>(d1[k], r, c, H, W, cbase);
                if (!cvalid) { d0[k] = 0.0; d1[k] = 0.0; }                 // outside the image: nothing that could exceed a maximum
                mx = sd_max(d0[k], d1[k]);
                if (mx > thr) cand |= 1u << k;
            }
            if (k == SD_T / 8 - 1 && w8 >= 6) m2at(WB, rr - (SD_T - 2), 0) = m2at(WB ^ 1, 2 + rr, 0);   // the seam: the last two rows of the step before
            m2at(WB, 2 + rr, 0) = mx;
        }
        __syncthreads();
    };
    for (int t = tb; t < te; t += 4) {
        if ((t & 31) == 0 || t == tb) { dk_w = dtab[t >> 5]; skf_w = dtab[8 + (t >> 5)]; }
#ifdef SD_NO_LOADSKIP
        skf_w = 0;
#endif
#ifdef SD_NO_DARK
        dk_w = 0;
#endif
        step(SdTag<0>(), t);
        if (t + 1 < te) step(SdTag<1>(), t + 1);
        if (t + 2 < te) step(SdTag<2>(), t + 2);
        if (t + 3 < te) step(SdTag<3>(), t + 3);
    }
    maxima(te - 1, (te - 1) & 1);
}

#include "retrack_fused.inc"
#include "ssc_body.inc"

// ------------------------------------------------------------------------------------------------ K4: ordered candidates
// the candidates of a detection (at most BP_MAX_PTS are kept) sorted by (row, column, layer): keys are unique, so a key's rank is
// the number of smaller keys - every thread counts for its keys against the whole list in LDS (broadcast reads)
__global__ __launch_bounds__(256) void rt_emit_kernel(RtArgs a, int first)
{
    __shared__ uint32_t key[BP_MAX_PTS];
    __shared__ double val[BP_MAX_PTS];
    const int ls = blockIdx.x, slot = first + ls;
    if (slot >= *a.rt_n) return;
    const int t = threadIdx.x;
    const int n = min(a.cand_n[ls], BP_MAX_PTS);
    uint32_t *crc = a.cand_rc + (int64_t)ls * BP_MAX_PTS;
    double *cval = a.cand_val + (int64_t)ls * BP_MAX_PTS;
    for (int i = t; i < n; i += 256) { key[i] = crc[i]; val[i] = cval[i]; }
    __syncthreads();
    for (int i = t; i < n; i += 256) {
        const uint32_t k = key[i];
        int rank = 0;
        for (int j = 0; j < n; j++) rank += key[j] < k ? 1 : 0;
        crc[rank] = k;
        cval[rank] = val[i];
    }
}

// ------------------------------------------------------------------------------------------------ K5: blob bookkeeping
// Two capacity classes.  The tables of the FULL class (2048 candidates, 4914 pairs in the LDS set tables) take 64 KB: two workgroups - two
// detections - per CU, while a real or synthetic scan has 250-600 candidates and a few hundred to 1400 pairs.  The SMALL class (1024
// candidates, 1228 pairs: the set then never grows past 2048 entries) takes 33 KB: four detections per CU.  Every detection goes
// through the small kernel first; one that does not fit (more candidates, tree nodes or pairs) is left untouched and marked for the
// full kernel, which runs right after it and returns at once for everything else.
template <bool SMALL> struct RtBlobCap {
    static constexpr int NP = SMALL ? 1024 : BP_MAX_PTS;
    static constexpr int NNODE = SMALL ? 320 : BP_MAX_NODES;
    static constexpr int CAPA = 2048, CAPB = SMALL ? 4096 : 8192;          // set tables (entries); tabB also holds NP packed points / doubles
    static constexpr int LDS_PAIRS = SMALL ? 1228 : BP_LDS_PAIRS;          // below this count the set never outgrows the tables
    static constexpr int NPL = SMALL ? 1280 : 3328;                        // pairs kept in LDS (full: what the 64 KB of static LDS leave)
    static constexpr int NCL = CAPB / 2;                                   // overlapping pairs gathered into tabB (uint32)
};
#define RT_BLOBS_REDO (-1)                // kp_n of a detection the small kernel left to the full one
template <bool SMALL> struct RtBlobLds {
    typedef RtBlobCap<SMALL> CP;
    int16_t xy[2 * CP::NP];           // [row, col] in response order
    int16_t idx[CP::NP];              // cKDTree.indices, later the aquicksort permutation
    uint8_t lay[CP::NP];              // layer (1 | 2) in response order; 0 = pruned
    BpNode nodes[CP::NNODE];
    int st[3 * 256];
    int16_t nbox[CP::NNODE][4];       // query_pairs' tracker box of every node: the root's bounds cut by the split planes on the way down
                                      // ({min0, max0, min1, max1}; filled as the nodes are built)
    uint16_t tabA[CP::CAPA];
    alignas(8) uint16_t tabB[CP::CAPB];   // hash table of the set order; before that the packed points of the tree build (NP x 8 B)
    uint32_t ovbits[(CP::LDS_PAIRS + 31) / 32 + 1];
    uint32_t pl[CP::NPL];             // the pairs, when they fit: the sequential set-order pass reads them one by one
    int vals[8];
};

// ---- the large nodes of the k-d tree, built by the WHOLE wavefront (round 5).  bp_build_node - bounds, libstdc++'s introselect, scipy's
// partition pass - on one lane is a chain of dependent LDS reads: ~110 us for the 530-element root of a real frame, 55 for its children
// (profiles: ranking + tree build were 323 of the 640 us of a lone detection's bookkeeping).  Both partitions are two-pointer scans that swap
// the i-th misplaced element from the left with the i-th from the right until the pointers cross; elements between the pointers are never
// touched before the pointers get there, so the two ordered lists of misplaced POSITIONS can be taken from the array as it stands (ballot +
// prefix count per 64 positions), the number of swaps is the number of i with L[i] < R[i], and the swaps are independent of each other:
// the same array, element for element, as the sequential code (an element equal to the pivot sits in both lists; it stops the scan from
// whichever side reaches it first, exactly as there; checked against the sequential algorithms on 20 000 random arrays with heavy ties).  Median-of-three, the <= 3-element insertion sort and the rare heap-select
// fallback stay sequential (uniform values / lane 0).  Lp, Rp: scratch of (end - start) uint16 each.
#define RB_WAVE_MIN 100                     // nodes with more elements take this path (a level of eight 66-element nodes is faster lane by lane)
__device__ __forceinline__ int rb_key(const BpPt *pt, int i, int d) { return (int16_t)(pt[i].v >> (16 * d)); }

// unguarded Hoare partition of [first + 1, last) around piv = key(first) as std::__unguarded_partition leaves it; returns the cut
__device__ int rb_partition_hoare(BpPt *pt, int first, int last, int d, int lane, uint16_t *Lp, uint16_t *Rp)
{
    const int piv = rb_key(pt, first, d);
    const uint64_t below = (1ull << lane) - 1ull;
    int nL = 0, nR = 0;
    // one pass over [first, last): L = positions >= first + 1 with key >= piv, ascending; R = positions with key <= piv, stored ascending
    // too and read from its end (R[i] = Rp[nR - 1 - i])
    for (int c0 = first; c0 < last; c0 += 64) {
        const int pos = c0 + lane;
        const int k = pos < last ? rb_key(pt, pos, d) : 0;
        const bool ge = pos < last && pos > first && k >= piv, le = pos < last && k <= piv;
        const uint64_t bl = __ballot(ge), br = __ballot(le);
        if (ge) Lp[nL + __popcll(bl & below)] = (uint16_t)pos;
        if (le) Rp[nR + __popcll(br & below)] = (uint16_t)pos;
        nL += __popcll(bl); nR += __popcll(br);
    }
    __syncthreads();
    const int nm = min(nL, nR);
    int kk = 0;
    for (int i0 = 0; i0 < nm; i0 += 64) {
        const int i = i0 + lane;
        const uint64_t bal = __ballot(i < nm && Lp[i] < Rp[nR - 1 - i]);
        kk += __popcll(bal);
        if (__popcll(bal) < min(64, nm - i0)) break;                       // (L ascends, R descends: once crossed, crossed for good)
    }
    for (int i = lane; i < kk; i += 64) { const int x = Lp[i], y = Rp[nR - 1 - i]; const BpPt t = pt[x]; pt[x] = pt[y]; pt[y] = t; }
    // where the left pointer stops: the next position with a key >= piv in the array AS IT IS NOW - the next entry of L, or the smallest
    // position the swaps have just filled with such a key (R[kk - 1]) when the pointer runs into the swapped region first
    const int cut = min(kk < nL ? (int)Lp[kk] : last, kk > 0 ? (int)Rp[nR - kk] : last);
    __syncthreads();
    return cut;
}

// std::nth_element(first, nth, last) on pt by coordinate d, element for element
__device__ void rb_nth_element_wave(BpPt *pt, int first, int nth, int last, int d, int lane, uint16_t *Lp, uint16_t *Rp)
{
    if (first == last || nth == last) return;
    int depth = 0;
    for (int n = last - first; n > 1; n >>= 1) depth++;
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) {                                                   // introselect's fallback: sequential, as bp_nth_element has it
            if (lane == 0) bp_heap_select_nth(pt, first, nth, last, (const int16_t *)nullptr, d);
            __syncthreads();
            return;
        }
        depth--;
        const int mid = first + (last - first) / 2, ia = first + 1, ib = mid, ic = last - 1;
        const int ka = rb_key(pt, ia, d), kb = rb_key(pt, ib, d), kc = rb_key(pt, ic, d);
        int sm;                                                             // __move_median_to_first
        if (ka < kb) sm = kb < kc ? ib : (ka < kc ? ic : ia);
        else sm = ka < kc ? ia : (kb < kc ? ic : ib);
        __syncthreads();
        if (lane == 0) { const BpPt t = pt[first]; pt[first] = pt[sm]; pt[sm] = t; }
        __syncthreads();
        const int cut = rb_partition_hoare(pt, first, last, d, lane, Lp, Rp);
        if (cut <= nth) first = cut; else last = cut;
    }
    if (lane == 0)
        for (int i = first + 1; i < last; i++) {                            // __insertion_sort
            const BpPt v = pt[i];
            const int kv = (int16_t)(v.v >> (16 * d));
            if (kv < rb_key(pt, first, d)) { for (int j = i; j > first; j--) pt[j] = pt[j - 1]; pt[first] = v; }
            else { int j = i; while (kv < rb_key(pt, j - 1, d)) { pt[j] = pt[j - 1]; j--; } pt[j] = v; }
        }
    __syncthreads();
}

// np.argsort of NumPy 1.22 (npy_aquicksort) by the whole wavefront: tosort = the permutation bp_aquicksort leaves, element for element.
// Quicksort's segments are disjoint, so the order they are processed in does not matter - only each segment's depth budget does (a child's
// is its parent's minus one; the budget is checked on segments that come off the stack, i.e. the LARGER child of a partition and the whole
// array).  Segments of more than QS_WAVE_MIN elements are partitioned by all lanes (median of three on uniform values, then the Hoare
// loop as the two ordered lists of stop positions: L = positions in (pl, pr) whose key is not below the pivot, R = positions in [pl, pr - 1)
// whose key is not above it, read from the right; the loop swaps L[i] with R[i] while L[i] < R[i] and ends at min(L[k], R[k - 1]) -
// tests/test_parallel_partition_model.py); the smaller ones are sorted one lane per segment, all at once, by the sequential code.
// 530 two-valued sigmas on one lane were ~100 us of a lone detection's bookkeeping.  work: 3 x 256 ints, Lp / Rp: num uint16 each.
#define QS_WAVE_MIN 64
__device__ void rb_aquicksort_wave(const uint8_t *v, int num, int16_t *ts, int lane, int *work, uint16_t *Lp, uint16_t *Rp)
{
    for (int i = lane; i < num; i += 64) ts[i] = (int16_t)i;
    __syncthreads();
    if (num < 2) return;
    int cdepth = 0;
    for (int k = num; k > 1; k >>= 1) cdepth++;
    cdepth *= 2;
    // work[3k..3k+2] = (pl, pr, cdepth << 1 | popped); big segments are taken from the top, small ones collected from the bottom of the
    // second half (at most num / 17 + 1 leaves of 17+ elements and as many pending segments: 256 entries hold 2048 keys)
    int sp = 0, nsmall = 0;
    int *small = work + 3 * 128;
    int pl = 0, pr = num - 1, cd = cdepth;
    bool popped = true;
    const uint64_t below = (1ull << lane) - 1ull;
    for (;;) {
        if (pr - pl <= QS_WAVE_MIN || nsmall + sp + 2 >= 128) {           // (the second condition cannot arise; a full list would only cost time)
            if (lane == 0) { small[3 * nsmall] = pl; small[3 * nsmall + 1] = pr; small[3 * nsmall + 2] = cd * 2 + (popped ? 1 : 0); }
            nsmall++;
        } else if (popped && cd < 0) {
            if (lane == 0) bp_aheapsort(v, ts + pl, pr - pl + 1);
            __syncthreads();
        } else {
            const int pm = pl + ((pr - pl) >> 1);
            int a = ts[pl], b = ts[pm], c = ts[pr];                        // (uniform reads)
            if (v[b] < v[a]) { const int t = a; a = b; b = t; }
            if (v[c] < v[b]) { const int t = c; c = b; b = t; }
            if (v[b] < v[a]) { const int t = a; a = b; b = t; }
            const int vp = v[b], old = ts[pr - 1];
            __syncthreads();
            if (lane == 0) { ts[pl] = (int16_t)a; ts[pr] = (int16_t)c; ts[pm] = (int16_t)old; ts[pr - 1] = (int16_t)b; }
            if (pm == pr - 1 && lane == 0) ts[pm] = (int16_t)b;          // (never: pr - pl > 64)
            __syncthreads();
            int nL = 0, nR = 0;
            for (int c0 = pl; c0 < pr; c0 += 64) {
                const int pos = c0 + lane;
                const bool in = pos < pr;
                const int k = in ? (int)v[ts[pos]] : 0;
                const bool ge = in && pos > pl && !(k < vp), le = in && pos < pr - 1 && !(vp < k);
                const uint64_t bl = __ballot(ge), br = __ballot(le);
                if (ge) Lp[nL + __popcll(bl & below)] = (uint16_t)pos;
                if (le) Rp[nR + __popcll(br & below)] = (uint16_t)pos;
                nL += __popcll(bl); nR += __popcll(br);
            }
            __syncthreads();
            const int nm = min(nL, nR);
            int kk = 0;
            for (int i0 = 0; i0 < nm; i0 += 64) {
                const int i = i0 + lane;
                const uint64_t bal = __ballot(i < nm && Lp[i] < Rp[nR - 1 - i]);
                kk += __popcll(bal);
                if (__popcll(bal) < min(64, nm - i0)) break;
            }
            for (int i = lane; i < kk; i += 64) { const int x = Lp[i], y = Rp[nR - 1 - i]; const int16_t t = ts[x]; ts[x] = ts[y]; ts[y] = t; }
            const int pi = kk == 0 ? (int)Lp[0] : min((int)Lp[kk], (int)Rp[nR - kk]);
            __syncthreads();
            if (lane == 0) { const int16_t t = ts[pi]; ts[pi] = ts[pr - 1]; ts[pr - 1] = t; }
            __syncthreads();
            cd--;
            int ql, qr;                                                   // the larger part goes to the stack, the smaller one is next
            if (pi - pl < pr - pi) { ql = pi + 1; qr = pr; pr = pi - 1; }
            else { ql = pl; qr = pi - 1; pl = pi + 1; }
            if (lane == 0) { work[3 * sp] = ql; work[3 * sp + 1] = qr; work[3 * sp + 2] = cd; }
            sp++;
            popped = false;
            continue;
        }
        if (sp == 0) break;
        sp--;
        __syncthreads();
        pl = work[3 * sp]; pr = work[3 * sp + 1]; cd = work[3 * sp + 2];
        popped = true;
    }
    __syncthreads();
    for (int k = lane; k < nsmall; k += 64) {
        const int a = small[3 * k], b = small[3 * k + 1], c = small[3 * k + 2];
        bp_aquicksort_range(v, ts, a, b, c >> 1, (c & 1) != 0);
    }
    __syncthreads();
}

// bp_build_node by the whole wavefront (uniform start / end); the same return value and node fields
__device__ int rb_build_node_wave(BpPt *pt, int start, int end, BpNode &nd, int lane, uint16_t *Lp, uint16_t *Rp)
{
    nd.start = (int16_t)start; nd.end = (int16_t)end; nd.less = nd.greater = -1; nd.split_dim = -1; nd.split = 0;
    if (end - start <= BP_LEAF) return -1;
    int mx0 = -32768, mn0 = 32767, mx1 = -32768, mn1 = 32767;
    for (int j = start + lane; j < end; j += 64) {
        const int v0 = rb_key(pt, j, 0), v1 = rb_key(pt, j, 1);
        mx0 = max(mx0, v0); mn0 = min(mn0, v0); mx1 = max(mx1, v1); mn1 = min(mn1, v1);
    }
    for (int m = 32; m >= 1; m >>= 1) {
        mx0 = max(mx0, __shfl_xor(mx0, m)); mn0 = min(mn0, __shfl_xor(mn0, m));
        mx1 = max(mx1, __shfl_xor(mx1, m)); mn1 = min(mn1, __shfl_xor(mn1, m));
    }
    int d = 0, size = 0;
    if (mx0 - mn0 > size) { d = 0; size = mx0 - mn0; }
    if (mx1 - mn1 > size) { d = 1; size = mx1 - mn1; }
    if (size <= 0) return -1;
    const int half = (end - start) / 2;
    rb_nth_element_wave(pt, start, start + half, end, d, lane, Lp, Rp);
    int split = rb_key(pt, start + half, d);
    // scipy's partition pass: p advances over keys < split, q retreats over keys >= split, misplaced pairs are swapped: afterwards the
    // keys below the split fill [start, p)
    const uint64_t below = (1ull << lane) - 1ull;
    int nL = 0, nR = 0;
    for (int c0 = start; c0 < end; c0 += 64) {
        const int pos = c0 + lane;
        const int k = pos < end ? rb_key(pt, pos, d) : 0;
        const bool ge = pos < end && k >= split, lt = pos < end && k < split;
        const uint64_t bl = __ballot(ge), br = __ballot(lt);
        if (ge) Lp[nL + __popcll(bl & below)] = (uint16_t)pos;
        if (lt) Rp[nR + __popcll(br & below)] = (uint16_t)pos;
        nL += __popcll(bl); nR += __popcll(br);
    }
    __syncthreads();
    {
        const int nm = min(nL, nR);
        int kk = 0;
        for (int i0 = 0; i0 < nm; i0 += 64) {
            const int i = i0 + lane;
            const uint64_t bal = __ballot(i < nm && Lp[i] < Rp[nR - 1 - i]);
            kk += __popcll(bal);
            if (__popcll(bal) < min(64, nm - i0)) break;
        }
        for (int i = lane; i < kk; i += 64) { const int x = Lp[i], y = Rp[nR - 1 - i]; const BpPt t = pt[x]; pt[x] = pt[y]; pt[y] = t; }
    }
    __syncthreads();
    int p = start + nR;                                                     // = start + the number of keys below the split
    if (p == start || p == end) {                                           // (no point on one side: slide to the smallest / largest - sequential, rare)
        if (lane == 0) {
            if (p == start) {
                int j = start; split = rb_key(pt, j, d);
                for (int k = start + 1; k < end; k++) if (rb_key(pt, k, d) < split) { j = k; split = rb_key(pt, j, d); }
                const BpPt t = pt[start]; pt[start] = pt[j]; pt[j] = t;
            } else {
                int j = end - 1; split = rb_key(pt, j, d);
                for (int k = start; k < end - 1; k++) if (rb_key(pt, k, d) > split) { j = k; split = rb_key(pt, j, d); }
                const BpPt t = pt[end - 1]; pt[end - 1] = pt[j]; pt[j] = t;
            }
        }
        __syncthreads();
        split = rb_key(pt, p == start ? start : end - 1, d);
        p = p == start ? start + 1 : end - 1;
    }
    nd.split_dim = (int16_t)d; nd.split = split;
    return p;
}

// bp_pyset_order by the whole wavefront: the same tables, slot for slot.  What is sequential in a CPython set is the INSERTION (the slot a
// key takes depends on the slots taken before it); everything around it is not - the tuple hashes (two 64-bit multiplications each: 64
// keys at a time, one per lane), clearing a new table, listing the old table's keys in slot order at a resize, reading the final table
// out.  The insertions themselves run on wave-uniform values (hash and key by v_readlane, uniform table reads), chunk by chunk up to
// the next resize.  order doubles as the scratch list of a resize (it is the output: free until the end).
template <bool TLDS, typename ORD>
__device__ int rb_pyset_order_wave(const uint32_t *pairs, int np, uint16_t *tabA, int capA, uint16_t *tabB, int capB, ORD *order, int lane)
{
    uint16_t *tab = tabA, *other = tabB;
    int cap_other = capB, cap_cur = capA;
    uint32_t mask = 7;
    if (lane < 8) tab[lane] = 0;
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
    int fill = 0;
    // src = nullptr: the pairs p0 .. p0 + cnt - 1 themselves; else the keys listed in src[0 .. cnt)
    auto insert_chunk = [&](uint16_t *t, uint32_t msk, const ORD *src, int p0, int cnt) {
        const int key1 = lane < cnt ? (src ? (int)src[p0 + lane] : p0 + lane + 1) : 0;
        const uint64_t h = key1 ? bp_tuple_hash(pairs[key1 - 1]) : 0ull;
        for (int j = 0; j < cnt; j++) {
            const uint64_t hj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(h >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(h & 0xffffffffull), j);
            const int kj = __builtin_amdgcn_readlane(key1, j);
            uint64_t perturb = hj;
            uint32_t i = (uint32_t)hj & msk;
            for (;;) {
                const int probes = (i + 9 <= msk) ? 9 : 0;
                int e = -1;
                for (int q = 0; q <= probes; q++) if (t[i + q] == 0) { e = (int)i + q; break; }
                if (e >= 0) { if (lane == 0) t[e] = (uint16_t)kj; break; }
                perturb >>= 5;
                i = (uint32_t)(((uint64_t)i * 5 + 1 + perturb) & msk);
            }
            // the next key reads what this one wrote: a wavefront's LDS operations execute in order (tables in LDS, TLDS); a table in
            // global memory (more than 4 914 pairs) is written through before it is read again
            if (!TLDS) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
    };
    for (int p = 0; p < np;) {
        // insertions until the next resize: the smallest k with (fill + k) * 5 >= mask * 3
        int room = (int)(((uint64_t)mask * 3 + 4) / 5) - fill;
        if (room < 1) room = 1;
        const int cnt = min(min(64, np - p), room);
        insert_chunk(tab, mask, (const ORD *)nullptr, p, cnt);
        p += cnt; fill += cnt;
        __syncthreads();
        if ((uint64_t)fill * 5 >= (uint64_t)mask * 3) {
            const int minused = fill > 50000 ? fill * 2 : fill * 4;
            uint32_t newsize = 8;
            while ((int)newsize <= minused) newsize <<= 1;
            if ((int)newsize > cap_other) return -1;
            for (uint32_t k = lane; k < newsize; k += 64) other[k] = 0;
            int m = 0;
            for (uint32_t k0 = 0; k0 <= mask; k0 += 64) {
                const uint32_t k = k0 + lane;
                const int key = k <= mask ? (int)tab[k] : 0;
                const uint64_t bal = __ballot(key != 0);
                if (key) order[m + __popcll(bal & below)] = (ORD)key;
                m += __popcll(bal);
            }
            __syncthreads();
            for (int q = 0; q < m; q += 64) insert_chunk(other, newsize - 1, order, q, min(64, m - q));
            __syncthreads();
            uint16_t *t = tab; tab = other; other = t;
            const int c = cap_cur; cap_cur = cap_other; cap_other = c;
            mask = newsize - 1;
        }
    }
    __syncthreads();
    int m = 0;
    for (uint32_t k0 = 0; k0 <= mask; k0 += 64) {
        const uint32_t k = k0 + lane;
        const int key = k <= mask ? (int)tab[k] : 0;
        const uint64_t bal = __ballot(key != 0);
        // (order may still hold a resize's list below m: every entry is rewritten before it is read again, and what is written at
        // position m + rank comes from slot k >= its old position - the final scan only ever overwrites entries it has passed)
        if (key) order[m + __popcll(bal & below)] = (ORD)(key - 1);
        m += __popcll(bal);
    }
    __syncthreads();
    return m;
}

#ifdef RB_EXP_PROF
__device__ unsigned long long rb_prof[16];
extern "C" int roam_debug_blob_prof(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rb_prof), sizeof(rb_prof)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rb_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define RB_P(k) { const unsigned long long tn_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) rbp_[k] += tn_ - rbt_; rbt_ = tn_; }
#else
#define RB_P(k)
#endif
// the bookkeeping of one detection (one wavefront); -> 1 when the small class left it to the full one (kp_n = RT_BLOBS_REDO)
template <bool SMALL>
__device__ int rt_blobs_body(const RtArgs &a, int first, RtBlobLds<SMALL> &L, bool forced)
{
#ifdef RB_EXP_PROF
    unsigned long long rbp_[12] = {0}, rbt_ = __builtin_readcyclecounter();
#endif
    typedef RtBlobCap<SMALL> CP;
    // (L: the workgroup's LDS, the caller's)
    const int ls = a.blob_order ? a.blob_order[blockIdx.x] : (int)blockIdx.x, slot = first + ls;
    if (slot >= *a.rt_n) return 0;
    const int lane = threadIdx.x;
    const int ncand = a.cand_n[ls];
    if (!SMALL && !forced && a.kp_n[ls] != RT_BLOBS_REDO) return 0;                      // the small kernel did it
    if (SMALL && ncand > CP::NP) {
        if (lane == 0) a.kp_n[ls] = RT_BLOBS_REDO;
        return 1;
    }
    const int n = min(ncand, BP_MAX_PTS);
    const uint32_t *crc = a.cand_rc + (int64_t)ls * BP_MAX_PTS;
    const double *cval = a.cand_val + (int64_t)ls * BP_MAX_PTS;
    double *kp = a.kp + (int64_t)ls * BP_MAX_PTS * 3;
    int flags = ncand > BP_MAX_PTS ? RT_F_CAND_OVERFLOW : 0;
    // 1. response order: peak_local_max sorts by -intensity; equal responses keep the C (row, col, layer) order.  The responses are
    // staged in LDS (the set tables are free until step 5) and every lane ranks four candidates per pass against broadcast reads
    // (straight out of global memory the n^2 / 64 dependent loads were a sixth of this kernel)
    double *sval = reinterpret_cast<double *>(L.tabB);
    for (int i = lane; i < n; i += 64) sval[i] = cval[i];
    __syncthreads();
    for (int i0 = lane; i0 < n; i0 += 256) {
        double vi[4];
        int rank[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 4; u++) vi[u] = i0 + 64 * u < n ? sval[i0 + 64 * u] : 0.0;
        for (int j = 0; j < n; j++) {
            const double vj = sval[j];
#pragma unroll
            for (int u = 0; u < 4; u++) rank[u] += (vj > vi[u] || (vj == vi[u] && j < i0 + 64 * u)) ? 1 : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + 64 * u;
            if (i < n) {
                const uint32_t p = crc[i];
                L.xy[2 * rank[u]] = (int16_t)(p >> 16); L.xy[2 * rank[u] + 1] = (int16_t)((p >> 2) & 0x3fff); L.lay[rank[u]] = (uint8_t)(p & 3);
            }
        }
    }
    __syncthreads();
    RB_P(0)
    // 2. cKDTree, level by level: every node of a level is built by its own lane (bounds, libstdc++ nth_element, scipy's
    // partition passes - sequential per node, but a level's nodes work on disjoint index ranges), so the critical path is the
    // largest node of every level (~2 n element visits) instead of all of them (~n log n: the build on lane 0 was 1.6 of this
    // kernel's 3.2 ms).  Nodes are numbered level by level; nothing downstream depends on the numbering, only on the links.
    BpPt *pt = reinterpret_cast<BpPt *>(L.tabB);          // {row, col, index} per element: keys without a dependent second read
    for (int i = lane; i < n; i += 64)
        pt[i].v = (uint64_t)(uint16_t)L.xy[2 * i] | ((uint64_t)(uint16_t)L.xy[2 * i + 1] << 16) | ((uint64_t)i << 32);
    int nn = 0;
    if (n > 0) {
        {
            int mn0 = 32767, mx0 = -32768, mn1 = 32767, mx1 = -32768;
            for (int i = lane; i < n; i += 64) {
                const int v0 = L.xy[2 * i], v1 = L.xy[2 * i + 1];
                mn0 = min(mn0, v0); mx0 = max(mx0, v0); mn1 = min(mn1, v1); mx1 = max(mx1, v1);
            }
            for (int d = 32; d >= 1; d >>= 1) {
                mn0 = min(mn0, __shfl_xor(mn0, d)); mx0 = max(mx0, __shfl_xor(mx0, d));
                mn1 = min(mn1, __shfl_xor(mn1, d)); mx1 = max(mx1, __shfl_xor(mx1, d));
            }
            if (lane == 0) { L.nbox[0][0] = (int16_t)mn0; L.nbox[0][1] = (int16_t)mx0; L.nbox[0][2] = (int16_t)mn1; L.nbox[0][3] = (int16_t)mx1; }
        }
        if (lane == 0) { L.nodes[0].start = 0; L.nodes[0].end = (int16_t)n; }
        nn = 1;
        __syncthreads();
        for (int lo = 0, hi = 1; lo < hi && nn > 0;) {
            for (int base = lo; base < hi; base += 64) {
                const int me = base + lane;
                const bool act = me < hi;
                BpNode nd;
                int p = -1, start = 0, end = 0;
                if (act) { start = L.nodes[me].start; end = L.nodes[me].end; }
                // the large nodes of the chunk one at a time by the whole wave, the others lane by lane
                uint64_t big = __ballot(act && end - start > RB_WAVE_MIN);
                const uint64_t bigs = big;
                while (big) {
                    const int j = __ffsll((long long)big) - 1;
                    big &= big - 1;
                    BpNode ndw;
                    const int pw = rb_build_node_wave(pt, __builtin_amdgcn_readlane(start, j), __builtin_amdgcn_readlane(end, j), ndw, lane,
                                                      reinterpret_cast<uint16_t *>(L.pl), reinterpret_cast<uint16_t *>(L.pl) + CP::NP);
                    if (lane == j) { p = pw; nd = ndw; }
                }
                if (act && !((bigs >> lane) & 1ull)) p = bp_build_node(L.xy, pt, start, end, nd);
                const uint64_t bal = __ballot(act && p >= 0);
                const int kids = 2 * __popcll(bal);
                if (nn + kids > CP::NNODE) { nn = -1; break; }                 // (uniform)
                if (act) {
                    if (p >= 0) {
                        const int c0 = nn + 2 * __popcll(bal & ((1ull << lane) - 1ull));
                        nd.less = (int16_t)c0; nd.greater = (int16_t)(c0 + 1);
                        L.nodes[c0].start = (int16_t)start; L.nodes[c0].end = (int16_t)p;
                        L.nodes[c0 + 1].start = (int16_t)p; L.nodes[c0 + 1].end = (int16_t)end;
                        for (int k = 0; k < 4; k++) L.nbox[c0][k] = L.nbox[c0 + 1][k] = L.nbox[me][k];
                        L.nbox[c0][2 * nd.split_dim + 1] = (int16_t)nd.split;      // less: max = split
                        L.nbox[c0 + 1][2 * nd.split_dim] = (int16_t)nd.split;      // greater: min = split
                    }
                    L.nodes[me] = nd;
                }
                nn += kids;
            }
            if (nn < 0) break;
            __syncthreads();
            lo = hi; hi = nn;
        }
    }
    __syncthreads();
    if (SMALL && nn < 0) {                                                   // more tree nodes than the small class holds: the full kernel's
        if (lane == 0) a.kp_n[ls] = RT_BLOBS_REDO;
        return 1;
    }
    for (int i = lane; i < n; i += 64) L.idx[i] = (int16_t)(pt[i].v >> 32);          // cKDTree.indices
    __syncthreads();
    RB_P(1)
    // dual-tree traversal -> ordered leaf x leaf blocks.  query_pairs recurses over node pairs with a distance tracker it pushes and pops; on
    // integer pixel coordinates every quantity of that tracker is an exact integer, so its state at a node pair is a function of the two
    // nodes' boxes alone (nbox) and the recursion needs no stack: the pairs are expanded LEVEL BY LEVEL, every item replaced in place by its
    // children in the recursion's order (a finished leaf x leaf block is its own child), so that the list stays in emission order throughout.
    // One lane per item, ballots for the offsets; two lists of BP_MAX_TASKS packed items ping-pong in the detection's pair scratch.  (On lane 0
    // the recursion was 45 % of this kernel: ~230 us of a lone detection.)  item = a | b << 10 | m << 20; m: 0 check, 1 no check, 2 / 3 block
    // with / without the distance test
    BpTask *tasks = a.tasks + (int64_t)ls * BP_MAX_TASKS;
    int nt = 0;
    double ub = 0;
    {
        bool any2l = false;
        for (int i = lane; i < n; i += 64) any2l = any2l || L.lay[i] == 2;
        const bool any2 = __ballot(any2l) != 0;
        const double r = 2 * (any2 ? a.sigma2 : a.sigma1) * 1.4142135623730951;      // _prune_blobs: distance = 2 * max sigma * sqrt(2)
        ub = r * r;
    }
    if (nn < 0) flags |= RT_F_TREE_OVERFLOW;
    else if (n > 1) {
        const int t_gt = (int)floor(ub), t_lt = (int)ceil(ub);              // integer d: d > ub <=> d > t_gt, d < ub <=> d < t_lt
        uint32_t *fa = a.pairs + (int64_t)ls * (BP_MAX_PAIRS + 1), *fb = fa + BP_MAX_TASKS;
        if (lane == 0) fa[0] = 0u;                                          // (root, root, check)
        int F = 1;
        bool over = false;
        const uint64_t below = (1ull << lane) - 1ull;
        for (;;) {
            __syncthreads();
            int total = 0;
            bool open = false;
            for (int c0 = 0; c0 < F; c0 += 64) {
                const bool act = c0 + lane < F;
                const uint32_t it = act ? fa[c0 + lane] : 0u;
                const int na = it & 1023, nb = (it >> 10) & 1023;
                int m = act ? (int)(it >> 20) : 2;
                uint32_t ch[4];
                int cnt = 0;
                if (act && m >= 2) { ch[0] = it; cnt = 1; }
                else if (act) {
                    const BpNode n1 = L.nodes[na], n2 = L.nodes[nb];
                    const bool l1 = n1.split_dim == -1, l2 = n2.split_dim == -1;
                    bool pruned = false;
                    if (m == 0) {
                        int mind = 0, maxd = 0;
                        for (int k = 0; k < 2; k++) {
                            const int a0 = L.nbox[na][2 * k], a1 = L.nbox[na][2 * k + 1], b0 = L.nbox[nb][2 * k], b1 = L.nbox[nb][2 * k + 1];
                            const int lo = max(max(a0 - b1, b0 - a1), 0), hi = max(a1 - b0, b1 - a0);
                            mind += lo * lo; maxd += hi * hi;
                        }
                        if (mind > t_gt) pruned = true;
                        else if (maxd < t_lt) m = 1;
                    }
                    const uint32_t mm = (uint32_t)m << 20;
                    if (pruned) cnt = 0;
                    else if (l1 && l2) { ch[0] = (uint32_t)na | ((uint32_t)nb << 10) | ((m == 0 ? 2u : 3u) << 20); cnt = 1; }
                    else if (m == 1) {
                        if (l1) { ch[0] = na | ((uint32_t)n2.less << 10) | mm; ch[1] = na | ((uint32_t)n2.greater << 10) | mm; cnt = 2; }
                        else if (na == nb) {
                            ch[0] = n1.less | ((uint32_t)n2.less << 10) | mm; ch[1] = n1.less | ((uint32_t)n2.greater << 10) | mm;
                            ch[2] = n1.greater | ((uint32_t)n2.greater << 10) | mm; cnt = 3;
                        } else { ch[0] = n1.less | ((uint32_t)nb << 10) | mm; ch[1] = n1.greater | ((uint32_t)nb << 10) | mm; cnt = 2; }
                    } else {
                        if (l1) { ch[0] = na | ((uint32_t)n2.less << 10); ch[1] = na | ((uint32_t)n2.greater << 10); cnt = 2; }
                        else if (l2) { ch[0] = n1.less | ((uint32_t)nb << 10); ch[1] = n1.greater | ((uint32_t)nb << 10); cnt = 2; }
                        else {
                            ch[0] = n1.less | ((uint32_t)n2.less << 10); ch[1] = n1.less | ((uint32_t)n2.greater << 10); cnt = 2;
                            if (na != nb) ch[cnt++] = n1.greater | ((uint32_t)n2.less << 10);
                            ch[cnt++] = n1.greater | ((uint32_t)n2.greater << 10);
                        }
                    }
                }
                const uint64_t b1 = __ballot(cnt >= 1), b2 = __ballot(cnt >= 2), b3 = __ballot(cnt >= 3), b4 = __ballot(cnt >= 4);
                const int off = total + __popcll(b1 & below) + __popcll(b2 & below) + __popcll(b3 & below) + __popcll(b4 & below);
                const int sum = __popcll(b1) + __popcll(b2) + __popcll(b3) + __popcll(b4);
                if (total + sum > BP_MAX_TASKS) { over = true; break; }     // (uniform)
                for (int j = 0; j < cnt; j++) fb[off + j] = ch[j];
                open = open || (cnt > 0 && (ch[0] >> 20) < 2u);             // (a node's children are all of one kind)
                total += sum;
            }
            if (over) break;
            uint32_t *t = fa; fa = fb; fb = t;
            F = total;
            if (!__ballot(open)) break;
        }
        __syncthreads();
        if (over) flags |= RT_F_TREE_OVERFLOW;
        else {
            nt = F;
            for (int i = lane; i < nt; i += 64) {
                const uint32_t it = fa[i];
                tasks[i].a = (int16_t)(it & 1023); tasks[i].b = (int16_t)((it >> 10) & 1023); tasks[i].mode = (int32_t)(it >> 20) - 2;
            }
        }
    }
    __syncthreads();
    RB_P(2)
    // 3. the pairs of the blocks in emission order (i-major, j ascending), 64 candidates per ballot
    uint32_t *pairs = a.pairs + (int64_t)ls * (BP_MAX_PAIRS + 1);
    int np = 0;
    for (int t = 0; t < nt; t++) {
        const BpTask tk = tasks[t];
        const BpNode n1 = L.nodes[tk.a], n2 = L.nodes[tk.b];
        const int la = n1.end - n1.start, lb = n2.end - n2.start, tot = la * lb;
        const bool same = tk.a == tk.b;
        for (int base = 0; base < tot; base += 64) {
            const int k = base + lane;
            bool ok = k < tot;
            int pi = 0, pj = 0;
            if (ok) {
                const int i = n1.start + k / lb, j = n2.start + k % lb;
                if (same && j <= i) ok = false;
                else {
                    pi = L.idx[i]; pj = L.idx[j];
                    if (!tk.mode) {
                        const double d0 = (double)L.xy[2 * pi] - (double)L.xy[2 * pj], d1 = (double)L.xy[2 * pi + 1] - (double)L.xy[2 * pj + 1];
                        ok = d0 * d0 + d1 * d1 <= ub;
                    }
                }
            }
            const uint64_t bal = __ballot(ok);
            const int o = np + __popcll(bal & ((1ull << lane) - 1ull));
            if (ok && o < BP_MAX_PAIRS) { pairs[o] = bp_pack(pi, pj); if (o < CP::NPL) L.pl[o] = bp_pack(pi, pj); }
            np += __popcll(bal);
        }
    }
    if (np > BP_MAX_PAIRS) { flags |= RT_F_PAIR_OVERFLOW; np = BP_MAX_PAIRS; }
    __syncthreads();
    RB_P(3)
    // 4. which pairs overlap by more than 0.5 (original sigmas: a pair with a pruned member never changes anything)
    if (SMALL && np > CP::LDS_PAIRS) {                                       // more pairs than the small set tables order: the full kernel's
        if (lane == 0) a.kp_n[ls] = RT_BLOBS_REDO;
        return 1;
    }
    const bool lds_set = np <= CP::LDS_PAIRS, lds_pl = np <= CP::NPL;
    uint32_t *ovb = lds_set ? L.ovbits : a.ovbits + (int64_t)ls * ((BP_MAX_PAIRS + 31) / 32 + 1);
    for (int w0 = 0; w0 < np; w0 += 64) {
        const int k = w0 + lane;
        bool ov = false;
        if (k < np) {
            const uint32_t pr = lds_pl ? L.pl[k] : pairs[k];
            const int i = (int)(pr >> 16), j = (int)(pr & 0xffffu);
            ov = bp_overlaps((double)L.xy[2 * i], (double)L.xy[2 * i + 1], L.lay[i] == 2 ? a.sigma2 : a.sigma1,
                             (double)L.xy[2 * j], (double)L.xy[2 * j + 1], L.lay[j] == 2 ? a.sigma2 : a.sigma1, 0.5);
        }
        const uint64_t bal = __ballot(ov);
        if (lane == 0) { ovb[w0 >> 5] = (uint32_t)bal; ovb[(w0 >> 5) + 1] = (uint32_t)(bal >> 32); }
    }
    __syncthreads();
    RB_P(4)
    // 5. Python-set iteration order of the pairs + the sequential pruning pass, then 6. NumPy-1.22 argsort of the sigmas
    uint16_t *order = a.order + (int64_t)ls * (BP_MAX_PAIRS + 1);
    {
        int m;
        if (lds_pl) m = rb_pyset_order_wave<true>(L.pl, np, L.tabA, CP::CAPA, L.tabB, CP::CAPB, order, lane);
        else if (lds_set) m = rb_pyset_order_wave<true>(pairs, np, L.tabA, CP::CAPA, L.tabB, CP::CAPB, order, lane);
        else {
            uint16_t *big = a.bigtab + (int64_t)ls * 2 * 131072;
            m = rb_pyset_order_wave<false>(pairs, np, big, 131072, big + 131072, 131072, order, lane);
        }
        if (m != np) flags |= RT_F_PAIR_OVERFLOW;
        if (lane == 0) L.vals[2] = m < 0 ? 0 : m;
    }
    __syncthreads();
    RB_P(5)
    // the overlapping pairs in set order, gathered by the whole wave (the hash tables are free again: 4096 pairs fit in tabB);
    // the sequential pass then walks LDS only - one lane chasing order[k] -> pairs[q] through global memory was 2/3 of this kernel
    const int m = L.vals[2];
    uint32_t *cl = reinterpret_cast<uint32_t *>(L.tabB);
    int ncl = 0;
    for (int k0 = 0; k0 < m; k0 += 64) {
        const int k = k0 + lane;
        bool ov = false;
        uint32_t pr = 0;
        if (k < m) {
            const int q = order[k];
            ov = (ovb[q >> 5] >> (q & 31)) & 1u;
            if (ov) pr = lds_pl ? L.pl[q] : pairs[q];
        }
        const uint64_t bal = __ballot(ov);
        const int o = ncl + __popcll(bal & ((1ull << lane) - 1ull));
        if (ov && o < CP::NCL) cl[o] = pr;
        ncl += __popcll(bal);
    }
    __syncthreads();
    if (lane == 0) {
        if (ncl <= CP::NCL) {
            for (int k = 0; k < ncl; k++) {
                const uint32_t pr = cl[k];
                const int i = (int)(pr >> 16), j = (int)(pr & 0xffffu);
                if (L.lay[i] == 0 || L.lay[j] == 0) continue;
                if (L.lay[i] > L.lay[j]) L.lay[j] = 0; else L.lay[i] = 0;     // sigma_i > sigma_j ? prune j : prune i (ties: i)
            }
        } else {
            for (int k = 0; k < m; k++) {
                const int q = order[k];
                if (!((ovb[q >> 5] >> (q & 31)) & 1u)) continue;
                const uint32_t pr = pairs[q];
                const int i = (int)(pr >> 16), j = (int)(pr & 0xffffu);
                if (L.lay[i] == 0 || L.lay[j] == 0) continue;
                if (L.lay[i] > L.lay[j]) L.lay[j] = 0; else L.lay[i] = 0;
            }
        }
        L.vals[1] = flags;
    }
    __syncthreads();
    RB_P(6)
    // survivors in response order (reuse xy / lay in place: a chunk of 64 is read before anything of it is overwritten, and what it writes
    // lies at or below its own positions), then sorted by sigma with NumPy 1.22's tie order.  (On lane 0 this loop was ~40 us of a lone
    // detection: 530 dependent LDS round trips.)
    int mb = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const int ly = i < n ? (int)L.lay[i] : 0;
        const int16_t x = i < n ? L.xy[2 * i] : (int16_t)0, y = i < n ? L.xy[2 * i + 1] : (int16_t)0;
        const uint64_t bal = __ballot(ly != 0);
        __syncthreads();
        if (ly) { const int o = mb + __popcll(bal & ((1ull << lane) - 1ull)); L.xy[2 * o] = x; L.xy[2 * o + 1] = y; L.lay[o] = (uint8_t)ly; }
        mb += __popcll(bal);
        __syncthreads();
    }
    RB_P(7)
    rb_aquicksort_wave(L.lay, mb, L.idx, lane, L.st, reinterpret_cast<uint16_t *>(L.pl), reinterpret_cast<uint16_t *>(L.pl) + CP::NP);
    RB_P(8)
    for (int q = lane; q < mb; q += 64) {
        const int i = L.idx[q];
        kp[3 * q] = (double)L.xy[2 * i]; kp[3 * q + 1] = (double)L.xy[2 * i + 1]; kp[3 * q + 2] = L.lay[i] == 2 ? a.sigma2 : a.sigma1;
    }
    if (lane == 0) { a.kp_n[ls] = mb; a.slot_flags[ls] = L.vals[1]; }
    RB_P(9)
#ifdef RB_EXP_PROF
    if (lane == 0) for (int k = 0; k < 12; k++) atomicAdd(&rb_prof[k], rbp_[k]);
    if (lane == 0) { atomicAdd(&rb_prof[12], 1ull); atomicAdd(&rb_prof[13], (unsigned long long)n); atomicAdd(&rb_prof[14], (unsigned long long)mb); }
#endif
    return 0;
}

// large batches: every detection through the small class (33 KB of LDS: four per CU), then the few it left through the full one (64 KB)
template <bool SMALL>
__global__ __launch_bounds__(64) void rt_blobs_kernel(RtArgs a, int first)
{
    __shared__ RtBlobLds<SMALL> L;
    rt_blobs_body<SMALL>(a, first, L, false);
}
// small batches (a single sequence's lone detection): K4 (candidate order), K5 (both classes) and K6 (SSC) of a detection in ONE launch,
// one wavefront - three kernels less in the chain every step enqueues whether or not a lane re-detects (a kernel that only returns costs
// 4-6 us of the device and of the enqueuing thread; a single sequence's pair is 290 us).  The full class's LDS per workgroup; the same
// code, so the same results; K4 on 64 threads instead of 256 costs a lone detection ~10 us.
__global__ __launch_bounds__(64) void rt_book_kernel(RtArgs a, int first)
{
    __shared__ union U_ {
        RtBlobLds<true> s; RtBlobLds<false> f;
        struct { uint32_t key[BP_MAX_PTS]; double val[BP_MAX_PTS]; } e;
        uint32_t bitmap[SSC_BATCH_BITMAP_BYTES / 4];
        __device__ U_() {}
    } L;
    const int ls = blockIdx.x, slot = first + ls, lane = threadIdx.x;
    if (slot >= *a.rt_n) return;
    {   // K4 (rt_emit_kernel)
        const int n = min(a.cand_n[ls], BP_MAX_PTS);
        uint32_t *crc = a.cand_rc + (int64_t)ls * BP_MAX_PTS;
        double *cval = a.cand_val + (int64_t)ls * BP_MAX_PTS;
        for (int i = lane; i < n; i += 64) { L.e.key[i] = crc[i]; L.e.val[i] = cval[i]; }
        __syncthreads();
        for (int i0 = lane; i0 < n; i0 += 256) {
            uint32_t k[4];
            int rank[4] = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 4; u++) k[u] = i0 + 64 * u < n ? L.e.key[i0 + 64 * u] : 0u;
            for (int j = 0; j < n; j++) {
                const uint32_t kj = L.e.key[j];
#pragma unroll
                for (int u = 0; u < 4; u++) rank[u] += kj < k[u] ? 1 : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (i0 + 64 * u < n) { crc[rank[u]] = k[u]; cval[rank[u]] = L.e.val[i0 + 64 * u]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");                   // the lists are read back (by other lanes) right below
        __syncthreads();
    }
    if (rt_blobs_body<true>(a, first, L.s, false)) {
        __syncthreads();
        rt_blobs_body<false>(a, first, L.f, true);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    __syncthreads();
    // K6 (ssc_batch_kernel)
    const int nk = min(a.kp_n[ls], BP_MAX_PTS);
    if (nk <= 0) { if (lane == 0) a.sel_n[ls] = 0; return; }
    ssc_body(a.kp + (int64_t)ls * BP_MAX_PTS * 3, nk, 200, 0.1, a.W, a.W, a.ssc_work + (int64_t)ls * 4 * BP_MAX_PTS,
             a.sel + (int64_t)ls * BP_MAX_PTS, a.sel_n + ls, L.bitmap, SSC_BATCH_BITMAP_BYTES);
}

// ------------------------------------------------------------------------------------------------ K7: append + keyframe refresh
__global__ __launch_bounds__(256) void rt_append_kernel(RtArgs a, int first)
{
    __shared__ float nx[KS + 256], ny[KS + 256];
    __shared__ uint8_t keep[KS + 256];
    __shared__ int sh[8];
    const int ls = blockIdx.x, slot = first + ls;
    if (slot >= *a.rt_n) return;
    const int b = a.rt_lane[slot], t = threadIdx.x;
    if (t == 0) a.cand_n[ls] = 0;                                             // the last reader of the list is done: clean for the next detection
    float *feat = a.feat + (int64_t)b * KS * 2;
    const int n_old = min(a.feat_n[b], KS);
    const int n_sel = min(a.sel_n[ls], 256);                                  // (ANMS returns at most 220; more would be dropped: flagged below)
    const double *kp = a.kp + (int64_t)ls * BP_MAX_PTS * 3;
    const int32_t *sel = a.sel + (int64_t)ls * BP_MAX_PTS;
    // vstack((old, fliplr(new[:, :2])))  (getFeatures.py:101-109)
    for (int i = t; i < n_old; i += 256) { nx[i] = feat[2 * i]; ny[i] = feat[2 * i + 1]; }
    for (int i = t; i < n_sel; i += 256) { const int q = sel[i]; nx[n_old + i] = (float)kp[3 * q + 1]; ny[n_old + i] = (float)kp[3 * q]; }
    __syncthreads();
    const int tot = n_old + n_sel;
    // np.unique(axis=0, return_index) + sort(idx): drop a row when an earlier row is identical
    for (int i = t; i < tot; i += 256) {
        bool dup = false;
        for (int j = 0; j < i && !dup; j++) dup = nx[j] == nx[i] && ny[j] == ny[i];
        keep[i] = dup ? 0 : 1;
    }
    __syncthreads();
    // ordered compaction (tot <= KS + 256 -> 5 items per thread)
    const int items = (KS + 256 + 255) / 256, lo = t * items, hi = min(lo + items, tot);
    int c = 0;
    for (int i = lo; i < hi; i++) c += keep[i];
    const int lane = t & 63, w = t >> 6;
    int inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int n = __shfl_up(inc, d); if (lane >= d) inc += n; }
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    int pos = inc - c, total = 0;
    for (int i = 0; i < 4; i++) { if (i < w) pos += sh[i]; total += sh[i]; }
    const int m = min(total, KS);
    const double v0 = a.vel[3 * b], v1 = a.vel[3 * b + 1], v2 = a.vel[3 * b + 2];
    double *und = a.kf_und + (int64_t)b * KS * 2;
    for (int i = lo; i < hi; i++)
        if (keep[i]) {
            if (pos < KS) {
                feat[2 * pos] = nx[i]; feat[2 * pos + 1] = ny[i];
                // possible_kf.updateInfo(latestPose, centered_new, ..., velocity): undistort (Mapping.py:59-66)
                const double x = ((double)nx[i] - CART_CENTER) * M_PER_PX, y = ((double)ny[i] - CART_CENTER) * M_PER_PX;
                const double dT = 0.25 * atan2(-y, -x) / TWO_PI;
                const double ang = v2 * dT, ca = cos(ang), sa = sin(ang);
                und[2 * pos] = ca * x - sa * y + v0 * dT;
                und[2 * pos + 1] = sa * x + ca * y + v1 * dT;
            }
            pos++;
        }
    if (t == 0) {
        a.feat_n[b] = m;
        if (a.res) {
            a.res[b].flags |= 8 | (a.slot_flags[ls] << 8) | ((total > KS || a.sel_n[ls] > 256) ? (RT_F_FEAT_OVERFLOW << 8) : 0);
            a.res[b].n_after_retrack = m;
        }
    }
}

// ------------------------------------------------------------------------------------------------ launcher
size_t retrack_boxtab_words(int W) { return 2 * (size_t)((W + RI_ROWS - 1) / RI_ROWS) * RI_GROUPS * RI_WAVES; }

hipError_t launch_retrack_boxtab(hipStream_t st, const uint32_t *map, int W, int cols, uint32_t *boxtab)
{
    if (W > 2048) return hipSuccess;                                       // (the one-sweep kernel is not used for such images)
    hipLaunchKernelGGL(rt_boxtab_kernel, dim3((unsigned)(retrack_boxtab_words(W) / 2)), dim3(64), 0, st, map, W, cols, boxtab);
    return hipGetLastError();
}

size_t retrack_darktab_words(int W) { return (size_t)((W + SD_OUT - 1) / SD_OUT) * SD_DT_WORDS; }

hipError_t launch_retrack_darktab(hipStream_t st, const uint32_t *map, int W, int cols, uint32_t *darktab)
{
    const int ns = (W + SD_OUT - 1) / SD_OUT, nt = W / SD_T + 1;
    if (nt > 256) return hipErrorInvalidValue;                               // (the sampling map addresses 4095 range bins: W <= 4094, nt <= 256)
    hipLaunchKernelGGL(rt_darktab_kernel, dim3(ns, nt), dim3(256), 0, st, map, W, cols, darktab);
    hipLaunchKernelGGL(rt_darkskip_kernel, dim3(ns), dim3(256), 0, st, nt, darktab);
    return hipGetLastError();
}

size_t retrack_fused_boxtab_words(int W) { return 2 * (size_t)((W + SD_OUT - 1) / SD_OUT + 1) * (size_t)((W + SD_T - 1) / SD_T); }
size_t retrack_fused_halo_words(int W) { return (size_t)((W + SD_T - 1) / SD_T) * SD_T * FD_HALO; }

hipError_t launch_retrack_fused_tables(hipStream_t st, const uint32_t *map, int W, int cols, uint32_t *mapT, uint32_t *boxtab, uint32_t *darktab)
{
    const int nband = (W + SD_OUT - 1) / SD_OUT, nblk = (W + SD_T - 1) / SD_T;
    hipLaunchKernelGGL(rf_mapT_kernel, dim3((W + 31) / 32, (W + 31) / 32), dim3(256), 0, st, map, W, mapT);
    hipLaunchKernelGGL(rf_boxtab_kernel, dim3(nblk, nband + 1), dim3(64), 0, st, mapT, W, cols, nblk, boxtab);
    // the dark steps of a band = the dark steps of a strip of the transposed image: rt_darktab_kernel on the transposed map
    return launch_retrack_darktab(st, mapT, W, cols, darktab);
}

// ---- which phases of the one-sweep integral kernel matter (host code, once per engine; geometry only).
// The determinant kernel never reads the blocks of the integral image that only dark steps would touch (rt_darktab_kernel), so a 16-row x
// 64-column TILE none of its strips loads need not be written; and a (band, group) PHASE none of whose tiles is needed need not be
// computed when its row sums cannot matter: on a band's left while every column so far has seen no lit pixel (the sums are exactly
// zero), on its right once nothing further along the band is needed.  Needed tiles contain every lit pixel (a lit pixel lies in the
// window of a lit step), so a phase that is left out has dark pixels only: the columns' running sums pass it unchanged.
// out[0] = number of phases, out[1..] = band | group << 8 | needed-tile bits << 12 in sweep order.
size_t retrack_phase_words(int W) { return 1 + (size_t)((W + RI_ROWS - 1) / RI_ROWS) * RI_GROUPS; }
int retrack_band_rows() { return RI_ROWS; }

bool retrack_build_phases(const uint32_t *map, const uint32_t *darktab, int W, int cols, uint32_t *out)
{
    const int H = W, nbands = (H + RI_ROWS - 1) / RI_ROWS, ns = (W + SD_OUT - 1) / SD_OUT, nt = H / SD_T + 1, NT = RI_GROUPS * RI_WAVES;
    static_assert(RI_ROWS % SD_T == 0, "a band of the integral kernel is one or more blocks of the determinant kernel");
    const int ndet = (H + SD_T - 1) / SD_T;
    if (nbands * RI_GROUPS > RI_PHL_MAX || nbands > 256) return false;
    std::vector<uint8_t> need((size_t)nbands * NT, 0), lit((size_t)nbands * NT, 0);
    std::vector<int> firstlit(W, H);
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++)
            if ((int)(map[(size_t)r * W + c] & 4095u) < cols) {
                lit[(size_t)(r / RI_ROWS) * NT + c / 64] = 1;
                if (firstlit[c] == H) firstlit[c] = r;
            }
    for (int s = 0; s < ns; s++) {
        const uint32_t *T = darktab + (size_t)s * SD_DT_WORDS;
        const int t_first = (int)T[16], t_last = (int)T[17];
        if (t_first >= nt) continue;                                        // the strip sees nothing
        const int tb = t_first >= 1 ? ((t_first - 1) & ~3) : 0, te = std::min(nt, t_last + 2);
        const int cbase = s * SD_OUT - 1 - SD_HL, c_lo = std::max(cbase, 0), c_hi = std::min(cbase + SD_BP, W) - 1;
        for (int j = tb; j <= te + 3 && j < ndet; j++) {
            const bool in_loop = j >= tb + 4;                               // (the four blocks of the prologue are always loaded)
            if (in_loop && ((T[8 + ((j - 4) >> 5)] >> ((j - 4) & 31)) & 1u)) continue;
            for (int k = c_lo / 64; k <= c_hi / 64; k++) need[(size_t)(j * SD_T / RI_ROWS) * NT + k] = 1;
        }
    }
    int n = 0;
    bool sound = true;
    for (int b = 0; b < nbands; b++) {
        bool needp[RI_GROUPS], zero[RI_GROUPS];
        uint32_t bits[RI_GROUPS];
        for (int g = 0; g < RI_GROUPS; g++) {
            bits[g] = 0; zero[g] = true;
            for (int w = 0; w < RI_WAVES; w++) if (need[(size_t)b * NT + g * RI_WAVES + w]) bits[g] |= 1u << w;
            needp[g] = bits[g] != 0;
            const int rend = std::min(b * RI_ROWS + RI_ROWS - 1, H - 1);
            for (int c = g * 64 * RI_WAVES; c < std::min(W, (g + 1) * 64 * RI_WAVES); c++) if (firstlit[c] <= rend) { zero[g] = false; break; }
        }
        for (int g = 0; g < RI_GROUPS; g++) {
            bool skipL = !needp[g], skipR = true;
            for (int q = 0; q <= g && skipL; q++) skipL = zero[q];
            for (int q = g; q < RI_GROUPS && skipR; q++) skipR = !needp[q];
            if (g * 64 * RI_WAVES >= W) continue;                           // (no such columns)
            if (skipL || skipR) {
                for (int w = 0; w < RI_WAVES; w++) if (lit[(size_t)b * NT + g * RI_WAVES + w]) sound = false;      // (cannot happen: see above)
                continue;
            }
            out[1 + n++] = (uint32_t)b | ((uint32_t)g << 8) | (bits[g] << 12);
        }
    }
    if (!sound) {                                                           // belt and braces: walk everything, write everything
        n = 0;
        for (int b = 0; b < nbands; b++)
            for (int g = 0; g < RI_GROUPS && g * 64 * RI_WAVES < W; g++) out[1 + n++] = (uint32_t)b | ((uint32_t)g << 8) | (((1u << RI_WAVES) - 1u) << 12);
    }
    out[0] = (uint32_t)n;
    return true;
}

hipError_t retrack_init()
{
    if (hipError_t ef = hipFuncSetAttribute(reinterpret_cast<const void *>(rt_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FD_LDS_BYTES); ef != hipSuccess) return ef;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rt_det_strip_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SD_LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(rt_integral_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RI_LDS_BYTES);
}

// determinants + maxima of the first P scratch slots of a chunk
static hipError_t launch_det(hipStream_t st, const RtArgs &a, int first, int P)
{
    const int W = a.W;
    if (a.size1 != 15 || a.size2 != 30) return hipErrorInvalidValue;      // the engine's fixed detector parameters (box sizes are compile-time)
    const int ns = (W + SD_OUT - 1) / SD_OUT;
    hipLaunchKernelGGL(rt_det_strip_kernel, dim3((unsigned)(((int64_t)P * ns + 7) / 8 * 8)), dim3(SD_THREADS), SD_LDS_BYTES, st, a, first, P, ns);
    return hipGetLastError();
}

bool retrack_sided(const RtArgs &a, int B, const RtSide *side)
{
    return side && side->chunk > 0 && a.W <= 2048 && !a.fused && a.slots >= 2 * side->chunk && B > side->chunk;
}

hipError_t launch_retrack(hipStream_t st, const RtArgs &a, int B, hipEvent_t *trace, int ntrace, hipEvent_t after_integral, int after_det, hipEvent_t after_emit, const RtSide *side)
{
    const int W = a.W, R = a.slots;
    // image-scale kernels chunk by chunk (the float64 integral images of `slots` detections are resident at once); the
    // candidate lists are per DETECTION, so the lane-serial bookkeeping runs once over all of them afterwards (round 2 ran it per
    // chunk: 1.4-2.7 ms of a near-idle GPU each time)
    // (the candidate counts are zero on entry: rt_append_kernel, the last kernel of this chain, clears what a detection used)
    hipError_t e = hipSuccess;
    const bool sided = retrack_sided(a, B, side);
    if (sided) {
        // The determinants of chunk c on a second stream BESIDE the integral images of chunk c + 1 (round 6, late): chunks of side->chunk
        // detections, their integral images alternating between the two halves of the scratch.  The two kernels are latency-bound chains
        // at 10 % VALU / LDS utilisation each, and one workgroup of each per CU (80 + 74 KB of LDS) is a better pairing than two of a kind:
        // 70.3-70.7 -> 69.2-69.3 ms per step with chunks of 1 024 (768: nothing, 640: worse); same candidates (profiles/det_side_check.py).
        const int C = side->chunk;
        int nc = 0;
        for (int first = 0; first < B; first += C, nc++) {
            const int P = min(C, B - first);
            const bool tr = trace && nc < ntrace;
            hipEvent_t *tev = trace + 3 * nc;
            RtArgs ac = a;
            ac.S = a.S + (size_t)(nc & 1) * C * a.SP * W;
            if (nc >= 2 && (e = hipStreamWaitEvent(st, side->ev_d[(nc - 2) & 3], 0)) != hipSuccess) return e;      // this bank's determinants are done
            if (tr && (e = hipEventRecord(tev[0], st)) != hipSuccess) return e;
            if (B - first >= RI_MIN_DETECTIONS) hipLaunchKernelGGL(rt_integral_kernel, dim3(P), dim3(64 * (RI_WAVES + 1)), RI_LDS_BYTES, st, ac, first, 0);
            const int P2 = min(P, RT_TWO_PASS_SLOTS);
            hipLaunchKernelGGL(rt_integ_cols_kernel, dim3((W + 63) / 64, (W + RC_BAND - 1) / RC_BAND, min(P2, RT_TWO_PASS_Z)), dim3(256), 0, st, ac, first, P2);
            hipLaunchKernelGGL(rt_integ_rows_kernel, dim3((W + RR_ROWS - 1) / RR_ROWS, min(P2, RT_TWO_PASS_Z)), dim3(256), 0, st, ac, first, P2);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            if (tr && (e = hipEventRecord(tev[1], st)) != hipSuccess) return e;
            if ((e = hipEventRecord(side->ev_i[nc & 3], st)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(side->st, side->ev_i[nc & 3], 0)) != hipSuccess) return e;
            if ((e = launch_det(side->st, ac, first, P)) != hipSuccess) return e;
            if (tr && (e = hipEventRecord(tev[2], side->st)) != hipSuccess) return e;      // (tev[1] .. tev[2]: the determinants, from the moment they could start)
            if ((e = hipEventRecord(side->ev_d[nc & 3], side->st)) != hipSuccess) return e;
        }
        for (int k = max(0, nc - 2); k < nc; k++)
            if ((e = hipStreamWaitEvent(st, side->ev_d[k & 3], 0)) != hipSuccess) return e;
        if (after_integral && after_det <= 1 && (e = hipEventRecord(after_integral, st)) != hipSuccess) return e;
    } else
    for (int first = 0; first < B; first += R) {
        const int P = min(R, B - first);
        const bool tr = trace && first / R < ntrace;
        hipEvent_t *tev = trace + 3 * (first / R);
        if (tr && (e = hipEventRecord(tev[0], st)) != hipSuccess) return e;
        if (W <= 2048 && a.fused && B - first >= RI_MIN_DETECTIONS) hipLaunchKernelGGL(rt_fused_kernel, dim3(P), dim3(FD_THREADS), FD_LDS_BYTES, st, a, first, 0);
        else if (W <= 2048 && B - first >= RI_MIN_DETECTIONS)               // (fewer lanes left than a one-sweep chunk needs: it would return at once)
        {
            // RI_ROUND > 0 (experiment): the chunk's detections in ROUNDS of that many workgroups, one launch each - the workgroups of a
            // round start together and stay in step (every detection is the same work), which the one-buffer form likes
            const int round = RI_ROUND > 0 ? RI_ROUND : P;
            for (int sub = 0; sub < P; sub += round)
                hipLaunchKernelGGL(rt_integral_kernel, dim3(min(round, P - sub)), dim3(64 * (RI_WAVES + 1)), RI_LDS_BYTES, st, a, first, sub);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;              // (a refused launch - LDS attribute, grid - surfaces here, not after the chain)
        // (the two-pass form of chunks below RI_MIN_DETECTIONS detections; its band totals live in a.colT, RT_TWO_PASS_SLOTS entries)
        const int P2 = min(P, RT_TWO_PASS_SLOTS);
        hipLaunchKernelGGL(rt_integ_cols_kernel, dim3((W + 63) / 64, (W + RC_BAND - 1) / RC_BAND, min(P2, RT_TWO_PASS_Z)), dim3(256), 0, st, a, first, P2);
        hipLaunchKernelGGL(rt_integ_rows_kernel, dim3((W + RR_ROWS - 1) / RR_ROWS, min(P2, RT_TWO_PASS_Z)), dim3(256), 0, st, a, first, P2);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (tr && (e = hipEventRecord(tev[1], st)) != hipSuccess) return e;
        if (after_integral && !after_det && first == 0 && (e = hipEventRecord(after_integral, st)) != hipSuccess) return e;      // (the pyramid of a later step may wait for it)
        if ((e = launch_det(st, a, first, P)) != hipSuccess) return e;
        if (after_integral && after_det == 1 && first + R >= B && (e = hipEventRecord(after_integral, st)) != hipSuccess) return e;
        if (tr && (e = hipEventRecord(tev[2], st)) != hipSuccess) return e;
    }
    if (B < 256) {
        RtArgs ab = a;
        ab.blob_order = nullptr;
        hipLaunchKernelGGL(rt_book_kernel, dim3(B), dim3(64), 0, st, ab, 0);                 // K4 + K5 + K6 in one launch
        if ((e = hipGetLastError()) != hipSuccess) return e;
    } else {
        hipLaunchKernelGGL(rt_emit_kernel, dim3(B), dim3(256), 0, st, a, 0);
        static const int emit_where = getenv("ROAM_EMIT_EVENT_WHERE") ? atoi(getenv("ROAM_EMIT_EVENT_WHERE")) : 1;     // 0 after rt_emit, 1 after the ordering (default), 2 after the small bookkeeping class
        if (after_emit && emit_where == 0 && (e = hipEventRecord(after_emit, st)) != hipSuccess) return e;       // (a front-end kernel of a later step may wait for it)
        // the bookkeeping is one latency-bound wavefront per detection and its time grows with the candidate list: longest lists first
        // (in the default step 2.1 -> ... ms for the kernel; the work is the same, the tail is not)
        RtArgs ab = a;
        ab.blob_order = nullptr;
        if (B >= 512 && a.blob_order_buf) {
            if ((e = launch_order_by_count(st, a.cand_n, B, BP_MAX_PTS, a.blob_order_buf, 1)) != hipSuccess) return e;
            ab.blob_order = a.blob_order_buf;
        }
        if (after_emit && emit_where == 1 && (e = hipEventRecord(after_emit, st)) != hipSuccess) return e;
        hipLaunchKernelGGL(rt_blobs_kernel<true>, dim3(B), dim3(64), 0, st, ab, 0);
        if (after_emit && emit_where == 2 && (e = hipEventRecord(after_emit, st)) != hipSuccess) return e;
        hipLaunchKernelGGL(rt_blobs_kernel<false>, dim3(B), dim3(64), 0, st, ab, 0);
        e = launch_ssc_batch(st, a.kp, (int64_t)BP_MAX_PTS * 3, a.kp_n, BP_MAX_PTS, B, 200, 0.1, W, W, a.ssc_work, a.sel, a.sel_n, a.rt_n, 0);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(rt_append_kernel, dim3(B), dim3(256), 0, st, a, 0);
    if (after_integral && after_det == 2 && (e = hipEventRecord(after_integral, st)) != hipSuccess) return e;     // (after the whole chain)
    return hipGetLastError();
}

hipError_t launch_retrack_collect(hipStream_t st, const roam_lane_result *res, const int32_t *scan_idx, int B, int force_all, const RtArgs &a)
{
    hipLaunchKernelGGL(rt_collect_kernel, dim3(1), dim3(256), 0, st, res, scan_idx, B, force_all, a.rt_lane, a.rt_scan, a.rt_n);
    return hipGetLastError();
}

// measurement: re-run the image-scale kernels of the detection for the first `P` scratch slots (their integral images are
// those of the last retracks; rt_n is set by the caller); which: 0 = integral image (cols + rows), 1 = determinants + maxima
hipError_t launch_retrack_emit(hipStream_t st, const RtArgs &a, int P)
{
    hipLaunchKernelGGL(rt_emit_kernel, dim3(P), dim3(256), 0, st, a, 0);
    return hipGetLastError();
}

hipError_t launch_retrack_part(hipStream_t st, const RtArgs &a_in, int P, int which)
{
    RtArgs a = a_in;
    a.fused = which >= 2 ? 1 : 0;                                          // 0 / 1 time and check the two-kernel form whatever the engine runs
    const int W = a.W;
    if (which >= 2) {
        if (!a.fd_mapT) return hipErrorInvalidValue;
        hipLaunchKernelGGL(rt_fused_kernel, dim3(P), dim3(FD_THREADS), FD_LDS_BYTES, st, a, 0, which - 2);
        return hipGetLastError();
    }
    if (which == 0) {
        hipLaunchKernelGGL(rt_integral_kernel, dim3(P), dim3(64 * (RI_WAVES + 1)), RI_LDS_BYTES, st, a, 0, 0);
        const int P2 = min(P, RT_TWO_PASS_SLOTS);
        hipLaunchKernelGGL(rt_integ_cols_kernel, dim3((W + 63) / 64, (W + RC_BAND - 1) / RC_BAND, min(P2, RT_TWO_PASS_Z)), dim3(256), 0, st, a, 0, P2);
        hipLaunchKernelGGL(rt_integ_rows_kernel, dim3((W + RR_ROWS - 1) / RR_ROWS, min(P2, RT_TWO_PASS_Z)), dim3(256), 0, st, a, 0, P2);
    } else {
        return launch_det(st, a, 0, P);                                     // (the caller clears the candidate counts: no bookkeeping follows that would)
    }
    return hipGetLastError();
}
