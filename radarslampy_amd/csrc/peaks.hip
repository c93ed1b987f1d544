// ingest + polar peak extraction (a1 + a2).
//
// Replaces getPointCloud.getPointCloudPolarInd (reference getPointCloud.py:11-54) fused with
// the u8 -> float32 decode of parseData.extractDataFromRadarImage (parseData.py:40,49-51).
//
// Live configuration (u8 record rows, <= 2048 range bins): ONE WAVEFRONT per azimuth row, see
// peaks_rows_u8_wave_kernel below.  Generic path (f32 rows or longer rows):
// one 256-thread workgroup per azimuth row (grid = rows x lanes): the row is staged once
// through LDS (coalesced HBM read of the u8 payload / f32 row), every range bin is tested
// for "left edge of a strict local maximum" (SciPy plateau rule: run of equal samples with
// a strict rise before and a strict fall after, midpoint (l+r)/2, end samples never peaks),
// candidates are compacted in range order with a ballot-free block scan, the float32
// mean / population-std threshold is evaluated with NumPy's pairwise summation order
// (blocks <= 128, 8 accumulators: leaves summed by independent threads, tree combined by
// one) using IEEE round-to-nearest intrinsics (no FMA contraction), survivors are compacted
// again and written as u16 range indices to a per-row staging area.  A second tiny kernel
// turns the per-row counts into offsets and emits the azimuth-major (P,2) int32 list.
//
// HBM traffic per scan: rows*cols bytes read (u8 path) + ~4 B per surviving peak.
#include "roam_internal.h"

#define PK_T 256
#define XI(i) (i)
// (float)k / 255.f for a power code k, exactly (see tests/test_abi_cpu.py::test_u8_decode_identity)
__device__ __forceinline__ float code_to_f32_pk(uint32_t k) { return (float)__dmul_rn((double)k, 1.0 / 255.0); }

__device__ __forceinline__ int block_excl_scan(int v, int *sh, int *total)
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

// NumPy pairwise_sum leaf (n <= 128), float32, round-to-nearest, no contraction
__device__ __forceinline__ float np_leaf_sum(const float *a, int n)
{
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; i++) res = __fadd_rn(res, a[i]);
        return res;
    }
    float r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    const int nn = n - (n & 7);
    for (i = 8; i < nn; i += 8) {
        r0 = __fadd_rn(r0, a[i + 0]); r1 = __fadd_rn(r1, a[i + 1]);
        r2 = __fadd_rn(r2, a[i + 2]); r3 = __fadd_rn(r3, a[i + 3]);
        r4 = __fadd_rn(r4, a[i + 4]); r5 = __fadd_rn(r5, a[i + 5]);
        r6 = __fadd_rn(r6, a[i + 6]); r7 = __fadd_rn(r7, a[i + 7]);
    }
    float res = __fadd_rn(__fadd_rn(__fadd_rn(r0, r1), __fadd_rn(r2, r3)),
                          __fadd_rn(__fadd_rn(r4, r5), __fadd_rn(r6, r7)));
    for (; i < n; i++) res = __fadd_rn(res, a[i]);
    return res;
}

// NumPy's recursion for length n: split at n2 = n/2 - (n/2)%8 while n > 128.  The recursion is
// unrolled at compile time (depth <= 6 covers n <= 8192); n is block-uniform, so the whole walk
// is scalar work that every thread repeats for itself - no LDS stack, no serial thread-0 section
// (the first version kept an explicit stack in LDS: ~12 us of dependent LDS latency per row).
template <int D>
struct PwWalk {
    // leaf number `target` of the in-order leaf sequence -> (my_lo, my_n)
    static __device__ __forceinline__ void select(int lo, int n, int target, int &cnt, int &my_lo, int &my_n)
    {
        if (n <= 128) { if (cnt == target) { my_lo = lo; my_n = n; } cnt++; return; }
        int n2 = n / 2; n2 -= n2 % 8;
        PwWalk<D - 1>::select(lo, n2, target, cnt, my_lo, my_n);
        PwWalk<D - 1>::select(lo + n2, n - n2, target, cnt, my_lo, my_n);
    }
    static __device__ __forceinline__ float combine(const float *leaf_sum, int n, int &li)
    {
        if (n <= 128) return leaf_sum[li++];
        int n2 = n / 2; n2 -= n2 % 8;
        const float a = PwWalk<D - 1>::combine(leaf_sum, n2, li);
        const float b = PwWalk<D - 1>::combine(leaf_sum, n - n2, li);
        return __fadd_rn(a, b);
    }
};
template <>
struct PwWalk<0> {
    static __device__ __forceinline__ void select(int lo, int n, int target, int &cnt, int &my_lo, int &my_n)
    {
        if (cnt == target) { my_lo = lo; my_n = n; }
        cnt++;
    }
    static __device__ __forceinline__ float combine(const float *leaf_sum, int, int &li) { return leaf_sum[li++]; }
};

// block-wide NumPy-ordered float32 sum of a[0..n) (a and leaf_sum in LDS); every thread gets the result
__device__ __forceinline__ float block_np_sum(float *leaf_sum, const float *a, int n)
{
    int cnt = 0, my_lo = 0, my_n = -1;
    PwWalk<6>::select(0, n, (int)threadIdx.x, cnt, my_lo, my_n);
    if (my_n >= 0) leaf_sum[threadIdx.x] = np_leaf_sum(a + my_lo, my_n);
    __syncthreads();
    int li = 0;
    const float r = PwWalk<6>::combine(leaf_sum, n, li);
    __syncthreads();
    return r;
}

template <bool U8>
__global__ __launch_bounds__(PK_T) void peaks_rows_kernel(PeakSrc src, int rows, int cols,
                                                          uint16_t *__restrict__ row_stage, int stage_cap,
                                                          int32_t *__restrict__ row_count)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ float leaf_sum[PK_T];
    __shared__ int scan_sh[8];
    const int half = (cols + 1) / 2;
    float *xs = reinterpret_cast<float *>(smem_raw);          // cols
    float *ph = xs + cols;                                    // half
    float *sq = ph + half;                                    // half
    uint16_t *pm = reinterpret_cast<uint16_t *>(sq + half);   // half

    const int b = blockIdx.y, r = blockIdx.x, t = threadIdx.x;
    const int64_t lane_sel = src.lane_index ? (int64_t)src.lane_index[b] : (int64_t)b;
    if (U8) {
        const uint8_t *p = reinterpret_cast<const uint8_t *>(src.base) + lane_sel * src.lane_stride +
                           (int64_t)r * src.row_stride + src.payload_off;
        for (int i = t; i < cols; i += PK_T) xs[XI(i)] = (float)__dmul_rn((double)p[i], 1.0 / 255.0);   // == (float)k/255.f for all 256 codes
    } else {
        const float *p = reinterpret_cast<const float *>(src.base) + lane_sel * src.lane_stride +
                         (int64_t)r * src.row_stride;
        for (int i = t; i < cols; i += PK_T) xs[XI(i)] = p[i];
    }
    __syncthreads();

    // ---- strict local maxima (plateau rule), contiguous chunk per thread keeps range order
    const int items = (cols + PK_T - 1) / PK_T;
    const int lo = t * items, hi = min(lo + items, cols);
    const int imax = cols - 1;
    int cnt = 0;
    for (int i = max(lo, 1); i < hi && i < imax; i++) {
        float v = xs[XI(i)];
        if (xs[XI(i - 1)] < v) {
            int ia = i + 1;
            while (ia < imax && xs[XI(ia)] == v) ia++;
            if (xs[XI(ia)] < v) cnt++;
        }
    }
    int M;
    int pos = block_excl_scan(cnt, scan_sh, &M);
    for (int i = max(lo, 1); i < hi && i < imax; i++) {
        float v = xs[XI(i)];
        if (xs[XI(i - 1)] < v) {
            int ia = i + 1;
            while (ia < imax && xs[XI(ia)] == v) ia++;
            if (xs[XI(ia)] < v) {
                int mid = (i + ia - 1) >> 1;
                ph[pos] = xs[XI(mid)];
                pm[pos] = (uint16_t)mid;
                pos++;
            }
        }
    }
    __syncthreads();                   // ph / pm complete before the leaf sums read them
    if (M == 0) {                      // numpy: mean of empty = NaN -> nothing passes
        if (t == 0) row_count[b * rows + r] = 0;
        return;
    }
    const float fM = (float)M;
    const float mean = __fdiv_rn(block_np_sum(leaf_sum, ph, M), fM);
    for (int k = t; k < M; k += PK_T) {
        float d = __fsub_rn(ph[k], mean);
        sq[k] = __fmul_rn(d, d);
    }
    __syncthreads();
    const float var = __fdiv_rn(block_np_sum(leaf_sum, sq, M), fM);
    const float thr = __fadd_rn(mean, rn_sqrtf(var));

    // ---- threshold + ordered compaction
    const int kitems = (M + PK_T - 1) / PK_T;
    const int klo = t * kitems, khi = min(klo + kitems, M);
    int c2 = 0;
    for (int k = klo; k < khi; k++) c2 += (ph[k] >= thr) ? 1 : 0;
    int total;
    int p2 = block_excl_scan(c2, scan_sh, &total);
    uint16_t *dst = row_stage + ((int64_t)b * rows + r) * stage_cap;
    for (int k = klo; k < khi; k++)
        if (ph[k] >= thr) {
            if (p2 < stage_cap) dst[p2] = pm[k];
            p2++;
        }
    if (t == 0) row_count[b * rows + r] = total;
}

// ---- specialised row kernel for the live configuration: u8 record rows, cols <= 2048 -----------
// The u8 -> float32 map k -> k/255 is strictly increasing, so maxima / plateaus are found on the
// integer codes and only the surviving candidates are converted.
#define PKF_MAXC 2048
// one wavefront per azimuth row: no workgroup barrier anywhere.
// The first version (256 threads per row, 8 bins per thread) spent its time in instruction issue (PMC: more SALU than VALU
// instructions - eight divergent plateau tests per thread, two block scans, a dozen barriers for
// 2 KB of data).  Here a lane owns 32 consecutive range bins (two 16-byte loads) and the plateau
// rule is evaluated on bit masks: R / E / F = bin is greater than / equal to / less than its left
// neighbour.  A run that starts with a rise at s is a peak iff the first bin after s that differs
// from it is a fall: adding (R << 1) to E lets the carry ripple through the run's equal bins and
// land on that first differing bin, so  PE = (E + (R << 1)) & ~E & F  marks the falls that close a
// peak run; 64-bit masks (own bins + the next lane's, fetched with DPP) cover runs that cross one
// lane boundary, the rare longer run walks the LDS copy of the row.  Candidates, NumPy-ordered
// sums (8 lanes per <=128-element leaf), threshold and both compactions stay inside the wavefront.
#ifndef PKW_WAVES
#define PKW_WAVES 1          // rows (wavefronts) per workgroup: 5 KB LDS workgroups slot in next to any other kernel's
#endif
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int *total)
{
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int n = __shfl_up(inc, d);
        if (lane >= d) inc += n;
    }
    *total = __shfl(inc, 63);
    return inc - v;
}

__global__ __launch_bounds__(64 * PKW_WAVES) void peaks_rows_u8_wave_kernel(PeakSrc src, int rows, int cols,
                                                                             uint16_t *__restrict__ row_stage, int stage_cap,
                                                                             int32_t *__restrict__ row_count)
{
    __shared__ __align__(16) uint8_t xb_s[PKW_WAVES][PKF_MAXC + 64];
    __shared__ __align__(16) uint8_t pc_s[PKW_WAVES][PKF_MAXC / 2];
    __shared__ __align__(16) uint16_t pm_s[PKW_WAVES][PKF_MAXC / 2];
    __shared__ float ls_s[PKW_WAVES][16];
    __shared__ float lut_s[PKW_WAVES][256];                // k -> (float)k / 255.f: the sums decode ~3 values per candidate
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y, r = blockIdx.x * PKW_WAVES + wv;
    if (r >= rows) return;
    uint8_t *xb = xb_s[wv], *pc = pc_s[wv];
    uint16_t *pm = pm_s[wv];
    float *ls = ls_s[wv];
    float *lut = lut_s[wv];
#pragma unroll
    for (int q = 0; q < 4; q++) lut[lane * 4 + q] = code_to_f32_pk((uint32_t)(lane * 4 + q));
    const int64_t lane_sel = src.lane_index ? (int64_t)src.lane_index[b] : (int64_t)b;
    const uint8_t *p = reinterpret_cast<const uint8_t *>(src.base) + lane_sel * src.lane_stride +
                       (int64_t)r * src.row_stride + src.payload_off;
    const int i0 = lane * 32;
    uint32_t w[8];
    // bins past `cols` never enter a mask (vm below) and are never read back from the LDS copy, so a lane may load
    // its 32 bytes whenever they lie inside the record row - for the Oxford layout (3768 payload bytes) that is every
    // lane, including the one that straddles the clip at 2025: no divergent byte-wise tail
    if (i0 + 32 <= (int)src.row_stride - src.payload_off) {
        const u32x4_a1 a = *reinterpret_cast<const u32x4_a1 *>(p + i0), c = *reinterpret_cast<const u32x4_a1 *>(p + i0 + 16);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = c.x; w[5] = c.y; w[6] = c.z; w[7] = c.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t v = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) { const int i = i0 + 4 * j + q; if (i < cols) v |= (uint32_t)p[i] << (8 * q); }
            w[j] = v;
        }
    }
    *reinterpret_cast<uint4 *>(xb + i0) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint4 *>(xb + i0 + 16) = make_uint4(w[4], w[5], w[6], w[7]);
    wave_lds_fence();
    // R / E masks of the own bins (bit k = bin i0 + k against bin i0 + k - 1)
    uint32_t pb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[7], 0x138, 0xf, 0xf, false) >> 24;      // last bin of lane - 1
    uint32_t R = 0, E = 0;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const uint32_t cb = (w[k >> 2] >> (8 * (k & 3))) & 255u;
        R |= (uint32_t)(cb > pb) << k;
        E |= (uint32_t)(cb == pb) << k;
        pb = cb;
    }
    const int nvalid = min(max(cols - i0, 0), 32);
    const uint32_t vm = nvalid >= 32 ? 0xffffffffu : ((1u << nvalid) - 1u);
    R &= vm; E &= vm;
    const uint32_t F = ~(R | E) & vm;
    if (lane == 0) R &= ~1u;                                   // bin 0 has no left neighbour: never a start
    const uint32_t En = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)E, 0x130, 0xf, 0xf, false);          // masks of lane + 1 (0 past the row)
    const uint32_t Fn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)F, 0x130, 0xf, 0xf, false);
    const unsigned long long E64 = (unsigned long long)E | ((unsigned long long)En << 32);
    const unsigned long long F64 = (unsigned long long)F | ((unsigned long long)Fn << 32);
    const unsigned long long Z64 = E64 + ((unsigned long long)R << 1);
    unsigned long long pe = Z64 & ~E64 & F64;
    int cnt = __popcll(pe);
    int extra_mid = -1;
    uint32_t extra_c = 0;
    if (Z64 < E64) {                                           // the last run of this lane runs past the next lane too
        const int sl = 31 - __clz((int)R);
        extra_c = xb[i0 + sl];
        int i = i0 + 64;
        while (i < cols && (uint32_t)xb[i] == extra_c) i++;
        if (i < cols && (uint32_t)xb[i] < extra_c) { extra_mid = (i0 + sl + i - 1) >> 1; cnt++; }
    }
    int M;
    int pos = wave_excl_scan(cnt, lane, &M);
    if (M == 0) {
        if (lane == 0) row_count[b * rows + r] = 0;
        return;
    }
    while (pe) {
        const int q = __ffsll((long long)pe) - 1;
        pe &= pe - 1;
        const uint32_t below = q >= 32 ? R : (R & ((1u << q) - 1u));
        const int sl = 31 - __clz((int)below);                 // the rise that opened the run closed at q
        pc[pos] = xb[i0 + sl];
        pm[pos] = (uint16_t)(i0 + ((sl + q - 1) >> 1));
        pos++;
    }
    if (extra_mid >= 0) { pc[pos] = (uint8_t)extra_c; pm[pos] = (uint16_t)extra_mid; }
    wave_lds_fence();

    // NumPy pairwise sums: 8 lanes per leaf (one per accumulator), up to 16 leaves in two passes
    int nleaf = 0, lo1 = 0, n1 = -1, lo2 = 0, n2 = -1;
    PwWalk<4>::select(0, M, lane >> 3, nleaf, lo1, n1);
    if (nleaf > 8) { int c2 = 0; PwWalk<4>::select(0, M, 8 + (lane >> 3), c2, lo2, n2); }
    const int j8 = lane & 7;
    auto leaf = [&](int lo, int n, auto val) -> float {
        float res = 0.f;
        if (n >= 8) {
            float rj = val(lo + j8);
            const int nn = n - (n & 7);
            for (int i = 8; i < nn; i += 8) rj = __fadd_rn(rj, val(lo + i + j8));
            rj = __fadd_rn(rj, __shfl_xor(rj, 1));
            rj = __fadd_rn(rj, __shfl_xor(rj, 2));
            rj = __fadd_rn(rj, __shfl_xor(rj, 4));
            res = rj;
            if (j8 == 0) for (int i = nn; i < n; i++) res = __fadd_rn(res, val(lo + i));
        } else if (n >= 0) {
            if (j8 == 0) for (int i = 0; i < n; i++) res = __fadd_rn(res, val(lo + i));    // short leaf (M < 8)
        }
        return res;
    };
    auto np_sum = [&](auto val) -> float {
        const float r1 = leaf(lo1, n1, val);
        if (n1 >= 0 && j8 == 0) ls[lane >> 3] = r1;
        if (nleaf > 8) {
            const float r2 = leaf(lo2, n2, val);
            if (n2 >= 0 && j8 == 0) ls[8 + (lane >> 3)] = r2;
        }
        wave_lds_fence();
        int li = 0;
        const float tot = PwWalk<4>::combine(ls, M, li);
        wave_lds_fence();
        return tot;
    };
    const float fM = (float)M;
    const float mean = __fdiv_rn(np_sum([&](int k) { return lut[pc[k]]; }), fM);
    const float var = __fdiv_rn(np_sum([&](int k) { const float d = __fsub_rn(lut[pc[k]], mean); return __fmul_rn(d, d); }), fM);
    const float thr = __fadd_rn(mean, rn_sqrtf(var));
    // threshold + ordered compaction: 16 consecutive candidates per lane (M <= 1024)
    const int k0 = lane * 16;
    const uint4 cw = *reinterpret_cast<const uint4 *>(pc + k0);
    const uint32_t cws[4] = {cw.x, cw.y, cw.z, cw.w};
    uint32_t keep = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const uint32_t c = (cws[q >> 2] >> (8 * (q & 3))) & 255u;
        keep |= (uint32_t)((k0 + q < M) && (lut[c] >= thr)) << q;
    }
    int total;
    int p2 = wave_excl_scan(__popc(keep), lane, &total);
    uint16_t *dst = row_stage + ((int64_t)b * rows + r) * stage_cap;
    while (keep) {
        const int q = __ffs((int)keep) - 1;
        keep &= keep - 1;
        if (p2 < stage_cap) dst[p2] = pm[k0 + q];
        p2++;
    }
    if (lane == 0) row_count[b * rows + r] = total;
}

// per lane: exclusive scan of row counts, then emit (az, rng) pairs azimuth-major
__global__ __launch_bounds__(PK_T) void peaks_gather_kernel(const uint16_t *__restrict__ row_stage, int stage_cap,
                                                            const int32_t *__restrict__ row_count, int rows,
                                                            int32_t *__restrict__ out, int cap,
                                                            int32_t *__restrict__ n_out)
{
    extern __shared__ int offs[];     // rows + 1
    __shared__ int scan_sh[8];
    const int b = blockIdx.x, t = threadIdx.x;
    const int ritems = (rows + PK_T - 1) / PK_T;
    const int lo = t * ritems, hi = min(lo + ritems, rows);
    int c = 0;
    for (int r = lo; r < hi; r++) c += min(row_count[b * rows + r], stage_cap);
    int total;
    int pos = block_excl_scan(c, scan_sh, &total);
    for (int r = lo; r < hi; r++) { offs[r] = pos; pos += min(row_count[b * rows + r], stage_cap); }
    if (t == 0) { offs[rows] = total; n_out[b] = total; }
    __syncthreads();
    const int lane = t & 63, w = t >> 6, nw = PK_T >> 6;
    int32_t *o = out + (int64_t)b * cap * 2;
    for (int r = w; r < rows; r += nw) {
        const int off = offs[r], n = offs[r + 1] - off;
        const uint16_t *s = row_stage + ((int64_t)b * rows + r) * stage_cap;
        for (int j = lane; j < n; j += 64) {
            int q = off + j;
            if (q < cap) { o[2 * q] = r; o[2 * q + 1] = (int)s[j]; }
        }
    }
}

hipError_t launch_peaks(hipStream_t st, PeakSrc src, int B, int rows, int cols, uint16_t *row_stage,
                        int stage_cap, int32_t *row_count, int32_t *out, int32_t cap, int32_t *n_out)
{
    const int half = (cols + 1) / 2;
    size_t lds = sizeof(float) * (size_t)(cols + cols / 8 + 2 + 2 * half) + sizeof(uint16_t) * (size_t)half + 16;
    dim3 grid(rows, B), block(PK_T);
    if (src.is_u8 && cols <= PKF_MAXC && cols >= 3)
        hipLaunchKernelGGL(peaks_rows_u8_wave_kernel, dim3((rows + PKW_WAVES - 1) / PKW_WAVES, B), dim3(64 * PKW_WAVES), 0, st, src, rows,
                           cols, row_stage, stage_cap, row_count);
    else if (src.is_u8)
        hipLaunchKernelGGL(peaks_rows_kernel<true>, grid, block, lds, st, src, rows, cols, row_stage, stage_cap, row_count);
    else
        hipLaunchKernelGGL(peaks_rows_kernel<false>, grid, block, lds, st, src, rows, cols, row_stage, stage_cap, row_count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(peaks_gather_kernel, dim3(B), dim3(PK_T), sizeof(int) * (size_t)(rows + 1), st,
                       row_stage, stage_cap, row_count, rows, out, cap, n_out);
    return hipGetLastError();
}
