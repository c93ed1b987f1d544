// Outlier rejection: pairwise-distance consistency graph + maximum clique (a8).
//
// Replaces outlierRejection.rejectOutliers (reference outlierRejection.py:16-95):
//   A[i][j] = | ||p_i-p_j|| - ||n_i-n_j|| | <= thr  in float64 (scipy cdist), inliers = a
//   maximum clique (networkx.find_cliques in the reference).  Result contract: the
//   lexicographically smallest maximum clique (oracle/c/clique.c explains why the
//   reference's own tie-break is not reproducible); equal to the reference's set whenever
//   the maximum clique is unique, equal in size always (when the search completes).
//
// consistency_graph_kernel: one wavefront per 64 columns of one row; the 64 predicates
//   become one adjacency word through a wave ballot.  float64 with explicit round-to-
//   nearest intrinsics (no FMA contraction) => bit-identical to scipy's arithmetic.
// max_clique_kernel: ONE WAVEFRONT PER PROBLEM.  Bitsets (<= 16 words for K <= 1024) are
//   held one word per lane; set algebra is one VALU op, population counts and first-set
//   searches are ballot / shuffle reductions.  Adjacency rows live in LDS when they fit
//   (K*nw*8 <= 64 KB) and are read through L2 otherwise.  Search = greedy minimum-degree
//   peeling for a lower bound, k-core reduction, then an exact depth-first branch and
//   bound in ascending vertex order with a greedy-colouring upper bound; visiting cliques
//   in lexicographic order and accepting only strict improvements yields the
//   lexicographically smallest maximum clique.  The DFS stack sits in a global scratch
//   slab (nw words per level, L2-resident).  No MFMA: the work is integer bit algebra.
#include "roam_internal.h"

#define CG_ROWS 8       // adjacency rows per workgroup
__global__ __launch_bounds__(256) void consistency_graph_kernel(const float *__restrict__ prev,
                                                                const float *__restrict__ next,
                                                                const int32_t *__restrict__ count, int K,
                                                                int kstride, double thr,
                                                                uint64_t *__restrict__ adj, int nw, int nws)
{
    const int i0 = blockIdx.x * CG_ROWS, b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int Kb = count ? min(count[b], K) : K;
    if (i0 >= Kb) return;                              // the host bound K only shrinks at (re)seeds: most rows are past the live count
    const int nwb = (Kb + 63) >> 6;                    // words that can hold a set bit for this lane
    const float *P = prev + (int64_t)b * kstride * 2, *N = next + (int64_t)b * kstride * 2;
    for (int job = wave; job < CG_ROWS * nwb; job += 4) {
        const int i = i0 + job / nwb, w = job % nwb;
        if (i >= Kb) continue;
        const int j = w * 64 + lane;
        bool e = false;
        if (j < Kb && j != i) {
            double ax = __dsub_rn((double)P[2 * i], (double)P[2 * j]), ay = __dsub_rn((double)P[2 * i + 1], (double)P[2 * j + 1]);
            double bx = __dsub_rn((double)N[2 * i], (double)N[2 * j]), by = __dsub_rn((double)N[2 * i + 1], (double)N[2 * j + 1]);
            double d0 = __dsqrt_rn(__dadd_rn(__dmul_rn(ax, ax), __dmul_rn(ay, ay)));
            double d1 = __dsqrt_rn(__dadd_rn(__dmul_rn(bx, bx), __dmul_rn(by, by)));
            e = fabs(__dsub_rn(d0, d1)) <= thr;
        }
        const uint64_t word = __ballot(e);
        if (lane == 0) adj[((int64_t)b * kstride + i) * nws + w] = word;
    }
}

hipError_t launch_consistency_graph(hipStream_t st, const float *prev, const float *next,
                                    const int32_t *count, int K, int kstride, int B, double thr,
                                    uint64_t *adj, int nws)
{
    if (K <= 0 || B <= 0) return hipSuccess;
    const int nw = (K + 63) / 64;
    dim3 grid((K + CG_ROWS - 1) / CG_ROWS, B);
    hipLaunchKernelGGL(consistency_graph_kernel, grid, dim3(256), 0, st, prev, next, count, K, kstride, thr, adj, nw, nws);
    return hipGetLastError();
}

// ------------------------------------------------------------------ wave-level bitset helpers
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
    int lo = __shfl((int)(v & 0xffffffffull), src);
    int hi = __shfl((int)(v >> 32), src);
    return ((uint64_t)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ int bs_count(uint64_t x) { return wave_sum_i(__popcll(x)); }
__device__ __forceinline__ int bs_first(uint64_t x)
{
    uint64_t bal = __ballot(x != 0);
    if (!bal) return -1;
    int fl = __ffsll((long long)bal) - 1;
    uint64_t wv = shfl64(x, fl);
    return fl * 64 + (__ffsll((long long)wv) - 1);
}
__device__ __forceinline__ uint64_t bit_if(int lane, int v) { return (lane == (v >> 6)) ? (1ull << (v & 63)) : 0ull; }

// number of colour classes of a greedy sequential colouring of P, stopping early once it
// exceeds `need` (the caller only asks whether colours(P) > need)
__device__ int colour_bound(uint64_t P, const uint64_t *A, int as, int nw, int lane, int need)
{
    uint64_t U = P;
    int c = 0;
    while (__ballot(U != 0)) {
        c++;
        if (c > need) return c;
        uint64_t Q = U;
        for (;;) {
            int v = bs_first(Q);
            if (v < 0) break;
            uint64_t row = (lane < nw) ? A[(int64_t)v * as + lane] : 0ull;
            uint64_t bv = bit_if(lane, v);
            Q &= ~(row | bv);
            U &= ~bv;
        }
    }
    return c;
}

#define CQ_LDS_ADJ_WORDS 8192          // 64 KB of adjacency rows in LDS (compact stride)

// Node reduction (wave-parallel over vertices, 64 at a time):
//   * k-core: a vertex with fewer than (best - size) neighbours inside NP cannot belong to a
//     clique that beats `best` -> removed, iterated to a fixed point;
//   * universal vertices (adjacent to every other vertex of NP) belong to EVERY maximum clique
//     of this subproblem -> moved into R at once.  Adding elements common to all candidates
//     does not change their lexicographic order, so the canonical result is preserved.
// Returns |NP| after reduction, or -1 when the subproblem cannot beat `best`.
__device__ int reduce_node(uint64_t &NP, uint64_t &R, int &size, int best, const uint64_t *A, int as,
                           int nw, int lane, uint64_t *sw)
{
    for (;;) {
        const int cnt = bs_count(NP);
        if (size + cnt <= best) return -1;
        if (cnt == 0) return 0;
        if (lane < 16) sw[lane] = NP;
        __syncthreads();
        const int thr = best - size;
        uint64_t RM = 0, UN = 0;
        for (int g = 0; g < nw; g++) {
            const int u = g * 64 + lane;
            const bool in = (sw[g] >> lane) & 1ull;
            int d = 0;
            if (in)
                for (int w = 0; w < nw; w++) d += __popcll(A[(int64_t)u * as + w] & sw[w]);
            const uint64_t brm = __ballot(in && d < thr);
            const uint64_t bun = __ballot(in && d == cnt - 1);
            if (lane == g) { RM = brm; UN = bun; }
        }
        __syncthreads();
        if (__ballot(RM != 0)) { NP &= ~RM; continue; }
        if (__ballot(UN != 0)) {
            R |= UN;
            size += bs_count(UN);
            NP &= ~UN;
            return bs_count(NP);
        }
        return cnt;
    }
}

__global__ __launch_bounds__(64) void max_clique_kernel(const uint64_t *__restrict__ adj_g,
                                                        const int32_t *__restrict__ count, int K, int kstride,
                                                        int nws, long long node_limit,
                                                        uint64_t *__restrict__ stack_g,
                                                        uint8_t *__restrict__ mask_out,
                                                        int32_t *__restrict__ n_in, int32_t *__restrict__ flags)
{
    extern __shared__ __align__(16) unsigned char cq_smem[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int Kb = count ? min(count[b], K) : K;
    uint8_t *mask = mask_out + (int64_t)b * kstride;
    if (Kb <= 0) {
        if (lane == 0) { n_in[b] = 0; flags[b] = 1; }
        return;
    }
    const int nw = (Kb + 63) >> 6;                 // active words per bitset
    // ---- LDS carve: deg[K] int, lsize[K+2] short, sw[16] u64, adjacency (compact stride nw)
    int *deg = reinterpret_cast<int *>(cq_smem);
    short *lsize = reinterpret_cast<short *>(deg + K);
    uint64_t *sw = reinterpret_cast<uint64_t *>(cq_smem + ((sizeof(int) * K + sizeof(short) * (K + 2) + 15) & ~(size_t)15));
    uint64_t *adj_l = sw + 16;
    const uint64_t *Ag = adj_g + (int64_t)b * kstride * nws;
    const bool use_lds = (Kb * nw <= CQ_LDS_ADJ_WORDS);
    const uint64_t *A;
    int as;
    if (use_lds) {
        for (int i = lane; i < Kb * nw; i += 64) {
            const int r = i / nw, w = i - r * nw;
            adj_l[i] = Ag[(int64_t)r * nws + w];
        }
        A = adj_l; as = nw;
    } else { A = Ag; as = nws; }
    __syncthreads();
    uint64_t *stk = stack_g + (int64_t)b * (kstride + 2) * 2 * nws;

    // all-vertices set in word-per-lane layout
    uint64_t ALL = 0;
    if (lane < nw) {
        int lo = lane * 64;
        if (Kb >= lo + 64) ALL = ~0ull;
        else if (Kb > lo) ALL = (1ull << (Kb - lo)) - 1ull;
    }

    // ---- greedy lower bound: peel the minimum-degree vertex (ties: largest index) until clique
    uint64_t S = ALL;
    int sizeS = Kb;
    if (lane < 16) sw[lane] = S;
    __syncthreads();
    for (int u = lane; u < Kb; u += 64) {
        int dsum = 0;
        for (int w = 0; w < nw; w++) dsum += __popcll(A[(int64_t)u * as + w] & sw[w]);
        deg[u] = dsum;
    }
    __syncthreads();
    for (;;) {
        if (lane < 16) sw[lane] = S;
        __syncthreads();
        int key = 0x7fffffff;
        for (int u = lane; u < Kb; u += 64)
            if ((sw[u >> 6] >> (u & 63)) & 1ull) key = min(key, deg[u] * 2048 + (2047 - u));
        key = wave_min_i(key);
        const int dmin = key >> 11, u0 = 2047 - (key & 2047);
        if (dmin >= sizeS - 1) break;
        S &= ~bit_if(lane, u0);
        sizeS--;
        __syncthreads();
        for (int x = lane; x < Kb; x += 64)
            if ((A[(int64_t)x * as + (u0 >> 6)] >> (u0 & 63)) & 1ull) deg[x]--;
        __syncthreads();
    }
    const int LB = sizeS;
    uint64_t REC = S;               // best clique recorded so far (the greedy one to start with)
    int best = LB - 1;              // threshold: find the lexicographically first clique of size >= LB

    // ---- exact search in ascending vertex order (levels hold {untried candidates, R, |R|})
    long long nodes = 0;
    int depth = -1;
    bool complete = true;
    uint64_t NP = ALL, Rn = 0;
    int sz = 0;
    bool pending = true;            // (NP, Rn, sz) is a subproblem waiting to be examined
    for (;;) {
        if (pending) {
            pending = false;
            const int c = reduce_node(NP, Rn, sz, best, A, as, nw, lane, sw);
            if (c == 0) { if (sz > best) { best = sz; REC = Rn; } }
            else if (c > 0) {
                nodes++;
                if (node_limit > 0 && nodes > node_limit) { complete = false; break; }
                const int need = best - sz;                      // need colours(NP) > need
                if (colour_bound(NP, A, as, nw, lane, need) > need) {
                    depth++;
                    if (lane < nw) {
                        stk[((int64_t)depth * 2) * nws + lane] = NP;
                        stk[((int64_t)depth * 2 + 1) * nws + lane] = Rn;
                    }
                    if (lane == 0) lsize[depth] = (short)sz;
                }
            }
            continue;
        }
        if (depth < 0) break;
        uint64_t cand = (lane < nw) ? stk[((int64_t)depth * 2) * nws + lane] : 0ull;
        const int sl = (int)lsize[depth];
        const int v = bs_first(cand);
        if (v < 0 || sl + bs_count(cand) <= best) { depth--; continue; }
        cand &= ~bit_if(lane, v);
        if (lane < nw) stk[((int64_t)depth * 2) * nws + lane] = cand;
        const uint64_t Rl = (lane < nw) ? stk[((int64_t)depth * 2 + 1) * nws + lane] : 0ull;
        const uint64_t row = (lane < nw) ? A[(int64_t)v * as + lane] : 0ull;
        NP = cand & row;
        Rn = Rl | bit_if(lane, v);
        sz = sl + 1;
        pending = true;
    }

    // ---- emit
    if (lane < 16) sw[lane] = REC;
    __syncthreads();
    for (int u = lane; u < Kb; u += 64) mask[u] = (uint8_t)((sw[u >> 6] >> (u & 63)) & 1ull);
    for (int u = Kb + lane; u < kstride && u < K; u += 64) mask[u] = 0;
    const int cnt = bs_count(REC);
    if (lane == 0) { n_in[b] = cnt; flags[b] = complete ? 1 : 0; }
}

// adj rows have stride nws words; stack scratch: B x (kstride+2) x 2 x nws words
hipError_t launch_max_clique(hipStream_t st, const uint64_t *adj, const int32_t *count, int K,
                             int kstride, int nws, int B, int64_t node_limit, uint64_t *stack,
                             uint8_t *mask, int32_t *n_in, int32_t *flags)
{
    if (B <= 0 || K <= 0) return hipSuccess;
    size_t lds = ((sizeof(int) * (size_t)K + sizeof(short) * (size_t)(K + 2) + 15) & ~(size_t)15) + 16 * 8;
    const size_t kw = (size_t)K * ((K + 63) / 64);
    lds += 8 * (kw < CQ_LDS_ADJ_WORDS ? kw : (size_t)CQ_LDS_ADJ_WORDS);
    if (node_limit <= 0) node_limit = 300000;
    hipLaunchKernelGGL(max_clique_kernel, dim3(B), dim3(64), lds, st, adj, count, K, kstride, nws,
                       (long long)node_limit, stack, mask, n_in, flags);
    return hipGetLastError();
}
