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

__global__ __launch_bounds__(256) void consistency_graph_kernel(const float *__restrict__ prev,
                                                                const float *__restrict__ next,
                                                                const int32_t *__restrict__ count, int K,
                                                                int kstride, double thr,
                                                                uint64_t *__restrict__ adj, int nw)
{
    const int i = blockIdx.y, b = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + wave;
    const int Kb = count ? count[b] : K;
    if (w >= nw) return;
    uint64_t word = 0;
    if (i < Kb) {
        const int j = w * 64 + lane;
        bool e = false;
        if (j < Kb && j != i) {
            const float *pi = prev + ((int64_t)b * kstride + i) * 2, *pj = prev + ((int64_t)b * kstride + j) * 2;
            const float *ni = next + ((int64_t)b * kstride + i) * 2, *nj = next + ((int64_t)b * kstride + j) * 2;
            double ax = __dsub_rn((double)pi[0], (double)pj[0]), ay = __dsub_rn((double)pi[1], (double)pj[1]);
            double bx = __dsub_rn((double)ni[0], (double)nj[0]), by = __dsub_rn((double)ni[1], (double)nj[1]);
            double d0 = __dsqrt_rn(__dadd_rn(__dmul_rn(ax, ax), __dmul_rn(ay, ay)));
            double d1 = __dsqrt_rn(__dadd_rn(__dmul_rn(bx, bx), __dmul_rn(by, by)));
            e = fabs(__dsub_rn(d0, d1)) <= thr;
        }
        word = __ballot(e);
    }
    if (lane == 0) adj[((int64_t)b * kstride + i) * nw + w] = word;
}

hipError_t launch_consistency_graph(hipStream_t st, const float *prev, const float *next,
                                    const int32_t *count, int K, int kstride, int B, double thr,
                                    uint64_t *adj, int nw)
{
    if (K <= 0 || B <= 0) return hipSuccess;
    dim3 grid((nw + 3) / 4, K, B);
    hipLaunchKernelGGL(consistency_graph_kernel, grid, dim3(256), 0, st, prev, next, count, K, kstride, thr, adj, nw);
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
__device__ int colour_bound(uint64_t P, const uint64_t *A, int nw, int lane, int need)
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
            uint64_t row = (lane < nw) ? A[(int64_t)v * nw + lane] : 0ull;
            uint64_t bv = bit_if(lane, v);
            Q &= ~(row | bv);
            U &= ~bv;
        }
    }
    return c;
}

#define CQ_LDS_ADJ_BYTES 65536

__global__ __launch_bounds__(64) void max_clique_kernel(const uint64_t *__restrict__ adj_g,
                                                        const int32_t *__restrict__ count, int K, int kstride,
                                                        int nw, long long node_limit,
                                                        uint64_t *__restrict__ stack_g,
                                                        uint8_t *__restrict__ mask_out,
                                                        int32_t *__restrict__ n_in, int32_t *__restrict__ flags)
{
    extern __shared__ __align__(16) unsigned char cq_smem[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int Kb = count ? count[b] : K;
    uint8_t *mask = mask_out + (int64_t)b * kstride;
    if (Kb <= 0) {
        if (lane == 0) { n_in[b] = 0; flags[b] = 1; }
        return;
    }
    // ---- LDS carve: deg[kstride] int, vstack[kstride] short, sw[16] u64, adjacency (optional)
    int *deg = reinterpret_cast<int *>(cq_smem);
    short *vstack = reinterpret_cast<short *>(deg + K);
    uint64_t *sw = reinterpret_cast<uint64_t *>(cq_smem + ((sizeof(int) * K + sizeof(short) * K + 15) & ~(size_t)15));
    uint64_t *adj_l = sw + 16;
    const uint64_t *Ag = adj_g + (int64_t)b * kstride * nw;
    const bool use_lds = ((size_t)K * nw * 8 <= CQ_LDS_ADJ_BYTES);
    const uint64_t *A;
    if (use_lds) {
        for (int i = lane; i < Kb * nw; i += 64) adj_l[i] = Ag[i];
        A = adj_l;
    } else
        A = Ag;
    __syncthreads();
    uint64_t *stk = stack_g + (int64_t)b * (kstride + 2) * nw;

    // all-vertices set in word-per-lane layout
    uint64_t ALL = 0;
    if (lane < nw) {
        int lo = lane * 64;
        if (Kb >= lo + 64) ALL = ~0ull;
        else if (Kb > lo) ALL = (1ull << (Kb - lo)) - 1ull;
    }

    // ---- degrees within S (S broadcast through LDS)
    auto degrees = [&](uint64_t S) {
        if (lane < 16) sw[lane] = S;
        __syncthreads();
        for (int u = lane; u < Kb; u += 64) {
            int dsum = 0;
            for (int w = 0; w < nw; w++) dsum += __popcll(A[(int64_t)u * nw + w] & sw[w]);
            deg[u] = dsum;
        }
        __syncthreads();
    };

    // ---- greedy lower bound: peel the minimum-degree vertex (ties: largest index) until clique
    uint64_t S = ALL;
    int sizeS = Kb;
    degrees(S);
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
            if ((A[(int64_t)x * nw + (u0 >> 6)] >> (u0 & 63)) & 1ull) deg[x]--;
        __syncthreads();
    }
    const int LB = sizeS;
    uint64_t REC = S;               // best clique recorded so far (greedy one to start with)
    int best = LB - 1;              // search threshold: find the lexicographically first clique of size >= LB

    // ---- k-core reduction: only vertices with >= LB-1 neighbours inside P can be in such a clique
    uint64_t P = ALL;
    for (;;) {
        degrees(P);
        if (lane < 16) sw[lane] = 0;
        __syncthreads();
        bool any = false;
        // collect removals per word with ballots (vertex u = 64*w + lane)
        uint64_t RM = 0;
        for (int w = 0; w < nw; w++) {
            int u = w * 64 + lane;
            bool rm = (u < Kb) && (deg[u] < LB - 1);
            uint64_t bal = __ballot(rm);
            if (lane == w) RM = bal;
        }
        RM &= P;
        any = __ballot(RM != 0) != 0;
        if (!any) break;
        P &= ~RM;
    }

    // ---- exact search, ascending vertex order
    long long nodes = 0;
    int depth = 0, size = 0;
    uint64_t R = 0;
    bool complete = true;
    if (lane < nw) stk[lane] = P;
    for (;;) {
        uint64_t cand = (lane < nw) ? stk[(int64_t)depth * nw + lane] : 0ull;
        int v = bs_first(cand);
        if (v >= 0 && size + bs_count(cand) <= best) v = -1;          // nothing below can improve
        if (v < 0) {
            if (depth == 0) break;
            depth--;
            size--;
            R &= ~bit_if(lane, (int)vstack[depth]);
            continue;
        }
        cand &= ~bit_if(lane, v);
        if (lane < nw) stk[(int64_t)depth * nw + lane] = cand;
        uint64_t row = (lane < nw) ? A[(int64_t)v * nw + lane] : 0ull;
        uint64_t NP = cand & row;
        const int np = bs_count(NP);
        if (size + 1 + np <= best) continue;
        if (np == 0) {                                              // maximal here and strictly better
            best = size + 1;
            REC = R | bit_if(lane, v);
            continue;
        }
        nodes++;
        if (node_limit > 0 && nodes > node_limit) { complete = false; break; }
        const int need = best - size - 1;                           // need colours(NP) > need
        if (colour_bound(NP, A, nw, lane, need) <= need) continue;
        if (lane == 0) vstack[depth] = (short)v;
        R |= bit_if(lane, v);
        size++;
        depth++;
        if (lane < nw) stk[(int64_t)depth * nw + lane] = NP;
    }

    // ---- emit
    if (lane < 16) sw[lane] = REC;
    __syncthreads();
    for (int u = lane; u < Kb; u += 64) mask[u] = (uint8_t)((sw[u >> 6] >> (u & 63)) & 1ull);
    for (int u = Kb + lane; u < kstride && u < K; u += 64) mask[u] = 0;
    const int cnt = bs_count(REC);
    if (lane == 0) { n_in[b] = cnt; flags[b] = complete ? 1 : 0; }
}

hipError_t launch_max_clique(hipStream_t st, const uint64_t *adj, const int32_t *count, int K,
                             int kstride, int nw, int B, int64_t node_limit, uint64_t *stack,
                             uint8_t *mask, int32_t *n_in, int32_t *flags)
{
    if (B <= 0) return hipSuccess;
    size_t lds = ((sizeof(int) * (size_t)K + sizeof(short) * (size_t)K + 15) & ~(size_t)15) + 16 * 8;
    if ((size_t)K * nw * 8 <= CQ_LDS_ADJ_BYTES) lds += (size_t)K * nw * 8;
    if (node_limit <= 0) node_limit = 300000;
    hipLaunchKernelGGL(max_clique_kernel, dim3(B), dim3(64), lds, st, adj, count, K, kstride, nw,
                       (long long)node_limit, stack, mask, n_in, flags);
    return hipGetLastError();
}
