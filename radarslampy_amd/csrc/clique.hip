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
//   (K*nw*8 <= 64 KB) and are read through L2 otherwise.  Search (see cq_solve): the graph is
//   dense, its complement sparse, so omega comes from a reduction-driven branch and bound on the
//   conflict graph (k-core, universal and pendant vertices, matching bound, branching on the most
//   conflicting vertex); the lexicographically smallest maximum clique then follows by fixing
//   vertices in ascending order with bounded existence queries.  The DFS stack sits in a global
//   scratch slab (nw words per level, L2-resident).  No MFMA: the work is integer bit algebra.
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

#define CQ_LDS_ADJ_WORDS 8192          // 64 KB of adjacency rows in LDS (compact stride)

// ---------------------------------------------------------------------------------------------- exact search
// The consistency graph of a scan pair is DENSE: the static features form one large clique and every other vertex is
// adjacent to most of it (median degree 217 of 239 on a 240-feature pair right after a re-detection; maximum clique 155).
// Its complement H - the "conflict" graph - is sparse, so the search is organised as a maximum-independent-set solver on
// H: per node a wave-parallel degree pass (64 vertices at a time, one popcount per adjacency word) drives
//   * k-core:     fewer than (best - size) neighbours inside P            -> cannot be in an improving clique, removed
//   * universal:  no conflict left inside P (H-degree 0)                  -> in every maximum clique of P, taken
//   * pendant:    exactly one conflict u (H-degree 1)                     -> SOME maximum clique of P contains it: taken, u removed
// to a fixed point; what remains is bounded by |P| - |greedy maximal matching of H[P]| (each conflict edge of a matching
// costs one vertex) and branched on the vertex with the most conflicts: first WITHOUT it (this descent is the classical
// minimum-degree peeling and finds a near-optimal clique at once), then with it.
// cq_solve answers "size of a maximum clique inside P0, if it exceeds `best`" and stops early at `target`.
struct CqCtx {
    const uint64_t *A;          // adjacency rows (LDS or global), stride `as` words
    int as, nw, nws, lane;
    uint64_t *sw;               // 16 words of LDS: the current set broadcast to all lanes
    uint64_t *stk;              // global scratch: 2 * nws words per level
    short *lsize, *lv, *lstage; // per level: |R|, branching vertex, stage
    long long nodes, node_limit;
    bool complete;
};

__device__ int cq_solve(CqCtx &c, uint64_t P0, int best, int target, uint64_t &REC)
{
    const int lane = c.lane, nw = c.nw;
    uint64_t P = P0, R = 0;
    int size = 0, depth = -1;
    bool pending = true;
    for (;;) {
        if (pending) {
            pending = false;
            int vbr = -1;
            bool dead = false;
            for (;;) {                                                  // reductions to a fixed point
                const int cnt = bs_count(P);
                if (size + cnt <= best) { dead = true; break; }
                if (cnt == 0) break;
                if (lane < 16) c.sw[lane] = P;
                __syncthreads();
                const int thr = best - size;                            // an improving clique needs >= thr neighbours inside P
                uint64_t RM = 0, UN = 0, PE = 0;
                int key = 0x7fffffff;
                for (int g = 0; g < nw; g++) {
                    const int u = g * 64 + lane;
                    const bool in = (c.sw[g] >> lane) & 1ull;
                    int d = 0;
                    if (in)
                        for (int w = 0; w < nw; w++) d += __popcll(c.A[(int64_t)u * c.as + w] & c.sw[w]);
                    const uint64_t brm = __ballot(in && d < thr);
                    const uint64_t bun = __ballot(in && d == cnt - 1);
                    const uint64_t bpe = __ballot(in && d == cnt - 2);
                    if (lane == g) { RM = brm; UN = bun; PE = bpe; }
                    if (in) key = min(key, d * 2048 + (2047 - u));      // fewest neighbours = most conflicts; ties: largest index
                }
                __syncthreads();
                if (__ballot(RM != 0)) { P &= ~RM; continue; }
                if (__ballot(UN != 0)) { R |= UN; size += bs_count(UN); P &= ~UN; continue; }
                const int pv = bs_first(PE);
                if (pv >= 0) {
                    const uint64_t row = (lane < nw) ? c.A[(int64_t)pv * c.as + lane] : 0ull;
                    const uint64_t bv = bit_if(lane, pv);
                    const int pu = bs_first(P & ~row & ~bv);           // its only conflict
                    R |= bv; size++;
                    P &= ~(bv | bit_if(lane, pu));
                    continue;
                }
                key = wave_min_i(key);
                vbr = 2047 - (key & 2047);
                break;
            }
            if (dead) continue;
            if (bs_count(P) == 0) {                                     // a maximal clique of this branch
                if (size > best) { best = size; REC = R; if (best >= target) return best; }
                continue;
            }
            c.nodes++;
            if (c.node_limit > 0 && c.nodes > c.node_limit) { c.complete = false; return best; }
            // matching bound on the conflict graph of P
            {
                const int cnt = bs_count(P);
                uint64_t Q = P;
                int slack = size + cnt - best;                          // prune once the matching reaches `slack`
                bool pruned = false;
                for (;;) {
                    const int v = bs_first(Q);
                    if (v < 0) break;
                    const uint64_t row = (lane < nw) ? c.A[(int64_t)v * c.as + lane] : 0ull;
                    const uint64_t bv = bit_if(lane, v);
                    const int u = bs_first(Q & ~row & ~bv);
                    Q &= ~bv;
                    if (u < 0) continue;
                    Q &= ~bit_if(lane, u);
                    if (--slack <= 0) { pruned = true; break; }
                }
                if (pruned) continue;
            }
            depth++;
            if (lane < nw) {
                c.stk[((int64_t)depth * 2) * c.nws + lane] = P;
                c.stk[((int64_t)depth * 2 + 1) * c.nws + lane] = R;
            }
            if (lane == 0) { c.lsize[depth] = (short)size; c.lv[depth] = (short)vbr; c.lstage[depth] = 0; }
            __syncthreads();
            continue;
        }
        if (depth < 0) return best;
        const int stage = c.lstage[depth], v = c.lv[depth], sl = c.lsize[depth];
        __syncthreads();
        if (stage >= 2) { depth--; continue; }
        if (lane == 0) c.lstage[depth] = (short)(stage + 1);
        const uint64_t Pl = (lane < nw) ? c.stk[((int64_t)depth * 2) * c.nws + lane] : 0ull;
        const uint64_t Rl = (lane < nw) ? c.stk[((int64_t)depth * 2 + 1) * c.nws + lane] : 0ull;
        if (sl + bs_count(Pl) <= best) { depth--; continue; }
        const uint64_t bv = bit_if(lane, v);
        if (stage == 0) { P = Pl & ~bv; R = Rl; size = sl; }           // without the most conflicting vertex
        else {
            const uint64_t row = (lane < nw) ? c.A[(int64_t)v * c.as + lane] : 0ull;
            P = Pl & row; R = Rl | bv; size = sl + 1;                   // with it: its conflicts leave
        }
        pending = true;
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
    // ---- LDS carve: three short[K+2] level arrays, sw[16] u64, adjacency (compact stride)
    short *lsize = reinterpret_cast<short *>(cq_smem);
    short *lv = lsize + (K + 2), *lstage = lv + (K + 2);
    uint64_t *sw = reinterpret_cast<uint64_t *>(cq_smem + ((3 * sizeof(short) * (K + 2) + 15) & ~(size_t)15));
    uint64_t *adj_l = sw + 16;
    const uint64_t *Ag = adj_g + (int64_t)b * kstride * nws;
    const bool use_lds = (Kb * nw <= CQ_LDS_ADJ_WORDS);
    CqCtx c;
    if (use_lds) {
        for (int i = lane; i < Kb * nw; i += 64) {
            const int r = i / nw, w = i - r * nw;
            adj_l[i] = Ag[(int64_t)r * nws + w];
        }
        c.A = adj_l; c.as = nw;
    } else { c.A = Ag; c.as = nws; }
    __syncthreads();
    c.nw = nw; c.nws = nws; c.lane = lane; c.sw = sw;
    c.stk = stack_g + (int64_t)b * (kstride + 2) * 2 * nws;
    c.lsize = lsize; c.lv = lv; c.lstage = lstage;
    c.nodes = 0; c.node_limit = node_limit; c.complete = true;

    uint64_t ALL = 0;                               // all-vertices set in word-per-lane layout
    if (lane < nw) {
        const int lo = lane * 64;
        if (Kb >= lo + 64) ALL = ~0ull;
        else if (Kb > lo) ALL = (1ull << (Kb - lo)) - 1ull;
    }
    // ---- phase 1: omega and one maximum clique (the witness)
    uint64_t WIT = 0;
    const int omega = cq_solve(c, ALL, 0, 0x7fffffff, WIT);
    // ---- phase 2: the lexicographically smallest maximum clique.  Vertices are fixed in ascending order; v is taken iff
    // a clique of the size still needed exists among the common neighbours above it.  The witness (a maximum clique that
    // extends the choices made so far) answers "yes" for its own members without a search; every other vertex costs one
    // bounded existence query, and a successful query replaces the witness.
    uint64_t C = ALL, RF = 0;
    int need = omega;
    while (need > 0 && c.complete) {
        const int v = bs_first(C);
        if (v < 0) break;                                               // cannot happen for an exact omega
        const uint64_t bv = bit_if(lane, v);
        const uint64_t row = (lane < nw) ? c.A[(int64_t)v * c.as + lane] : 0ull;
        const uint64_t S = C & row;                                    // C holds only vertices above the last decision
        const bool inwit = __ballot((WIT & bv) != 0) != 0;
        bool take = inwit;
        if (!take && bs_count(S) >= need - 1) {
            if (need == 1) { take = true; WIT = RF | bv; }
            else {
                uint64_t RQ = 0;
                const int got = cq_solve(c, S, need - 2, need - 1, RQ);
                if (got >= need - 1) { take = true; WIT = RF | bv | RQ; }
            }
        }
        if (take) { RF |= bv; C = S; need--; }
        else C &= ~bv;
    }
    const uint64_t REC = (c.complete && need == 0) ? RF : WIT;          // incomplete search: the best clique known

    // ---- emit
    if (lane < 16) sw[lane] = REC;
    __syncthreads();
    for (int u = lane; u < Kb; u += 64) mask[u] = (uint8_t)((sw[u >> 6] >> (u & 63)) & 1ull);
    for (int u = Kb + lane; u < kstride && u < K; u += 64) mask[u] = 0;
    const int cnt = bs_count(REC);
    if (lane == 0) { n_in[b] = cnt; flags[b] = c.complete ? 1 : 0; }
}

// adj rows have stride nws words; stack scratch: B x (kstride+2) x 2 x nws words
hipError_t launch_max_clique(hipStream_t st, const uint64_t *adj, const int32_t *count, int K,
                             int kstride, int nws, int B, int64_t node_limit, uint64_t *stack,
                             uint8_t *mask, int32_t *n_in, int32_t *flags)
{
    if (B <= 0 || K <= 0) return hipSuccess;
    size_t lds = ((3 * sizeof(short) * (size_t)(K + 2) + 15) & ~(size_t)15) + 16 * 8;
    const size_t kw = (size_t)K * ((K + 63) / 64);
    lds += 8 * (kw < CQ_LDS_ADJ_WORDS ? kw : (size_t)CQ_LDS_ADJ_WORDS);
    if (node_limit <= 0) node_limit = 300000;
    hipLaunchKernelGGL(max_clique_kernel, dim3(B), dim3(64), lds, st, adj, count, K, kstride, nws,
                       (long long)node_limit, stack, mask, n_in, flags);
    return hipGetLastError();
}
