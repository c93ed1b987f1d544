// Outlier rejection: pairwise-distance consistency graph + maximum clique (a8).
//
// Replaces outlierRejection.rejectOutliers (reference outlierRejection.py:16-95):
//   A[i][j] = | ||p_i-p_j|| - ||n_i-n_j|| | <= thr  in float64 (scipy cdist), inliers = a
//   maximum clique.  Result contract: THE REFERENCE'S clique, ties included - the first strictly-largest clique in
//   networkx.find_cliques order (outlierRejection.py:63-75).  That order is a deterministic property of networkx's
//   iterative Bron-Kerbosch and of CPython's set (integer nodes hash to themselves); oracle/c/clique.c states both,
//   nx_walk below re-creates them on the device.
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

// max_clique_kernel's helpers run on ONE wavefront: what they need between an LDS write and another lane's read is the order of the
// wave's own LDS operations (s_waitcnt + no compiler motion), not a workgroup barrier - the kernel's second wavefront (the cand chain
// of the walk, below) meets the first at its own two barriers only
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
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
// wave-wide maximum / minimum of an int through the DPP network (row shifts, then the two row broadcasts of gfx9): six VALU
// instructions instead of six ds_bpermute round trips; the result is wave-uniform (read back from lane 63)
__device__ __forceinline__ int wave_max_i(int v)
{
#define CQ_DPP_MAX(ctrl, rmask) v = max(v, __builtin_amdgcn_update_dpp(v, v, ctrl, rmask, 0xf, false))
    CQ_DPP_MAX(0x111, 0xf);     // row_shr:1
    CQ_DPP_MAX(0x112, 0xf);     // row_shr:2
    CQ_DPP_MAX(0x114, 0xf);     // row_shr:4
    CQ_DPP_MAX(0x118, 0xf);     // row_shr:8   -> lane 15 of every row holds the row's maximum
    CQ_DPP_MAX(0x142, 0xa);     // row_bcast:15 -> rows 1 and 3
    CQ_DPP_MAX(0x143, 0xc);     // row_bcast:31 -> rows 2 and 3
#undef CQ_DPP_MAX
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i(int v) { return -wave_max_i(-v); }
// A bitset lives in the first nw <= 16 lanes (one 64-bit word each), so its reductions are SCALAR work: the words that are not zero
// (one ballot) are fetched with v_readlane and counted / searched on the scalar unit.  (The butterfly of six ds_bpermute exchanges
// these replaced took ~700 cycles per call; the walk and the solver call them a dozen times per node.)
__device__ __forceinline__ uint64_t bs_word(uint64_t x, int w)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(x & 0xffffffffull), w);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(x >> 32), w);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int bs_count(uint64_t x)
{
    uint64_t bal = __ballot(x != 0);
    int s = 0;
    while (bal) {
        const int w = __ffsll((long long)bal) - 1;
        bal &= bal - 1;
        s += __popcll(bs_word(x, w));
    }
    return s;
}
__device__ __forceinline__ int bs_first(uint64_t x)
{
    const uint64_t bal = __ballot(x != 0);
    if (!bal) return -1;
    const int fl = __ffsll((long long)bal) - 1;
    return fl * 64 + (__ffsll((long long)bs_word(x, fl)) - 1);
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
#ifdef NX_EXP_STATS
__device__ unsigned long long nx_prof[24];
__shared__ unsigned long long nx_acc[24];           // accumulated in LDS (an add without return does not stall the wave), flushed at the end
#define NX_ADD(k, v) __hip_atomic_fetch_add(&nx_acc[k], (unsigned long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define NX_T0 const unsigned long long t0_ = __builtin_readcyclecounter();
#define NX_T1(k) { if (c.lane == 0) NX_ADD(k, __builtin_readcyclecounter() - t0_); }
#define NX_CNT(k) { if (c.lane == 0) NX_ADD(k, 1); }
#define NX_L0 unsigned long long tl_ = __builtin_readcyclecounter();
#define NX_L1(k) { const unsigned long long tn_ = __builtin_readcyclecounter(); if (lane == 0) NX_ADD(k, tn_ - tl_); tl_ = tn_; }
extern "C" int roam_debug_clique_prof(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nx_prof), sizeof(nx_prof)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[24] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(nx_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define NX_T0
#define NX_T1(k)
#define NX_CNT(k)
#define NX_L0
#define NX_L1(k)
#endif
// what the wavefront of the search hands to its helper and gets back.  cmd 1, a descent of the walk: cand, adj[q] -> cand & adj[q]
// (nx_walk); cmd 2, a degree pass of the solver: thr in deg, |P| in used, P in the first wavefront's sw -> the RM / UN / PE words of the
// upper vertex groups in live / occ / row, their branching key in mask (cq_solve); cmd 0: leave
struct NxMail { uint64_t live[64], occ[64], row[64]; int cmd, used, mask, ident, perfect, srctab, dsttab, deg; };
struct CqCtx {
    const uint64_t *A;          // adjacency rows (LDS or global), stride `as` words
    int as, nw, nws, lane;
    int wv;                     // 0: the wavefront that runs the search, 1: its helper (the cand chain of the walk)
    uint64_t *sw;               // 16 words of LDS: the current set broadcast to all lanes
    uint64_t *sw0;              // ... the first wavefront's
    NxMail *mail;
    uint64_t *stk;              // global scratch: 2 * nws words per level
    short *lsize, *lv, *lstage; // per level: |R|, branching vertex, stage
    long long nodes, node_limit;
    bool complete;
#ifdef NX_EXP_STATS
    int nq; long long nodes1;
#endif
};

// the degree pass of cq_solve over the vertex groups [g0, g1): per group of 64 vertices the words RM (fewer than thr neighbours inside
// P), UN (no conflict left), PE (one conflict) to lane g, and the lane's best branching key
__device__ __forceinline__ void cq_degrees(const CqCtx &c, const uint64_t *sw, int g0, int g1, int thr, int cnt,
                                           uint64_t &RM, uint64_t &UN, uint64_t &PE, int &key)
{
    const int lane = c.lane, nw = c.nw;
    for (int g = g0; g < g1; g++) {
        const int u = g * 64 + lane;
        const bool in = (sw[g] >> lane) & 1ull;
        int d = 0;
        if (in)
            for (int w = 0; w < nw; w++) d += __popcll(c.A[(int64_t)u * c.as + w] & sw[w]);
        const uint64_t brm = __ballot(in && d < thr);
        const uint64_t bun = __ballot(in && d == cnt - 1);
        const uint64_t bpe = __ballot(in && d == cnt - 2);
        if (lane == g) { RM = brm; UN = bun; PE = bpe; }
        if (in) key = min(key, d * 2048 + (2047 - u));      // fewest neighbours = most conflicts; ties: largest index
    }
}

template <bool TWO>
__device__ __forceinline__ int cq_solve(CqCtx &c, uint64_t P0, int best, int target, uint64_t &REC)
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
                WSYNC();
                const int thr = best - size;                            // an improving clique needs >= thr neighbours inside P
                uint64_t RM = 0, UN = 0, PE = 0;
                int key = 0x7fffffff;
#ifdef CQ_EXP_NOSPLIT
                if (false) {
#else
                if (TWO && nw >= 2) {
#endif
                    // the upper half of the vertex groups on the helper wavefront (round 5: the pass is the solver's unit of time -
                    // a dozen of them per search node - and the helper was idle outside the walk's descents)
                    NxMail *mb = c.mail;
                    const int gs = (nw + 1) >> 1;
                    if (lane == 0) { mb->cmd = 2; mb->deg = thr; mb->used = cnt; }
                    __syncthreads();
                    cq_degrees(c, c.sw, 0, gs, thr, cnt, RM, UN, PE, key);
                    __syncthreads();
                    if (lane >= gs && lane < nw) { RM = mb->live[lane]; UN = mb->occ[lane]; PE = mb->row[lane]; }
                    key = min(key, mb->mask);
                } else
                    cq_degrees(c, c.sw, 0, nw, thr, cnt, RM, UN, PE, key);
                WSYNC();
                if (__ballot(RM != 0)) { P &= ~RM; continue; }
                if (__ballot(UN != 0)) { R |= UN; size += bs_count(UN); P &= ~UN; continue; }
                if (__ballot(PE != 0)) {
                    // ALL the pendants of this pass, one after the other WITHOUT a new degree pass in between (round 4: one pendant per
                    // pass made the degree pass - K x nw adjacency words - the cost of every sparse conflict graph): removing vertices
                    // only lowers conflict degrees, so a pendant stays pendant (or becomes universal) whatever was taken before it; one
                    // that left P meanwhile was another pendant's partner
                    uint64_t PEc = PE;
                    for (;;) {
                        const int pv = bs_first(PEc);
                        if (pv < 0) break;
                        const uint64_t bv = bit_if(lane, pv);
                        PEc &= ~bv;
                        if (!__ballot((P & bv) != 0)) continue;
                        const uint64_t row = (lane < nw) ? c.A[(int64_t)pv * c.as + lane] : 0ull;
                        const int pu = bs_first(P & ~row & ~bv);       // its only conflict, if it is still there
                        R |= bv; size++;
                        P &= ~(bv | bit_if(lane, pu));
                    }
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
            // matching bound on the conflict graph of P - from node CQ_MATCH_AFTER on.  The greedy matching is a sequential pass over the
            // vertices of P (~0.25 us each on a lone wave); on scan-pair graphs - real and synthetic, 53 to 256 correspondences - it never
            // pruned a node (same node counts with and without) and was 80 % of the solver's time (round 5: profiles/clique_lone.py,
            // 4.8 -> 1.1 ms over sixteen real / bench-like pairs for phase 1; u256 11.9 -> 4.5 ms per 4096 problems).  It is kept for
            // searches that grow past a few hundred nodes, where it did earn its cost in round 2
#ifndef CQ_MATCH_AFTER
#define CQ_MATCH_AFTER 256
#endif
#ifndef CQ_EXP_NOMATCH
            if (c.nodes > CQ_MATCH_AFTER) {
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
#endif
            depth++;
            if (lane < nw) {
                c.stk[((int64_t)depth * 2) * c.nws + lane] = P;
                c.stk[((int64_t)depth * 2 + 1) * c.nws + lane] = R;
            }
            if (lane == 0) { c.lsize[depth] = (short)size; c.lv[depth] = (short)vbr; c.lstage[depth] = 0; }
            WSYNC();
            continue;
        }
        if (depth < 0) return best;
        const int stage = c.lstage[depth], v = c.lv[depth], sl = c.lsize[depth];
        WSYNC();
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


// ---------------------------------------------------------------------------------------------- the reference's tie-break
// rejectOutliers keeps the first strictly-largest clique that networkx.find_cliques yields.  find_cliques is an iterative
// Bron-Kerbosch with Tomita pivoting on Python sets of ints, so its order is fixed by CPython's set: open addressing (slot
// key & mask, 9 linear probes, then i*5+1+perturb), tables of 8 / 32 / 128 / 512 / 2048 slots as a set grows by insertion,
// the smallest power of two above 2*len for a copy, iteration and pop() in slot order, "&" iterating the smaller operand.
// nx_walk follows that tree with the SAME sets, but enters a child only when a clique of the size still needed exists among
// its candidates (witness clique or cq_solve).  A child's sets are new objects in the reference, so skipping a subtree leaves
// the parent's sets - and the order of everything after it - untouched; the first leaf of size omega is the reference's clique
// and the walk never backtracks.
//
// A set = membership bits by KEY (word per lane) + the table layout.  Whenever every key is smaller than the table (always
// true for the big sets: 77+ members live in 512 slots) each key sits in its own slot whatever the insertion history, the
// iteration order is ascending and no table is kept ("ident").  Otherwise the insertion sequence is replayed: the probe
// sequence only needs the OCCUPANCY of the table, held as one 64-slot word per lane and read with v_readlane, so an
// insertion is a few scalar bit operations; the keys go to a small LDS table for later iteration.
struct NxSet {
    uint64_t live;      // members, bit per key
    uint64_t occ;       // explicit layout: occupied slots (live or dummy), bit per slot
    uint16_t *tab;      // explicit layout: slot -> key
    int mask, used;
    bool ident;
    bool perfect;       // explicit layout in which every key sits at its home slot key & mask (no key was ever displaced)
};
struct NxLds { uint16_t *tab[6], *seq, *seq2, *seq3, *slot; uint32_t *T; int ts; NxMail *mail; };

__host__ __device__ inline int nx_table_slots(int K)
{
    int ts = 8;                                     // an explicit table exists only when some key >= its size: size < K
    while (ts * 2 < K && ts < 512) ts *= 2;
    return ts;
}
__host__ __device__ inline size_t nx_scratch_bytes(int K)       // seq, seq2, seq3, slot + T of one wavefront
{
    return ((4 * (size_t)(K + 2)) * sizeof(uint16_t) + (size_t)nx_table_slots(K) * sizeof(uint32_t) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t nx_lds_bytes(int K)
{
    return (((size_t)7 * nx_table_slots(K) * sizeof(uint16_t) + 15) & ~(size_t)15) + 2 * nx_scratch_bytes(K) + sizeof(NxMail);
}
// the walk's LDS: seven tables (0/1 the subg chain, 2/3 the cand chain, 4 adj[q] of wavefront 0, 5 ext_u, 6 adj[q] of wavefront 1), the
// lists of each wavefront, the mailbox
__device__ inline void nx_lds_carve(NxLds &L, unsigned char *mem, int K, int wv)
{
    L.ts = nx_table_slots(K);
    uint16_t *base = reinterpret_cast<uint16_t *>(mem);
    for (int t = 0; t < 6; t++) L.tab[t] = base + t * L.ts;
    if (wv) L.tab[4] = L.tab[5] = base + 6 * L.ts;
    unsigned char *scr = mem + (((size_t)7 * L.ts * sizeof(uint16_t) + 15) & ~(size_t)15) + (size_t)wv * nx_scratch_bytes(K);
    L.seq = reinterpret_cast<uint16_t *>(scr);
    L.seq2 = L.seq + (K + 2);
    L.seq3 = L.seq2 + (K + 2);
    L.slot = L.seq3 + (K + 2);
    L.T = reinterpret_cast<uint32_t *>(L.slot + (K + 2));
    L.mail = reinterpret_cast<NxMail *>(mem + (((size_t)7 * L.ts * sizeof(uint16_t) + 15) & ~(size_t)15) + 2 * nx_scratch_bytes(K));
}

__device__ __forceinline__ uint64_t rl64(uint64_t v, int src)
{
    src = __builtin_amdgcn_readfirstlane(src);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int bs_last(uint64_t x)
{
    const uint64_t bal = __ballot(x != 0);
    if (!bal) return -1;
    const int hl = 63 - __clzll((long long)bal);
    const uint64_t wv = bs_word(x, hl);
    return hl * 64 + 63 - __clzll((long long)wv);
}
__device__ __forceinline__ uint64_t bits_from(int lane, int first)      // keys / slots >= first
{
    const int w = first >> 6;
    return lane > w ? ~0ull : (lane == w ? (~0ull << (first & 63)) : 0ull);
}

// members of S that are in F, in S's iteration order -> seq[0..n); maxkey = the largest of them
__device__ int nx_seq(const CqCtx &c, const NxSet &S, uint64_t F, uint16_t *seq, int &maxkey)
{
    const int lane = c.lane;
    const uint64_t X = S.live & F;
    maxkey = bs_last(X);
    int n = 0;
    const uint64_t below = (1ull << lane) - 1ull;
    if (S.ident) {
        for (int w = 0; w < c.nw; w++) {
            const uint64_t bits = rl64(X, w);
            if (!bits) continue;
            if ((bits >> lane) & 1ull) seq[n + __popcll(bits & below)] = (uint16_t)(w * 64 + lane);
            n += __popcll(bits);
        }
    } else {
        WSYNC();
        if (lane < 16) c.sw[lane] = X;
        WSYNC();
        for (int ch = 0; ch * 64 <= S.mask; ch++) {
            const uint64_t ob = rl64(S.occ, ch);
            if (!ob) continue;
            bool in = (ob >> lane) & 1ull;
            int key = 0;
            if (in) { key = S.tab[ch * 64 + lane]; in = (c.sw[key >> 6] >> (key & 63)) & 1ull; }
            const uint64_t bal = __ballot(in);
            if (in) seq[n + __popcll(bal & below)] = (uint16_t)key;
            n += __popcll(bal);
        }
    }
    WSYNC();
    return n;
}

// slots the keys P[0..m) take when they are inserted IN THIS ORDER into an empty table of mask + 1 slots (set_insert_clean /
// set_add_entry without dummies; the keys are distinct).  Sequentially that is a chain of m dependent probes; here every key
// (a lane each) walks its probe sequence against T[s] = the EARLIEST key currently assigned to slot s and takes the first slot no
// earlier key holds; T is rebuilt and the walk repeated until nothing moves.  A key's choice depends on earlier keys only, so after
// k rounds the first k keys sit where the sequential insertion puts them, and the fixed point is the sequential result; the rounds
// needed are the longest chain of displacements (a handful at the load factors CPython allows), not m.  Returns the occupancy.
__device__ uint64_t nx_phase(const CqCtx &c, const uint16_t *P, int m, int mask, uint16_t *slot, uint32_t *T)
{
    const int lane = c.lane;
    for (int i = lane; i < m; i += 64) slot[i] = (uint16_t)(P[i] & mask);
    for (;;) {
        WSYNC();
        for (int s0 = lane; s0 <= mask; s0 += 64) T[s0] = 0xffffffffu;
        WSYNC();
        // (atomicMin through a generic pointer to LDS raises a memory-aperture fault here: the minimum by repeated plain stores)
        for (int i0 = 0; i0 < m; i0 += 64) {
            const int i = i0 + lane;
            for (;;) {
                const bool want = i < m && T[slot[i]] > (uint32_t)i;
                if (!__ballot(want)) break;
                if (want) T[slot[i]] = (uint32_t)i;
                WSYNC();
            }
        }
        WSYNC();
        bool moved = false;
        for (int i = lane; i < m; i += 64) {
            const unsigned key = P[i];
            unsigned perturb = key, i0 = key & (unsigned)mask, got;
            for (;;) {
                const int lim = (i0 + 9 <= (unsigned)mask) ? 9 : 0;
                int j = 0;
                for (; j <= lim; j++) if (T[i0 + j] >= (uint32_t)i) break;
                if (j <= lim) { got = i0 + j; break; }
                perturb >>= 5;
                i0 = (i0 * 5 + 1 + perturb) & (unsigned)mask;
            }
            if (got != slot[i]) { moved = true; slot[i] = (uint16_t)got; }
        }
        if (!__ballot(moved)) break;
    }
    uint64_t occ = 0;
    for (int ch = 0; ch * 64 <= mask; ch++) {
        const uint64_t b = __ballot(ch * 64 + lane <= mask && T[ch * 64 + lane] != 0xffffffffu);
        if (lane == ch) occ = b;
    }
    return occ;
}

// P2[0..m) = the keys P[0..m) in the order of their slots (what set_table_resize re-inserts, what iteration yields)
__device__ void nx_slot_order(const CqCtx &c, const uint16_t *P, const uint16_t *slot, int m, uint64_t occ, uint16_t *P2)
{
    const int lane = c.lane;
    WSYNC();
    if (lane < 16) c.sw[lane] = occ;                                   // tables of at most 1024 slots reach this point
    WSYNC();
    for (int i = lane; i < m; i += 64) {
        const int sl = slot[i], w = sl >> 6;
        int r = __popcll(c.sw[w] & ((1ull << (sl & 63)) - 1ull));
        for (int k = 0; k < w; k++) r += __popcll(c.sw[k]);
        P2[r] = P[i];
    }
    WSYNC();
}

// slot of `key` in a table of at most 128 slots whose occupancy is the pair (o0, o1): everything wave-uniform, scalar ALU only
__device__ __forceinline__ int nx_probe2(uint64_t o0, uint64_t o1, int mask, int key)
{
    unsigned perturb = (unsigned)key, i = (unsigned)key & (unsigned)mask;
    for (;;) {
        const int lim = (i + 9 <= (unsigned)mask) ? 9 : 0;
        const int sh = (int)(i & 63);
        uint64_t win = ((i >> 6) ? o1 : o0) >> sh;
        if (sh + lim >= 64) win |= o1 << (64 - sh);                   // (i < 64 here: i + 9 <= mask <= 127)
        const uint64_t fr = ~win & ((2ull << lim) - 1ull);
        if (fr) return (int)i + __ffsll((long long)fr) - 1;
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (unsigned)mask;
    }
}

// the scalar replay of a SMALL table (5 keys in 8 slots, 19 in 32): the keys sit in ONE register (lane j = the j-th key inserted), an
// insertion is v_readlane + a dozen scalar instructions + a lane select for the slot, and "re-insert in slot order" at a resize is one
// ds_permute by the rank of the slot.
// one pass: keys of lanes 0..m-1 into an empty table; the lane's slot lands in sv, the occupancy in (o0, o1)
__device__ __forceinline__ void nx_run_small(int lane, int kv, int &sv, uint64_t &o0, uint64_t &o1, int m, int mask)
{
    o0 = 0; o1 = 0;
    for (int j = 0; j < m; j++) {
        const int key = __builtin_amdgcn_readlane(kv, j);
        int sl = key & mask;
        if ((((sl >> 6) ? o1 : o0) >> (sl & 63)) & 1ull) sl = nx_probe2(o0, o1, mask, key);      // (home slot taken: the probe sequence)
        if (sl < 64) o0 |= 1ull << sl; else o1 |= 1ull << (sl - 64);
        sv = lane == j ? sl : sv;
    }
}
__device__ __forceinline__ int nx_reorder_small(int lane, int kv, int sv, uint64_t o0, int m)
{
    const int r = __popcll(o0 & ((1ull << (sv & 63)) - 1ull));       // (a table of <= 32 slots) rank of the lane's slot
    return __builtin_amdgcn_ds_permute((lane < m ? r : lane) << 2, kv);
}
// nx_build for sets of at most 128 members in tables of at most 256 slots (every explicit table there is while K <= 512).  Kept out of
// line: inlined into the walk (three copies) the kernel needed 199 registers and the build of ROCm 7.2 produced a binary that faulted.
//   * no two keys with the same home slot in the FINAL table => every key finds its home free whenever it is inserted: the layout is
//     "key at key & mask" whatever the order, and the earlier tables (which only decide that order) need no replay either.  The rule
//     for most sets of 19..76 members here (128 slots, ids below ~170: only k and k + 128 can meet).  Returns 1.
//   * otherwise the small tables the set grew through (5 keys in 8 slots, 19 in 32) are replayed key by key on the scalar unit
//     (nx_run_small; the keys sit in ONE register, lane j = the j-th key inserted) - or not at all: along the forced levels of the
//     walk a child's first 19 keys are mostly its parent's, so the outcome of the two small tables is remembered per set (ci = 0 the
//     subg chain, 1 the cand chain, -1 neither) under the 19 keys that produced it;
//   * and the FINAL table is filled by all keys at once (round 5; it was one key at a time, ~100 cycles each on a lone wave, and the
//     LDS fixed point of nx_phase for more than 64 keys: 25 us a set): T[s] = the earliest key that wants slot s (ds_min), every key
//     walks its probe sequence to the first slot no EARLIER key holds, until nothing moves - the fixed point is the sequential
//     result (see nx_phase), the rounds needed are the longest chain of displacements: two or three at these load factors.
__shared__ uint32_t nx_Tw[2][128];                  // one per wavefront
__shared__ uint16_t nx_ck[2][32], nx_cv[2][32];      // [ci][0..18] the keys / the outcome, [ci][19] of nx_ck: 1 = valid
// INVARIANT: the memo is indexed by CHAIN (ci), not by wavefront.  It is race-free because each chain belongs to one wavefront at a
// time: chain 0 (the subg chain) only ever runs on the first wavefront, chain 1 (the cand chain) on the helper between the two mailbox
// barriers of a level - and the first wavefront never builds a chain-1 table (tab[2] / tab[3]) between those barriers.  A change that
// lets both wavefronts build tables of one chain at once needs a memo per wavefront.
__device__ __forceinline__ int nx_probe_T(const uint32_t *nx_T, int key, int idx, int mask)
{
    unsigned perturb = (unsigned)key, i0 = (unsigned)key & (unsigned)mask;
    for (;;) {
        const int lim = (i0 + 9 <= (unsigned)mask) ? 9 : 0;
        for (int j = 0; j <= lim; j++) if (nx_T[i0 + j] >= (uint32_t)idx) return (int)i0 + j;
        perturb >>= 5;
        i0 = (i0 * 5 + 1 + perturb) & (unsigned)mask;
    }
}
__device__ __attribute__((noinline)) int nx_build_fast(int lane, uint64_t *occ_out, const uint16_t *seq, int n_, int size_, bool copy_, uint16_t *tab, int ci_, int wv_)
{
    uint32_t *nx_T = nx_Tw[__builtin_amdgcn_readfirstlane(wv_)];
    // (arguments of an out-of-line function arrive in vector registers: say that these are wave-uniform, or every loop below
    // becomes a divergent one and v_readlane a waterfall)
    const int n = __builtin_amdgcn_readfirstlane(n_), size = __builtin_amdgcn_readfirstlane(size_), ci = __builtin_amdgcn_readfirstlane(ci_);
    const bool copy = __builtin_amdgcn_readfirstlane((int)copy_) != 0;
    NX_L0
    int k0 = lane < n ? (int)seq[lane] : 0;
    const int k1 = lane + 64 < n ? (int)seq[lane + 64] : 0;
    const int kin = k0;
    const bool grows = !copy && n >= 5 && size > 8, big = grows && n >= 19 && size > 32;
    int s0 = 0, s1 = 0;
    bool first = true, perfect = false;
    uint64_t l0 = 0, l1 = 0;                                            // slots taken by displaced keys
    uint32_t t0 = 0xffffffffu, t1 = 0xffffffffu;
    for (;;) {
        // The final table (first of all: with no two keys at the same home there is nothing else to do): keys 0..n-1 (lane j and, from
        // 64 on, lane j - 64 of the second register) into size empty slots, in that order.
        // T[s] = the earliest key whose HOME is s.  A key that finds its home taken - by an earlier key with the same home, or by a
        // displaced key that landed there first - is "displaced"; everybody else sits at home.  The displaced keys are few and are
        // placed one after the other in insertion order with wave-uniform arithmetic: slot s is taken when key d arrives iff an
        // earlier key has its home there (T[s] < d: that key sits there, or whoever displaced it came even earlier - one vector
        // compare gives all those slots as a bit mask) or an earlier displaced key landed there; a LATER home key found in the slot
        // taken is displaced in turn.
        const int m = n, msk = size - 1;
        const bool a0 = lane < m, a1 = lane + 64 < m;
        s0 = k0 & msk; s1 = k1 & msk;
        l0 = 0; l1 = 0;
        WSYNC();
        nx_T[lane] = 0xffffffffu;
        if (msk > 63) nx_T[lane + 64] = 0xffffffffu;
        WSYNC();
        if (a0) atomicMin(&nx_T[s0], (uint32_t)lane);
        if (a1) atomicMin(&nx_T[s1], (uint32_t)(lane + 64));
        WSYNC();
        t0 = nx_T[lane];
        t1 = msk > 63 ? nx_T[lane + 64] : 0xffffffffu;
        uint64_t p0 = __ballot(a0 && nx_T[s0] != (uint32_t)lane), p1 = __ballot(a1 && nx_T[s1] != (uint32_t)(lane + 64));
        if (first) {
            first = false;
            NX_L1(16)
            if (!(p0 | p1)) { perfect = true; break; }
            if (grows) {
                bool hit = false;
                if (big && ci >= 0) hit = !__ballot(lane < 19 && nx_ck[ci][lane] != (uint16_t)k0) && nx_ck[ci][19] == 1;
#ifdef NX_EXP_STATS
                if (lane == 0 && big) NX_ADD(hit ? 15 : 10, 1);
#endif
                if (hit) { if (lane < 19) k0 = nx_cv[ci][lane]; }
                else {
                    // (the small tables key by key on the scalar unit: half of their keys are displaced, which the scheme of the
                    // final table does not like - measured: 6.1 k cycles per set against 4.3 k this way)
                    int sv = 0;
                    uint64_t q0, q1;
                    nx_run_small(lane, k0, sv, q0, q1, 5, 7); k0 = nx_reorder_small(lane, k0, sv, q0, 5);
                    if (big) {
                        nx_run_small(lane, k0, sv, q0, q1, 19, 31); k0 = nx_reorder_small(lane, k0, sv, q0, 19);
                        if (ci >= 0) { if (lane < 19) { nx_ck[ci][lane] = (uint16_t)kin; nx_cv[ci][lane] = (uint16_t)k0; } if (lane == 19) nx_ck[ci][19] = 1; }
                    }
                }
                NX_L1(17)
                continue;
            }
        }
        while (p0 | p1) {
#ifdef NX_EXP_STATS
            if (lane == 0) NX_ADD(9, 1);
#endif
            int d;
            if (p0) { d = __ffsll((long long)p0) - 1; p0 &= p0 - 1; } else { d = 64 + __ffsll((long long)p1) - 1; p1 &= p1 - 1; }
            const int key = d < 64 ? __builtin_amdgcn_readlane(k0, d) : __builtin_amdgcn_readlane(k1, d - 64);
            const uint64_t o0 = __ballot(t0 < (uint32_t)d) | l0, o1 = __ballot(t1 < (uint32_t)d) | l1;      // (slots past the table: INF, never "taken" - and never probed)
            unsigned perturb = (unsigned)key, i = (unsigned)key & (unsigned)msk;
            int got;
            for (;;) {
                const int lim = (i + 9 <= (unsigned)msk) ? 9 : 0;
                const int sh = (int)(i & 63);
                uint64_t win = ((i >> 6) ? o1 : o0) >> sh;
                if (!(i >> 6) && sh + lim >= 64) win |= o1 << (64 - sh);
                const uint64_t fr = ~win & ((2ull << lim) - 1ull);
                if (fr) { got = (int)i + __ffsll((long long)fr) - 1; break; }
                perturb >>= 5;
                i = (i * 5 + 1 + perturb) & (unsigned)msk;
            }
            const uint32_t tv = got < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)t0, got) : (uint32_t)__builtin_amdgcn_readlane((int)t1, got - 64);
            if (tv != 0xffffffffu) { if (tv < 64u) p0 |= 1ull << tv; else p1 |= 1ull << (tv - 64u); }           // its home key comes later: displaced
            if (got < 64) l0 |= 1ull << got; else l1 |= 1ull << (got - 64);
            if (d < 64) s0 = lane == d ? got : s0; else s1 = lane == d - 64 ? got : s1;
        }
        NX_L1(18)
        break;
    }
    if (lane < n) tab[s0] = (uint16_t)k0;
    if (lane + 64 < n) tab[s1] = (uint16_t)k1;
    const uint64_t b0 = __ballot(t0 != 0xffffffffu && lane < size) | l0, b1 = __ballot(t1 != 0xffffffffu) | l1;
    *occ_out = lane == 0 ? b0 : lane == 1 ? b1 : 0ull;
    NX_L1(19)
    return perfect ? 1 : 0;
}

// table of a set built by inserting seq[0..n) one by one (copy = false: growth 8 -> 32 -> 128 -> 512 -> 2048, set_add_entry
// resizing to used*4 once fill*5 >= mask*3 and re-inserting in slot order) or in one go into the copy's table (copy = true)
__device__ void nx_build(const CqCtx &c, NxLds &L, NxSet &D, const uint16_t *seq, int n, int maxkey, bool copy, uint16_t *tab)
{
    const int lane = c.lane;
    int size = 8;
    if (copy) { if (n >= 5) while (size <= 2 * n) size <<= 1; }
    else size = n < 5 ? 8 : n < 19 ? 32 : n < 77 ? 128 : n < 307 ? 512 : 2048;
    D.used = n; D.tab = tab; D.occ = 0; D.mask = size - 1;
    D.perfect = false;
    D.ident = maxkey < size;
#ifdef NX_EXP_NOBUILD
    D.ident = true;
#endif
    if (D.ident) return;
    NX_CNT(8)
    if (size <= 128 && n <= 128) {
        uint64_t o;
        int pf;
        const int ci = (tab == L.tab[0] || tab == L.tab[1]) ? 0 : (tab == L.tab[2] || tab == L.tab[3]) ? 1 : -1;
        { NX_T0 pf = nx_build_fast(c.lane, &o, seq, n, size, copy, tab, ci, c.wv); NX_T1(13) }
        WSYNC();
        D.occ = o;
        D.perfect = __builtin_amdgcn_readfirstlane(pf) != 0;
#ifdef NX_EXP_STATS
        if (D.perfect) NX_CNT(12)
#endif
        return;
    }
    const uint16_t *P = seq;
    uint16_t *bufa = L.seq2, *bufb = L.seq3;
    if (!copy) {
        // the tables the set went through before its last resize: 5 keys in 8 slots, 19 in 32, 77 in 128
        const int cut[3] = {5, 19, 77}, msk[3] = {7, 31, 127};
        int have = 0;                                                   // P[0..have) = the re-inserted keys, then seq[have..)
        for (int ph = 0; ph < 3 && n >= cut[ph] && msk[ph] < size - 1; ph++) {
            const int m = cut[ph];
            // list of this phase: the previous phase's keys in slot order + the keys inserted since
            uint16_t *cur = (P == bufa) ? bufb : bufa;
            for (int i = lane; i < m; i += 64) cur[i] = (i < have) ? P[i] : seq[i];
            WSYNC();
            const uint64_t occ = nx_phase(c, cur, m, msk[ph], L.slot, L.T);
            uint16_t *nxt = (cur == bufa) ? bufb : bufa;
            nx_slot_order(c, cur, L.slot, m, occ, nxt);
            P = nxt; have = m;
        }
        if (have) {
            uint16_t *cur = (P == bufa) ? bufb : bufa;
            for (int i = lane; i < n; i += 64) cur[i] = (i < have) ? P[i] : seq[i];
            WSYNC();
            P = cur;
        }
    }
    const uint64_t occ = nx_phase(c, P, n, size - 1, L.slot, L.T);
    for (int i = lane; i < n; i += 64) tab[L.slot[i]] = P[i];
    WSYNC();
    D.occ = occ;
}

__device__ __forceinline__ int nx_incr_size(int n) { return n < 5 ? 8 : n < 19 ? 32 : n < 77 ? 128 : n < 307 ? 512 : 2048; }

// D = {x in ITER's order if x in F}: set_intersection / the iterating branch of set_difference
__device__ void nx_filter_build(const CqCtx &c, NxLds &L, NxSet &D, const NxSet &ITER, uint64_t F, uint16_t *tab)
{
    const uint64_t X = ITER.live & F;
    D.live = X;
    // the common case needs no replay at all: every key below the table size => each key in its own slot, whatever the order
    const int n0 = bs_count(X), mk0 = bs_last(X);
    if (mk0 < nx_incr_size(n0)) { D.used = n0; D.tab = tab; D.occ = 0; D.mask = nx_incr_size(n0) - 1; D.ident = true; D.perfect = false; return; }
    int maxkey, n;
    { NX_T0 n = nx_seq(c, ITER, F, L.seq, maxkey); NX_T1(14) }
    nx_build(c, L, D, L.seq, n, maxkey, false, tab);
}

// adj[q] = {v for v in G[q] if v != q}: ascending insertion of the row
__device__ void nx_adj_set(const CqCtx &c, NxLds &L, NxSet &D, uint64_t row, uint16_t *tab)
{
    NxSet asc; asc.live = row; asc.ident = true; asc.perfect = false; asc.mask = 0; asc.used = 0; asc.occ = 0; asc.tab = nullptr;
    nx_filter_build(c, L, D, asc, ~0ull, tab);
}

// S & adj[q] (either operand order: the smaller one is iterated, adj[q] on equal sizes)
__device__ void nx_and_adj(const CqCtx &c, NxLds &L, NxSet &D, const NxSet &S, int slen, uint64_t row, int deg, uint16_t *tab)
{
    if (deg > slen) { nx_filter_build(c, L, D, S, row, tab); return; }
    // adj[q] is the iterated operand.  It was built by ascending insertion: when all its keys are below its table size it iterates
    // in ascending order, and so does the result's insertion sequence - no table of adj[q] is needed (nearly always: 77+ neighbours
    // sit in 512 slots)
    if (bs_last(row) < nx_incr_size(deg)) {
        NxSet asc; asc.live = row; asc.ident = true; asc.perfect = false; asc.mask = 0; asc.used = deg; asc.occ = 0; asc.tab = nullptr;
        nx_filter_build(c, L, D, asc, S.live, tab);
        return;
    }
    NxSet A;
    nx_adj_set(c, L, A, row, L.tab[4]);
    nx_filter_build(c, L, D, A, S.live, tab);
}

// pivot = max(subg, key = |cand & adj[u]|), the FIRST maximum in subg's iteration order
__device__ int nx_pivot(const CqCtx &c, const NxSet &SG, uint64_t candbits)
{
    const int lane = c.lane;
    WSYNC();
    if (lane < 16) c.sw[lane] = candbits;
    WSYNC();
    int best = -1;                                                      // (count << 12) | (4095 - position), then the key
    int bestu = 0;
    const int chunks = SG.ident ? c.nw : (SG.mask >> 6) + 1;
    for (int ch = 0; ch < chunks; ch++) {
        const uint64_t bits = rl64(SG.ident ? SG.live : SG.occ, ch);
        if (!bits) continue;
        if (!((bits >> lane) & 1ull)) continue;
        const int pos = ch * 64 + lane;
        const int u = SG.ident ? pos : (int)SG.tab[pos];
        int d = 0;
        for (int w = 0; w < c.nw; w++) d += __popcll(c.A[(int64_t)u * c.as + w] & c.sw[w]);
        const int key = (d << 12) | (4095 - pos);
        if (key > best) { best = key; bestu = u; }
    }
    const int m = wave_max_i(best);
    const uint64_t who = __ballot(best == m);
    const int src = __ffsll((long long)who) - 1;
    WSYNC();
    return __builtin_amdgcn_readlane(bestu, src);
}

// ext_u = cand - adj[u]
__device__ void nx_sub_adj(const CqCtx &c, NxLds &L, NxSet &E, const NxSet &CD, uint64_t row, int deg)
{
    const int lane = c.lane;
    if ((CD.used >> 2) > deg) {                                         // set_copy_and_difference (cand is fresh here: no dummies)
        int size = 8;
        if (CD.used >= 5) while (size <= 2 * CD.used) size <<= 1;
        if (size - 1 == CD.mask) {                                      // same table size: slot-for-slot copy
            E = CD; E.tab = L.tab[5];
            if (!CD.ident) { for (int i = lane; i <= CD.mask; i += 64) L.tab[5][i] = CD.tab[i]; WSYNC(); }
        } else {
            int maxkey;
            const int n = nx_seq(c, CD, ~0ull, L.seq, maxkey);
            E.live = CD.live;
            nx_build(c, L, E, L.seq, n, maxkey, true, L.tab[5]);
        }
        E.live &= ~row;                                                 // discards leave dummies: the layout stays
        E.used = bs_count(E.live);
        return;
    }
    nx_filter_build(c, L, E, CD, ~row, L.tab[5]);
}

// q = ext_u.pop(): the first live slot (the finger only ever moves forward here)
__device__ int nx_pop(const CqCtx &c, NxSet &E)
{
    const int lane = c.lane;
    int q;
    if (E.ident) q = bs_first(E.live);
    else {
        q = -1;
        for (int ch = 0; ch * 64 <= E.mask && q < 0; ch++) {
            uint64_t ob = rl64(E.occ, ch);
            while (ob) {
                const int sl = __ffsll((long long)ob) - 1;
                ob &= ob - 1;
                const int key = E.tab[ch * 64 + sl];
                const bool alive = __ballot((E.live & bit_if(lane, key)) != 0) != 0;
                if (lane == ch) E.occ &= ~(1ull << sl);                  // popped or already a dummy: never looked at again
                if (alive) { q = key; break; }
            }
        }
    }
    if (q >= 0) { E.live &= ~bit_if(lane, q); E.used--; }
    return q;
}

// ext_u of at most four members (by far the commonest case: the pivot is adjacent to almost everything): the 8-slot table is replayed
// on the scalar unit and the members come back packed in pop order, 16 bits each.  cand iterates in ascending order (ident) or in the
// order of its explicit table (nx_seq lists the few members that are left).  false: not this case.
__device__ __forceinline__ bool nx_sub_adj_tiny(const CqCtx &c, NxLds &L, const NxSet &CD, uint64_t row, int deg, uint64_t &elist, int &en)
{
    if ((CD.used >> 2) > deg) return false;
    uint64_t X = CD.live & ~row;
    const int n = bs_count(X);
    if (n > 4) return false;
    if (!CD.ident) { int mk; nx_seq(c, CD, ~row, L.seq, mk); }
    unsigned occ = 0;
    unsigned e0 = ~0u, e1 = ~0u, e2 = ~0u, e3 = ~0u;                   // (slot << 16) | key; unused entries sort to the end
#define NX_TINY_INS(e, k)                                                                                               \
    if (k < n) {                                                                                                        \
        int key;                                                                                                        \
        if (CD.ident) { key = bs_first(X); X &= ~bit_if(c.lane, key); }                                                 \
        else key = __builtin_amdgcn_readfirstlane((int)L.seq[k]);                                                       \
        unsigned perturb = (unsigned)key, i = (unsigned)key & 7u;                                                       \
        while ((occ >> i) & 1u) { perturb >>= 5; i = (i * 5 + 1 + perturb) & 7u; }     /* mask 7: no linear probes */     \
        occ |= 1u << i;                                                                                                 \
        e = (i << 16) | (unsigned)key;                                                                                  \
    }
    NX_TINY_INS(e0, 0) NX_TINY_INS(e1, 1) NX_TINY_INS(e2, 2) NX_TINY_INS(e3, 3)
#undef NX_TINY_INS
#define NX_CSWAP(a, b) { const unsigned lo_ = min(a, b), hi_ = max(a, b); a = lo_; b = hi_; }
    NX_CSWAP(e0, e1) NX_CSWAP(e2, e3) NX_CSWAP(e0, e2) NX_CSWAP(e1, e3) NX_CSWAP(e1, e2)
#undef NX_CSWAP
    elist = (uint64_t)(e0 & 0xffffu) | ((uint64_t)(e1 & 0xffffu) << 16) | ((uint64_t)(e2 & 0xffffu) << 32) | ((uint64_t)(e3 & 0xffffu) << 48);
    en = n;
    return true;
}

// The long forced prefix of the walk, taken in one go.  In a scan pair's graph most of the clique is UNIVERSAL inside cand (adjacent to
// every other candidate): such a vertex has the maximum pivot key |cand| - 1, so networkx picks the first of them (in subg's
// iteration order) as pivot, ext_u = {pivot}, and descends - one level per universal vertex, omega minus a handful of levels in
// which nothing is decided.  While subg and cand keep every key in its own slot (ascending iteration) those levels are known
// in advance: T = the vertices of subg with key |cand| - 1 only ever shrinks (T' = T & adj[v] when v is taken; nobody joins), the
// pivots are its members in ascending order, and a level is nothing but "Q += v, cand -= v, subg &= adj[v]".  The bulk stops
// where the real walk has something to do: the first tied vertex lies outside cand (an excluded vertex is the pivot), a child set
// would need an explicit table, or one candidate is left.
// occupancy of a perfect table of mask + 1 >= 128 slots from its members: slot word h = OR of the key words w with w = h mod (words)
__device__ __forceinline__ uint64_t nx_fold_occ(uint64_t live, int mask, int lane, int nw)
{
    const int hw = (mask + 1) >> 6;
    uint64_t o = 0;
    for (int w = 0; w < nw; w++) {
        const uint64_t x = bs_word(live, w);
        if (lane == (w & (hw - 1))) o |= x;
    }
    return o;
}

__device__ void nx_bulk(const CqCtx &c, NxSet &subg, NxSet &cand, uint64_t &RF, int &size)
{
    // "every key in its own slot" comes in two forms: all keys below the table size (ident: ascending iteration) or an explicit table
    // of 128+ slots without a displaced key (perfect: iteration by key & mask; a subset of a perfect table re-inserted into a table
    // of the SAME size is perfect again, whatever the insertion order - so a forced level is still "clear the bit")
    const bool sp = !subg.ident && subg.perfect && subg.mask >= 127, cp = !cand.ident && cand.perfect && cand.mask >= 127;
    if (!(subg.ident || sp) || !(cand.ident || cp) || cand.used < 2) return;
    const int lane = c.lane, nw = c.nw;
    WSYNC();
    if (lane < 16) c.sw[lane] = cand.live;
    WSYNC();
    uint64_t T = 0;
    const int full = cand.used - 1;
    for (int ch = 0; ch < nw; ch++) {
        const uint64_t bits = bs_word(subg.live, ch);
        if (!bits) continue;
        const bool in = (bits >> lane) & 1ull;
        int d = -1;
        if (in) {
            const int u = ch * 64 + lane;
            d = 0;
            for (int w = 0; w < nw; w++) d += __popcll(c.A[(int64_t)u * c.as + w] & c.sw[w]);
        }
        const uint64_t b = __ballot(in && d == full);
        if (lane == ch) T = b;
        if (__ballot(in && d > full)) { T = 0; break; }                 // an excluded vertex adjacent to ALL of cand is the pivot: nothing to skip
    }
    WSYNC();
    const int hws = sp ? ((subg.mask + 1) >> 6) - 1 : 0xffff;           // iteration order of subg: by (word & hws) * 64 + bit
    bool folded = false;
    for (;;) {
        if (cand.used < 2) break;
        // the first tied vertex in subg's iteration order
        int key = 0x7fffffff;
        if (T) { const int bpos = __ffsll((long long)T) - 1; key = ((((lane & hws) << 6) | bpos) << 11) | (lane * 64 + bpos); }
        key = wave_min_i(key);
        if (key == 0x7fffffff) break;
        const int v = key & 2047;
        const uint64_t bv = bit_if(lane, v);
        if (!__ballot((cand.live & bv) != 0)) break;                    // an excluded vertex ties with the universal ones: it is the pivot
        const uint64_t row = (lane < nw) ? c.A[(int64_t)v * c.as + lane] : 0ull;
        const uint64_t nc = cand.live & ~bv, ns = subg.live & row;
        const int ncn = cand.used - 1, nsn = bs_count(ns);
        // the children keep "every key in its own slot": ident sets while their keys stay below the (shrinking) table size, perfect
        // sets while the table size does not change
        if (cp ? (nx_incr_size(ncn) != cand.mask + 1) : (bs_last(nc) >= nx_incr_size(ncn))) break;
        if (sp ? (nx_incr_size(nsn) != subg.mask + 1) : (bs_last(ns) >= nx_incr_size(nsn))) break;
        RF |= bv; size++;
        cand.live = nc; cand.used = ncn; if (!cp) cand.mask = nx_incr_size(ncn) - 1;
        subg.live = ns; subg.used = nsn; if (!sp) subg.mask = nx_incr_size(nsn) - 1;
        T &= row;
        folded = true;
    }
    if (folded) {                                                       // the explicit tables lose the slots of what left
        if (cp) cand.occ = nx_fold_occ(cand.live, cand.mask, lane, nw);
        if (sp) subg.occ = nx_fold_occ(subg.live, subg.mask, lane, nw);
    }
}

// the second wavefront of the workgroup: cand & adj[q] of every descent of the walk, until told to leave (cmd 0)
__device__ __forceinline__ void nx_helper(CqCtx &c, NxLds &L)
{
    const int lane = c.lane;
    NxMail *mb = L.mail;
    for (;;) {
        __syncthreads();
        const int cmd = mb->cmd;
        if (cmd == 0) return;
        if (cmd == 2) {
            const int nw = c.nw, gs = (nw + 1) >> 1, thr = mb->deg, cnt = mb->used;
            uint64_t RM = 0, UN = 0, PE = 0;
            int key = 0x7fffffff;
            cq_degrees(c, c.sw0, gs, nw, thr, cnt, RM, UN, PE, key);
            if (lane >= gs && lane < nw) { mb->live[lane] = RM; mb->occ[lane] = UN; mb->row[lane] = PE; }
            key = wave_min_i(key);
            if (lane == 0) mb->mask = key;
            __syncthreads();
            continue;
        }
        NxSet S, D;
        S.live = mb->live[lane]; S.occ = mb->occ[lane];
        S.used = mb->used; S.mask = mb->mask; S.ident = mb->ident != 0; S.perfect = mb->perfect != 0;
        S.tab = L.tab[mb->srctab];
        const uint64_t row = mb->row[lane];
        const int deg = mb->deg;
        uint16_t *dst = L.tab[mb->dsttab];
        WSYNC();
        { NX_T0 nx_and_adj(c, L, D, S, S.used, row, deg, dst); NX_T1(21) }
        mb->live[lane] = D.live; mb->occ[lane] = D.occ;
        if (lane == 0) { mb->used = D.used; mb->mask = D.mask; mb->ident = D.ident; mb->perfect = D.perfect; }
        __syncthreads();
    }
}

// RF = the first clique of size omega in networkx.find_cliques order.  false: the bounded searches ran out of nodes.
template <bool TWO>
__device__ __forceinline__ bool nx_walk(CqCtx &c, NxLds &L, int Kb, uint64_t ALL, int omega, uint64_t WIT, uint64_t &RF)
{
    const int lane = c.lane, nw = c.nw;
    NxSet subg, cand, ext;
    uint64_t elist = 0;                                                 // ext_u of <= 4 members: packed pop order (nx_sub_adj_tiny)
    int en = -1;                                                        // ... its remaining count, -1 = ext is the set `ext`
    int cur = 0;
    cand.live = ALL; cand.ident = true; cand.perfect = false; cand.used = Kb; cand.occ = 0; cand.tab = L.tab[2];                     // set(G): every key < table size
    cand.mask = (Kb < 5 ? 8 : Kb < 19 ? 32 : Kb < 77 ? 128 : Kb < 307 ? 512 : 2048) - 1;
    subg = cand; subg.tab = L.tab[0];                                                                          // cand.copy(): likewise
    RF = 0;
    if (lane < 2) nx_ck[lane][19] = 0;                                  // nothing remembered yet (nx_build_fast)
    WSYNC();
    int size = 0;
    bool entered = true;                                                // a node was just entered: its pivot and ext_u are due
    // (every helper appears ONCE in this loop: with the set-up of a node written out before the loop as well, the kernel grew past
    // what this compiler turns into a working binary - the build faulted in code that never ran)
#pragma nounroll
    for (;;) {
        if (entered) {
            entered = false;
            // Every node the walk enters holds a clique of size omega between Q and Q + cand (the root trivially, a child by the existence
            // answer that let the walk in).  When cand has exactly the omega - |Q| vertices still needed, that clique IS Q + cand, it is
            // the only clique of that size in the subtree, and no order has to be followed to find it: done.  (Round 5: this ends the
            // walk as soon as the last contested vertex is decided - before it, the uncontested rest of the clique was walked level by
            // level, the last ~19 levels with explicit 8- / 32-slot tables: 115 us of a 120-us problem of 60 correspondences.)
#ifndef NX_EXP_NOSHORT
            if (cand.used == omega - size) { RF |= cand.live; return true; }
#endif
            NX_CNT(6)
#ifndef NX_EXP_NOBULK
            { NX_T0 nx_bulk(c, subg, cand, RF, size); NX_T1(0) }
#endif
            int pu;
            { NX_T0 pu = nx_pivot(c, subg, cand.live); NX_T1(1) }
            const uint64_t prow = (lane < nw) ? c.A[(int64_t)pu * c.as + lane] : 0ull;
            const int pdeg = bs_count(prow);
            { NX_T0 if (!nx_sub_adj_tiny(c, L, cand, prow, pdeg, elist, en)) { en = -1; nx_sub_adj(c, L, ext, cand, prow, pdeg); } NX_T1(2) }
        }
        NX_CNT(7)
        int q;
        if (en >= 0) {
            q = en ? (int)(elist & 0xffffull) : -1;
            elist >>= 16; en--;
        } else
            { NX_T0 q = nx_pop(c, ext); NX_T1(3) }
        if (q < 0) return false;                                        // cannot happen while the existence answers are exact
        const uint64_t bq = bit_if(lane, q);
        cand.live &= ~bq;                                               // cand.remove(q): a dummy stays in its slot
        cand.used--;
        const uint64_t row = (lane < nw) ? c.A[(int64_t)q * c.as + lane] : 0ull;
        const uint64_t Sq = subg.live & row, Cq = cand.live & row;
        const int s1 = size + 1;
        if (!__ballot(Sq != 0)) {                                       // maximal clique Q + q
            if (s1 >= omega) { RF |= bq; return true; }
            continue;
        }
        const int ncq = bs_count(Cq);
        if (ncq == 0) continue;
        const int need = omega - s1;
        if (need <= 0) { RF |= bq; return true; }
        if (ncq < need) continue;
        const uint64_t QB = RF | bq;
        bool ok = !__ballot((QB & ~WIT) != 0) && !__ballot((WIT & ~QB & ~Cq) != 0);      // the witness lives in this subtree
        if (!ok) {
            uint64_t RQ = 0;
#ifdef NX_EXP_STATS
            c.nq++;
#endif
            int got;
            { NX_T0 got = cq_solve<TWO>(c, Cq, need - 1, need, RQ); NX_T1(4) }
            if (!c.complete) return false;
            if (got >= need) { ok = true; WIT = QB | RQ; }
        }
        if (!ok) continue;
#ifndef NX_EXP_NOSHORT
        if (ncq == need) { RF = QB | Cq; return true; }                 // (the same one level earlier: no child sets to build)
#endif
        // descend: subg_q = subg & adj[q], cand_q = cand & adj[q] - two independent replays, the second one on the helper wavefront
        // (round 5: a lone wavefront issues an instruction every ~7 cycles, and the two sets were 2/3 of a level)
        const int deg = bs_count(row);
        NxSet nsub, ncand;
        if (TWO) {
            NxMail *mb = L.mail;
            mb->live[lane] = cand.live; mb->occ[lane] = cand.occ; mb->row[lane] = row;
            if (lane == 0) {
                mb->cmd = 1; mb->used = cand.used; mb->mask = cand.mask; mb->ident = cand.ident; mb->perfect = cand.perfect;
                mb->srctab = 2 + cur; mb->dsttab = 2 + (cur ^ 1); mb->deg = deg;
            }
            __syncthreads();                                            // the helper starts (nx_helper)
            { NX_T0 nx_and_adj(c, L, nsub, subg, subg.used, row, deg, L.tab[cur ^ 1]); NX_T1(5) }
            { NX_T0 __syncthreads(); NX_T1(20) }                        // ... and is done
            ncand.live = mb->live[lane]; ncand.occ = mb->occ[lane];
            ncand.used = mb->used; ncand.mask = mb->mask; ncand.ident = mb->ident != 0; ncand.perfect = mb->perfect != 0;
            ncand.tab = L.tab[2 + (cur ^ 1)];
        } else {
#pragma nounroll
            for (int which = 0; which < 2; which++) {
                const NxSet &S = which ? cand : subg;
                NxSet &D = which ? ncand : nsub;
                { NX_T0 nx_and_adj(c, L, D, S, S.used, row, deg, L.tab[2 * which + (cur ^ 1)]); NX_T1(5) }
            }
        }
        subg = nsub; cand = ncand; cur ^= 1;
        RF = QB; size = s1;
        entered = true;
    }
}

template <bool TWO>
__global__ __launch_bounds__(TWO ? 128 : 64) void max_clique_kernel(const uint64_t *__restrict__ adj_g,
                                                        const int32_t *__restrict__ count, int K, int kstride,
                                                        int nws, long long node_limit,
                                                        uint64_t *__restrict__ stack_g,
                                                        uint8_t *__restrict__ mask_out,
                                                        int32_t *__restrict__ n_in, int32_t *__restrict__ flags,
                                                        const int32_t *__restrict__ order)
{
    extern __shared__ __align__(16) unsigned char cq_smem[];
    const int b = order ? order[blockIdx.x] : blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int Kb = count ? min(count[b], K) : K;
    uint8_t *mask = mask_out + (int64_t)b * kstride;
    if (Kb <= 0) {
        if (threadIdx.x == 0) { n_in[b] = 0; flags[b] = 1; }
        return;
    }
    const int nw = (Kb + 63) >> 6;                 // active words per bitset
    // ---- LDS carve: three short[K+2] level arrays, sw[16] u64, the walk's tables, adjacency (compact stride)
    short *lsize = reinterpret_cast<short *>(cq_smem);
    short *lv = lsize + (K + 2), *lstage = lv + (K + 2);
    uint64_t *sw = reinterpret_cast<uint64_t *>(cq_smem + ((3 * sizeof(short) * (K + 2) + 15) & ~(size_t)15));
    unsigned char *nx_mem = reinterpret_cast<unsigned char *>(sw + 32);        // (16 words per wavefront) the walk's tables and lists (nx_lds_bytes)
    uint64_t *adj_l = reinterpret_cast<uint64_t *>(nx_mem + nx_lds_bytes(K));
    const uint64_t *Ag = adj_g + (int64_t)b * kstride * nws;
    const bool use_lds = (Kb * nw <= CQ_LDS_ADJ_WORDS);
    CqCtx c;
    if (use_lds) {
        if (wv == 0)
            for (int i = lane; i < Kb * nw; i += 64) {
                const int r = i / nw, w = i - r * nw;
                adj_l[i] = Ag[(int64_t)r * nws + w];
            }
        c.A = adj_l; c.as = nw;
    } else { c.A = Ag; c.as = nws; }
    WSYNC();
    c.nw = nw; c.nws = nws; c.lane = lane; c.wv = wv; c.sw = sw + 16 * wv; c.sw0 = sw;
    c.stk = stack_g + (int64_t)b * (kstride + 2) * 2 * nws;
    c.lsize = lsize; c.lv = lv; c.lstage = lstage;
    c.nodes = 0; c.node_limit = node_limit; c.complete = true;
    // The workgroup is two wavefronts.  The first runs the search; the second waits at the workgroup barrier for the cand chain of the
    // walk's descents (nx_helper) and leaves when the first says so.  They meet at those barriers ONLY - everything else in this file
    // synchronises a wavefront with itself (WSYNC).
    { NxLds Lt; nx_lds_carve(Lt, nx_mem, K, 0); c.mail = Lt.mail; }
    if (TWO && wv == 1) {
        NxLds L1;
        nx_lds_carve(L1, nx_mem, K, 1);
        nx_helper(c, L1);
        return;
    }
    NxLds L;
    nx_lds_carve(L, nx_mem, K, 0);
#ifdef NX_EXP_STATS
    if (lane < 24) nx_acc[lane] = 0;
    WSYNC();
#endif

    uint64_t ALL = 0;                               // all-vertices set in word-per-lane layout
    if (lane < nw) {
        const int lo = lane * 64;
        if (Kb >= lo + 64) ALL = ~0ull;
        else if (Kb > lo) ALL = (1ull << (Kb - lo)) - 1ull;
    }
    // ---- phase 1: omega and one maximum clique (the witness)
    uint64_t WIT = 0;
    const int omega = cq_solve<TWO>(c, ALL, 0, 0x7fffffff, WIT);
#ifdef NX_EXP_STATS
    c.nq = 0; c.nodes1 = c.nodes;
#endif
    // ---- phase 2: the first clique of size omega in networkx.find_cliques order (nx_walk)
    uint64_t REC = WIT;
    // ... unless there is only one: every clique of size omega lies in the (omega - 1)-core of the graph (vertices with at least omega - 1
    // neighbours inside it, to a fixed point), and a core of exactly omega vertices IS that clique - the witness - whatever the order.
    // A few degree passes (~1 us each) against a walk of 100+ us: a quarter of the real / bench-like pairs (5 of 16 lone sets).
    bool unique = false;
#ifndef NX_EXP_NOCORE
    if (c.complete && omega > 1) {
        uint64_t P = ALL;
        for (;;) {
            const int cnt = bs_count(P);
            if (cnt <= omega) { unique = cnt == omega; break; }
            if (lane < 16) c.sw[lane] = P;
            WSYNC();
            uint64_t RM = 0, UN = 0, PE = 0;
            int key = 0x7fffffff;
            cq_degrees(c, c.sw, 0, nw, omega - 1, cnt, RM, UN, PE, key);
            WSYNC();
            if (!__ballot(RM != 0)) break;
            P &= ~RM;
        }
    }
#endif
#ifdef NX_EXP_NOWALK
    if (false) {
#else
    if (c.complete && omega > 0 && !unique) {
#endif
        uint64_t RF = 0;
        { NX_T0 if (nx_walk<TWO>(c, L, Kb, ALL, omega, WIT, RF)) REC = RF; NX_T1(11) }
    }
    if (TWO) {
        if (lane == 0) L.mail->cmd = 0;                                 // the helper may go
        __syncthreads();
    }

    // ---- emit
    if (lane < 16) sw[lane] = REC;
    WSYNC();
    for (int u = lane; u < Kb; u += 64) mask[u] = (uint8_t)((sw[u >> 6] >> (u & 63)) & 1ull);
    for (int u = Kb + lane; u < kstride && u < K; u += 64) mask[u] = 0;
    const int cnt = bs_count(REC);
    if (lane == 0) { n_in[b] = cnt; flags[b] = c.complete ? 1 : 0; }
#ifdef NX_EXP_STATS
    WSYNC();
    if (lane < 24) atomicAdd(&nx_prof[lane], nx_acc[lane]);
    if (lane == 0) flags[b] |= (min(c.nq, 255) << 8) | ((int)min((c.nodes - c.nodes1), 32767ll) << 16);
    if (lane == 0) n_in[b] = cnt | ((int)min(c.nodes1, 32767ll) << 16);
#endif
}

// order[0..B) = the problems by falling number of correspondences (a counting sort in one workgroup): the kernel's time is set by its
// longest problems - the pairs right after a re-detection - and a workgroup that starts last should not be one of them
// (bucket = 1024 - clamp(count, 0, K) >> shift: a negative count sorts as 0, and K >> shift above 1024 saturates into bucket 0 - the
// caller picks shift so that it does not: launch_order_by_count)
__global__ __launch_bounds__(1024) void cq_order_kernel(const int32_t *__restrict__ count, int B, int K, int32_t *__restrict__ order, int shift)
{
    __shared__ int hist[1026];
    const int t = threadIdx.x;
    for (int i = t; i <= 1025; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int i = t; i < B; i += 1024) atomicAdd(&hist[1024 - min(max(min(count[i], K), 0) >> shift, 1024)], 1);       // bucket 0 = the largest problems
    __syncthreads();
    if (t < 64) {                                                                                   // exclusive prefix over 1025 buckets: 17 per lane
        int loc[17], sum = 0;
        for (int k = 0; k < 17; k++) { const int i = t * 17 + k; loc[k] = i <= 1024 ? hist[i] : 0; sum += loc[k]; }
        int inc = sum;
        for (int d = 1; d < 64; d <<= 1) { const int n = __shfl_up(inc, d); if (t >= d) inc += n; }
        int run = inc - sum;
        for (int k = 0; k < 17; k++) { const int i = t * 17 + k; if (i <= 1024) hist[i] = run; run += loc[k]; }
    }
    __syncthreads();
    for (int i = t; i < B; i += 1024) order[atomicAdd(&hist[1024 - min(max(min(count[i], K), 0) >> shift, 1024)], 1)] = i;
}

hipError_t launch_order_by_count(hipStream_t st, const int32_t *count, int B, int cmax, int32_t *order, int shift)
{
    if (shift < 0 || shift > 30 || (cmax >> shift) > 1024) return hipErrorInvalidValue;             // the 1025 buckets would saturate
    hipLaunchKernelGGL(cq_order_kernel, dim3(1), dim3(1024), 0, st, count, B, cmax, order, shift);
    return hipGetLastError();
}

// adj rows have stride nws words; stack scratch: B x (kstride+2) x 2 x nws words
hipError_t launch_max_clique(hipStream_t st, const uint64_t *adj, const int32_t *count, int K,
                             int kstride, int nws, int B, int64_t node_limit, uint64_t *stack,
                             uint8_t *mask, int32_t *n_in, int32_t *flags, int32_t *order)
{
    if (B <= 0 || K <= 0) return hipSuccess;
    if (!count || B < 512 || getenv("ROAM_CLIQUE_NO_ORDER")) order = nullptr;
    if (order) hipLaunchKernelGGL(cq_order_kernel, dim3(1), dim3(1024), 0, st, count, B, K, order, 0);
    size_t lds = ((3 * sizeof(short) * (size_t)(K + 2) + 15) & ~(size_t)15) + 32 * 8 + nx_lds_bytes(K);
    const size_t kw = (size_t)K * ((K + 63) / 64);
    lds += 8 * (kw < CQ_LDS_ADJ_WORDS ? kw : (size_t)CQ_LDS_ADJ_WORDS);
    if (node_limit <= 0) node_limit = 300000;
    // Two wavefronts per problem (the walk's two set chains side by side) shorten a problem by a quarter and halve the problems a CU
    // holds.  4096 copies of ONE problem, same box: K 139 1.21 -> 1.63 ms with two, a 240-feature pair 3.38 -> 2.94; the default
    // bench step (4096 different problems: the long ones set the time) 81.3 -> 80.9 ms, one sequence without motion distortion
    // 1 390 -> 1 526 scan-pairs/s.  ROAM_CLIQUE_TWO_WAVES=0 / 1 forces one / two (A/B runs).
    static const int force = [] { const char *e = getenv("ROAM_CLIQUE_TWO_WAVES"); return e ? atoi(e) : -1; }();
    const bool two = force >= 0 ? force != 0 : true;
    if (two)
        hipLaunchKernelGGL(max_clique_kernel<true>, dim3(B), dim3(128), lds, st, adj, count, K, kstride, nws,
                           (long long)node_limit, stack, mask, n_in, flags, (const int32_t *)order);
    else
        hipLaunchKernelGGL(max_clique_kernel<false>, dim3(B), dim3(64), lds, st, adj, count, K, kstride, nws,
                           (long long)node_limit, stack, mask, n_in, flags, (const int32_t *)order);
    return hipGetLastError();
}
