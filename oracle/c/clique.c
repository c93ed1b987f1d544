/* ORACLE — TEST INFRASTRUCTURE ONLY (see peaks.c header).
 *
 * CPU restatement of outlierRejection.rejectOutliers (reference outlierRejection.py:16-95):
 *   D0 = cdist(prev,prev), D1 = cdist(new,new) in float64 (scipy euclidean: sqrt(sum d*d)),
 *   A[i][j] = |D0 - D1| <= thr_px  (thr = 0.5 m / 0.0864 m/px, :10-11,57-58),
 *   inlier set = a MAXIMUM clique of A (:63-78).
 * The reference keeps the first strictly-largest clique in networkx.find_cliques order,
 * which depends on CPython set iteration order and is not reproducible when several
 * maximum cliques exist (16 of size 67 on its own 95-point fixture).  The contract here
 * (SURVEY.md §7.3-2): the SIZE always equals the reference's; the SET is the
 * lexicographically smallest maximum clique (sorted vertex lists compared) — which is the
 * reference's set whenever the maximum clique is unique.
 *
 * Method (independent of the device code's search order): omega by a colour-bounded
 * branch and bound (Tomita-style), then the lexicographic minimum by fixing vertices in
 * ascending order, each time asking the same solver whether a clique of the required
 * size still exists among the remaining higher-numbered common neighbours.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define MAXW 32                    /* up to 2048 vertices */
typedef struct { uint64_t w[MAXW]; } Bits;

static int g_nw;
static Bits *g_adj;
static int g_best;
static int g_target;               /* stop as soon as a clique >= target is found (0: never) */
static int g_found;
static int64_t g_nodes;

static inline int popc(const Bits *b) { int c = 0; for (int i = 0; i < g_nw; i++) c += __builtin_popcountll(b->w[i]); return c; }
static inline int empty(const Bits *b) { for (int i = 0; i < g_nw; i++) if (b->w[i]) return 0; return 1; }
static inline int first_bit(const Bits *b) { for (int i = 0; i < g_nw; i++) if (b->w[i]) return i * 64 + __builtin_ctzll(b->w[i]); return -1; }

/* greedy sequential colouring of P; order[] receives vertices, col[] their colour bound */
static int colour_sort(const Bits *P, int *order, int *col)
{
    Bits U = *P, Q;
    int n = 0, c = 0;
    while (!empty(&U)) {
        c++;
        Q = U;
        while (!empty(&Q)) {
            int v = first_bit(&Q);
            Q.w[v >> 6] &= ~(1ull << (v & 63));
            U.w[v >> 6] &= ~(1ull << (v & 63));
            for (int i = 0; i < g_nw; i++) Q.w[i] &= ~g_adj[v].w[i];
            order[n] = v; col[n] = c; n++;
        }
    }
    return n;
}

static void expand(Bits *P, int size)
{
    if (g_found) return;
    g_nodes++;
    int np = popc(P);
    if (np == 0) {
        if (size > g_best) { g_best = size; if (g_target && g_best >= g_target) g_found = 1; }
        return;
    }
    int *order = (int *)malloc(sizeof(int) * np * 2), *col = order + np;
    colour_sort(P, order, col);
    for (int k = np - 1; k >= 0; k--) {
        if (size + col[k] <= g_best) break;
        int v = order[k];
        Bits N;
        for (int i = 0; i < g_nw; i++) N.w[i] = P->w[i] & g_adj[v].w[i];
        expand(&N, size + 1);
        if (g_found) break;
        P->w[v >> 6] &= ~(1ull << (v & 63));
    }
    free(order);
}

/* max clique size inside vertex set S (lower bound lb already known: prune below it) */
static int omega_of(const Bits *S, int lb, int target)
{
    Bits P = *S;
    g_best = lb; g_target = target; g_found = 0;
    expand(&P, 0);
    return g_best;
}

/* adjacency from the two point sets; adj_out: K rows of nw u64 words (bit j of row i) */
void oracle_consistency_graph(const float *prev, const float *next, int K, double thr,
                              uint64_t *adj_out, int nw)
{
    memset(adj_out, 0, sizeof(uint64_t) * (size_t)K * nw);
    for (int i = 0; i < K; i++)
        for (int j = 0; j < K; j++) {
            if (i == j) continue;                 /* self loops are ignored by the clique search */
            double ax = (double)prev[2 * i] - (double)prev[2 * j], ay = (double)prev[2 * i + 1] - (double)prev[2 * j + 1];
            double bx = (double)next[2 * i] - (double)next[2 * j], by = (double)next[2 * i + 1] - (double)next[2 * j + 1];
            double d0 = sqrt(ax * ax + ay * ay), d1 = sqrt(bx * bx + by * by);
            if (fabs(d0 - d1) <= thr) adj_out[(size_t)i * nw + (j >> 6)] |= 1ull << (j & 63);
        }
}

/* mask_out[K] u8; returns clique size.  nodes_out (optional) = search nodes visited */
int oracle_max_clique_lex(const uint64_t *adj, int K, int nw, uint8_t *mask_out, int64_t *nodes_out)
{
    g_nw = nw;
    g_adj = (Bits *)calloc((size_t)K, sizeof(Bits));
    for (int i = 0; i < K; i++) memcpy(g_adj[i].w, adj + (size_t)i * nw, sizeof(uint64_t) * nw);
    g_nodes = 0;
    Bits all; memset(&all, 0, sizeof(all));
    for (int i = 0; i < K; i++) all.w[i >> 6] |= 1ull << (i & 63);
    int omega = K ? omega_of(&all, 0, 0) : 0;
    memset(mask_out, 0, (size_t)K);
    Bits C = all;
    int need = omega;
    for (int v = 0; v < K && need > 0; v++) {
        if (!((C.w[v >> 6] >> (v & 63)) & 1)) continue;
        Bits S; memset(&S, 0, sizeof(S));
        for (int i = 0; i < nw; i++) S.w[i] = C.w[i] & g_adj[v].w[i];
        /* only higher-numbered vertices */
        for (int u = 0; u <= v; u++) S.w[u >> 6] &= ~(1ull << (u & 63));
        int ok;
        if (need - 1 == 0) ok = 1;
        else if (popc(&S) < need - 1) ok = 0;
        else ok = omega_of(&S, need - 2, need - 1) >= need - 1;
        if (ok) { mask_out[v] = 1; C = S; need--; }
    }
    if (nodes_out) *nodes_out = g_nodes;
    free(g_adj);
    return omega;
}
