/* ORACLE — TEST INFRASTRUCTURE ONLY (see peaks.c header).
 *
 * CPU restatement of outlierRejection.rejectOutliers (reference outlierRejection.py:16-95):
 *   D0 = cdist(prev,prev), D1 = cdist(new,new) in float64 (scipy euclidean: sqrt(sum d*d)),
 *   A[i][j] = |D0 - D1| <= thr_px  (thr = 0.5 m / 0.0864 m/px, :10-11,57-58),
 *   inlier set = a MAXIMUM clique of A (:63-78).
 * The reference keeps the first strictly-largest clique in networkx.find_cliques order.  With
 * integer nodes that order is deterministic (it follows from networkx's search and CPython's set
 * table, not from a randomised hash) and ties are the norm on real data (16 maximum cliques of
 * size 67 on its own 95-point fixture), so it is part of the contract: oracle_max_clique_nx in the
 * second half of this file restates it and is what oracle.rejectOutliers uses (round 4; rounds 1-3
 * returned the lexicographically smallest maximum clique and called the reference's order
 * irreproducible - wrong).
 *
 * First half: omega by a colour-bounded branch and bound (Tomita-style) - independent of the
 * device code's search order - and oracle_max_clique_lex, the old lexicographic rule, kept as a
 * second opinion on the SIZE and to show in the tests that the two rules differ.
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

/* ================================================================================================
 * The reference's own tie-break: rejectOutliers keeps the FIRST strictly-largest clique in the order
 * networkx.find_cliques yields them (outlierRejection.py:63-75).  That order is deterministic - the nodes are
 * Python ints, whose hash is the value - but it is a property of two implementations, restated here exactly:
 *
 *   networkx (2.6 .. 3.4; checked against the live 3.4.2) clique.py find_cliques: iterative Bron-Kerbosch with
 *     Tomita pivoting on Python sets - adj[u] = {v for v in G[u] if v != u} (G[u] in ascending order for a graph
 *     made from a dense 0/1 matrix: convert_matrix.from_numpy_array adds the edges row by row), cand = set(G),
 *     subg = cand.copy(), pivot u = max(subg, key=|cand & adj[u]|) (FIRST maximum in subg's iteration order),
 *     ext_u = cand - adj[u], q = ext_u.pop(), cand.remove(q), subg_q = subg & adj[q], cand_q = cand & adj[q];
 *   CPython (3.7 .. 3.12; checked against the live 3.10.12) Objects/setobject.c: open addressing with 9 linear
 *     probes then the perturbed jump i*5+1+perturb, growth to used*4 once fill*5 >= mask*3, set_copy /
 *     set_merge (table of the smallest power of two > 2*used; slot-for-slot copy when the masks agree and there are
 *     no dummies), set_intersection (iterates the SMALLER operand - the right one on equal sizes - and adds its
 *     members found in the other to a fresh set), set_difference (iterates the left operand unless
 *     len(left)/4 > len(right): then copy + discard), set_pop (first live slot from the finger), remove -> dummy.
 *
 * Two walks: oracle_max_clique_nx(..., prune=0) enumerates every maximal clique like the reference does (exponential;
 * small graphs, used to validate the other); prune=1 first computes omega with the colour-bounded search above and
 * then walks the SAME tree but descends into a child only if a clique of the size still needed exists among its
 * candidates - the child's sets are new objects, so skipping a subtree leaves the parent's state (and therefore the
 * order of everything that follows) untouched, and the first clique of size omega reached is the reference's. */
#define PS_EMPTY (-1)
#define PS_DUMMY (-2)
typedef struct { int16_t *tab; int mask, fill, used, finger; } PSet;

static PSet *ps_new(void)
{
    PSet *s = (PSet *)malloc(sizeof(PSet));
    s->tab = (int16_t *)malloc(sizeof(int16_t) * 8);
    for (int i = 0; i < 8; i++) s->tab[i] = PS_EMPTY;
    s->mask = 7; s->fill = s->used = s->finger = 0;
    return s;
}
static void ps_free(PSet *s) { if (s) { free(s->tab); free(s); } }

static void ps_insert_clean(int16_t *tab, int mask, int key)
{
    unsigned perturb = (unsigned)key, i = (unsigned)key & (unsigned)mask;
    for (;;) {
        int probes = (i + 9 <= (unsigned)mask) ? 9 : 0;
        unsigned j = i;
        do { if (tab[j] == PS_EMPTY) { tab[j] = (int16_t)key; return; } j++; } while (probes--);
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (unsigned)mask;
    }
}
static void ps_resize(PSet *s, int minused)
{
    int newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    int16_t *nt = (int16_t *)malloc(sizeof(int16_t) * (size_t)newsize);
    for (int i = 0; i < newsize; i++) nt[i] = PS_EMPTY;
    for (int i = 0; i <= s->mask; i++) if (s->tab[i] >= 0) ps_insert_clean(nt, newsize - 1, s->tab[i]);
    free(s->tab); s->tab = nt; s->mask = newsize - 1; s->fill = s->used;
}
static int ps_lookup(const PSet *s, int key)
{
    unsigned perturb = (unsigned)key, mask = (unsigned)s->mask, i = (unsigned)key & mask;
    for (;;) {
        int probes = (i + 9 <= mask) ? 9 : 0;
        unsigned j = i;
        do { if (s->tab[j] == PS_EMPTY) return -1; if (s->tab[j] == key) return (int)j; j++; } while (probes--);
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & mask;
    }
}
static void ps_add(PSet *s, int key)
{
    unsigned perturb = (unsigned)key, mask = (unsigned)s->mask, i = (unsigned)key & mask;
    int freeslot = -1, hit = -1;
    while (hit < 0) {
        int probes = (i + 9 <= mask) ? 9 : 0;
        unsigned j = i;
        do {
            if (s->tab[j] == PS_EMPTY) { hit = (int)j; break; }
            if (s->tab[j] == key) return;
            if (s->tab[j] == PS_DUMMY) freeslot = (int)j;
            j++;
        } while (probes--);
        if (hit >= 0) break;
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & mask;
    }
    if (freeslot >= 0) { s->tab[freeslot] = (int16_t)key; s->used++; return; }
    s->tab[hit] = (int16_t)key; s->fill++; s->used++;
    if ((unsigned)s->fill * 5 < mask * 3) return;
    ps_resize(s, s->used > 50000 ? s->used * 2 : s->used * 4);
}
static int ps_discard(PSet *s, int key)
{
    int j = ps_lookup(s, key);
    if (j < 0) return 0;
    s->tab[j] = PS_DUMMY; s->used--;
    return 1;
}
static int ps_pop(PSet *s)
{
    int i = s->finger & s->mask;
    while (s->tab[i] < 0) { i++; if (i > s->mask) i = 0; }
    int key = s->tab[i];
    s->tab[i] = PS_DUMMY; s->used--; s->finger = i + 1;
    return key;
}
static PSet *ps_copy(const PSet *o)
{
    PSet *s = ps_new();
    if (o->used == 0) return s;
    if ((s->fill + o->used) * 5 >= s->mask * 3) ps_resize(s, (s->used + o->used) * 2);
    if (s->mask == o->mask && o->fill == o->used) {
        memcpy(s->tab, o->tab, sizeof(int16_t) * (size_t)(o->mask + 1));
        s->fill = o->fill; s->used = o->used;
        return s;
    }
    s->fill = s->used = o->used;
    for (int i = 0; i <= o->mask; i++) if (o->tab[i] >= 0) ps_insert_clean(s->tab, s->mask, o->tab[i]);
    return s;
}
static PSet *ps_and(const PSet *so, const PSet *other)
{
    PSet *r = ps_new();
    if (other->used > so->used) { const PSet *t = so; so = other; other = t; }
    for (int i = 0; i <= other->mask; i++) if (other->tab[i] >= 0 && ps_lookup(so, other->tab[i]) >= 0) ps_add(r, other->tab[i]);
    return r;
}
static PSet *ps_sub(const PSet *so, const PSet *other)
{
    if ((so->used >> 2) > other->used) {        /* set_copy_and_difference; its "resize the dummies away" rule cannot trigger: */
        PSet *r = ps_copy(so);                  /* fewer than used/4 dummies against a table of more than 2*used slots */
        for (int i = 0; i <= other->mask; i++) if (other->tab[i] >= 0) ps_discard(r, other->tab[i]);
        return r;
    }
    PSet *r = ps_new();
    for (int i = 0; i <= so->mask; i++) if (so->tab[i] >= 0 && ps_lookup(other, so->tab[i]) < 0) ps_add(r, so->tab[i]);
    return r;
}
static void ps_bits(const PSet *s, Bits *b)
{
    memset(b, 0, sizeof(*b));
    for (int i = 0; i <= s->mask; i++) if (s->tab[i] >= 0) b->w[s->tab[i] >> 6] |= 1ull << (s->tab[i] & 63);
}

typedef struct { PSet *subg, *cand, *ext; } NxLevel;

/* mask_out[K] u8 = the first strictly-largest clique in networkx.find_cliques order; returns its size.
 * stats_out (optional, 4 x int64): maximal cliques yielded before stopping, children skipped by the bound,
 * existence searches run, tree nodes entered.  prune = 0: plain enumeration of every maximal clique. */
static int nx_core(const uint64_t *adjw, int K, int nw, int prune, uint8_t *mask_out, int64_t *stats_out, int collect_cap, int *n_collected)
{
    int64_t st_yield = 0, st_skip = 0, st_query = 0, st_nodes = 0;
    memset(mask_out, 0, (size_t)(K > 0 ? K : 0) * (size_t)(collect_cap > 0 ? collect_cap : 1));
    if (stats_out) memset(stats_out, 0, sizeof(int64_t) * 4);
    if (K <= 0) return 0;
    g_nw = nw;
    g_adj = (Bits *)calloc((size_t)K, sizeof(Bits));
    for (int i = 0; i < K; i++) memcpy(g_adj[i].w, adjw + (size_t)i * nw, sizeof(uint64_t) * nw);
    g_nodes = 0;
    int omega = 0;
    if (prune) {
        Bits all; memset(&all, 0, sizeof(all));
        for (int i = 0; i < K; i++) all.w[i >> 6] |= 1ull << (i & 63);
        omega = omega_of(&all, 0, 0);
    }
    PSet **adj = (PSet **)malloc(sizeof(PSet *) * (size_t)K);
    for (int u = 0; u < K; u++) {                         /* {v for v in G[u] if v != u}: ascending insertion */
        adj[u] = ps_new();
        for (int v = 0; v < K; v++) if (v != u && ((g_adj[u].w[v >> 6] >> (v & 63)) & 1)) ps_add(adj[u], v);
    }
    int *Q = (int *)malloc(sizeof(int) * (size_t)(K + 1)), *best = (int *)malloc(sizeof(int) * (size_t)(K + 1));
    NxLevel *stack = (NxLevel *)malloc(sizeof(NxLevel) * (size_t)(K + 1));
    int nq = 0, sp = 0, bestn = 0;
    PSet *cand = ps_new();
    for (int u = 0; u < K; u++) ps_add(cand, u);          /* set(G) */
    PSet *subg = ps_copy(cand);
    nq = 1;                                               /* Q.append(None) */
    PSet *ext = NULL;
#define NX_PIVOT_EXT()                                                                                      \
    do {                                                                                                    \
        Bits cb; ps_bits(cand, &cb);                                                                        \
        int pu = -1, pl = -1;                                                                               \
        for (int i_ = 0; i_ <= subg->mask; i_++) {                                                          \
            const int u_ = subg->tab[i_];                                                                   \
            if (u_ < 0) continue;                                                                           \
            int l_ = 0;                                                                                     \
            for (int w_ = 0; w_ < nw; w_++) l_ += __builtin_popcountll(cb.w[w_] & g_adj[u_].w[w_] & ~((w_ == (u_ >> 6)) ? (1ull << (u_ & 63)) : 0ull)); \
            if (l_ > pl) { pl = l_; pu = u_; }                                                              \
        }                                                                                                   \
        ext = ps_sub(cand, adj[pu]);                                                                        \
        st_nodes++;                                                                                         \
    } while (0)
    NX_PIVOT_EXT();
    int done = 0;
    while (!done) {
        if (ext->used > 0) {
            const int q = ps_pop(ext);
            ps_discard(cand, q);                          /* cand.remove(q) */
            Q[nq - 1] = q;
            PSet *subg_q = ps_and(subg, adj[q]);
            if (subg_q->used == 0) {                      /* yield Q[:] */
                st_yield++;
                if (collect_cap > 0) {                    /* every clique of size omega, in the order they are yielded */
                    if (nq == omega && *n_collected < collect_cap) {
                        uint8_t *m = mask_out + (size_t)(*n_collected) * K;
                        for (int i = 0; i < nq; i++) m[Q[i]] = 1;
                        (*n_collected)++;
                    }
                    bestn = omega;
                } else if (nq > bestn) { bestn = nq; memcpy(best, Q, sizeof(int) * (size_t)nq); if (prune && bestn >= omega) done = 1; }
                ps_free(subg_q);
            } else {
                PSet *cand_q = ps_and(cand, adj[q]);
                int descend = cand_q->used > 0;
                if (descend && prune) {                   /* a clique of size omega - nq among cand_q ? */
                    const int need = omega - nq;
                    if (cand_q->used < need) descend = 0;
                    else if (need > 0) {
                        Bits cb; ps_bits(cand_q, &cb);
                        st_query++;
                        descend = omega_of(&cb, need - 1, need) >= need;
                    }
                    if (!descend) st_skip++;
                }
                if (descend) {
                    stack[sp].subg = subg; stack[sp].cand = cand; stack[sp].ext = ext; sp++;
                    nq++;
                    subg = subg_q; cand = cand_q;
                    NX_PIVOT_EXT();
                } else { ps_free(subg_q); ps_free(cand_q); }
            }
        } else {
            nq--;                                         /* Q.pop() */
            ps_free(subg); ps_free(cand); ps_free(ext);
            subg = cand = ext = NULL;
            if (sp == 0) break;                           /* stack.pop() raises IndexError: the generator ends */
            sp--;
            subg = stack[sp].subg; cand = stack[sp].cand; ext = stack[sp].ext;
        }
    }
#undef NX_PIVOT_EXT
    if (collect_cap <= 0) for (int i = 0; i < bestn; i++) mask_out[best[i]] = 1;
    ps_free(subg); ps_free(cand); ps_free(ext);
    while (sp > 0) { sp--; ps_free(stack[sp].subg); ps_free(stack[sp].cand); ps_free(stack[sp].ext); }
    for (int u = 0; u < K; u++) ps_free(adj[u]);
    free(adj); free(Q); free(best); free(stack); free(g_adj);
    if (stats_out) { stats_out[0] = st_yield; stats_out[1] = st_skip; stats_out[2] = st_query; stats_out[3] = st_nodes; }
    return bestn;
}

int oracle_max_clique_nx(const uint64_t *adjw, int K, int nw, int prune, uint8_t *mask_out, int64_t *stats_out)
{
    return nx_core(adjw, K, nw, prune, mask_out, stats_out, 0, NULL);
}

/* ALL maximum cliques in the order find_cliques yields them (bounded walk with backtracking): masks_out (cap, K) u8,
 * returns how many were written.  For the tests that show which of the tied cliques a reference run picked. */
int oracle_max_cliques_nx_all(const uint64_t *adjw, int K, int nw, uint8_t *masks_out, int cap)
{
    int n = 0;
    nx_core(adjw, K, nw, 1, masks_out, NULL, cap, &n);
    return n;
}

/* the set restatement alone, for the test against the live interpreter: a little program of set operations.
 * ops (n,3) int32 rows [opcode, a, b] on a register file of sets; opcodes: 0 new r[a] from keys[b0..b1) (b = index into
 * spans), 1 r[a] = r[b].copy(), 2 r[a] = r[b] & r[c] (c = a>>8, a &= 255), 3 r[a] = r[b] - r[c], 4 r[a].discard(b),
 * 5 out <- r[a].pop(), 6 out <- list(r[a]).  Outputs are appended to out (returns the count). */
int64_t oracle_pyset_program(const int32_t *ops, int64_t nops, const int32_t *keys, const int32_t *spans,
                             int32_t *out, int64_t out_cap)
{
    PSet *r[64] = {0};
    int64_t no = 0;
    for (int64_t k = 0; k < nops; k++) {
        const int op = ops[3 * k], a = ops[3 * k + 1] & 255, c = ops[3 * k + 1] >> 8, b = ops[3 * k + 2];
        if (op == 0) { ps_free(r[a]); r[a] = ps_new(); for (int i = spans[2 * b]; i < spans[2 * b + 1]; i++) ps_add(r[a], keys[i]); }
        else if (op == 1) { PSet *t = ps_copy(r[b]); ps_free(r[a]); r[a] = t; }
        else if (op == 2) { PSet *t = ps_and(r[b], r[c]); ps_free(r[a]); r[a] = t; }
        else if (op == 3) { PSet *t = ps_sub(r[b], r[c]); ps_free(r[a]); r[a] = t; }
        else if (op == 4) ps_discard(r[a], b);
        else if (op == 5) { if (no < out_cap) out[no] = r[a]->used ? ps_pop(r[a]) : -1; no++; }
        else if (op == 6) { for (int i = 0; i <= r[a]->mask; i++) if (r[a]->tab[i] >= 0) { if (no < out_cap) out[no] = r[a]->tab[i]; no++; } if (no < out_cap) out[no] = -1; no++; }
    }
    for (int i = 0; i < 64; i++) ps_free(r[i]);
    return no;
}
