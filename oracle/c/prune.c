/* ORACLE — TEST INFRASTRUCTURE ONLY (see peaks.c header).
 *
 * CPU restatement of the ORDER-DEPENDENT bookkeeping behind the reference's feature detection
 * (reference getFeatures.py:22-53,66-72), i.e. of what its pinned third-party stack does:
 *
 *  (1) skimage.feature.blob._prune_blobs (scikit-image 0.19.2, requirements.txt:5):
 *          tree  = scipy.spatial.cKDTree(blobs[:, :2])            (leafsize 16, balanced, compact)
 *          pairs = np.array(list(tree.query_pairs(distance)))      <- a Python SET of (i, j) tuples
 *          for (i, j) in pairs: if overlap(b_i, b_j) > 0.5: zero the sigma of the smaller (ties: b_i)
 *      The outcome depends on the order in which the pairs are visited (chains of overlapping blobs),
 *      i.e. on (a) the order in which cKDTree.query_pairs emits the pairs (they are inserted into the set
 *      in that order) and (b) CPython's set iteration order (hash of a 2-tuple of ints, open addressing
 *      with 9 linear probes + perturbation, growth x4).  Both are deterministic and are restated here:
 *          oracle_ckdtree_pairs   scipy ckdtree build (median split by std::nth_element with a plain
 *                                 value comparator = libstdc++ introselect, then the Hoare-style split
 *                                 pass) + query_pairs' dual-tree traversal with its
 *                                 RectRectDistanceTracker (p = 2)
 *          oracle_pyset_order     CPython >= 3.8 tuple hash (xxHash variant) + setobject.c insertion /
 *                                 resize / iteration
 *      tests/test_oracle_reference_dump.py checks both against the LIVE scipy / CPython of the test
 *      machine on random inputs (index permutation, pair emission order and set order: exact), and
 *      against the blob circles the reference drew into img/blob/tiny/*.jpg (fixture tiny_track.npz).
 *
 *  (2) getFeatures.adaptiveNMS (getFeatures.py:66-72): np.argsort(blobs[:, 2]) with NumPy's DEFAULT
 *      (unstable) sort.  Only two sigma values occur (5.005 and 10), so the order of the ~400 ties
 *      decides which blobs SSC keeps.  The reference pins numpy==1.22.3 (requirements.txt:3), whose
 *      argsort for float64 is npy_aquicksort (introsort: median of 3, insertion sort below 17
 *      elements, heapsort beyond depth 2*floor(log2 n)); restated in oracle_aquicksort_f64.  With it the
 *      oracle reproduces 2170 of the 2192 ANMS selections visible (green circles) in the reference's own
 *      11 rendered frames, five frames without a single difference; NumPy >= 2 (AVX-512 / highway
 *      sorts) orders the ties differently and reproduces only ~70 %.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------ libstdc++ std::nth_element */
typedef struct { const double *data; int m, d; } KeyCmp;
static inline int lt_(const KeyCmp *c, int64_t a, int64_t b) { return c->data[a * c->m + c->d] < c->data[b * c->m + c->d]; }

static void adjust_heap(int64_t *f, int64_t hole, int64_t len, int64_t value, const KeyCmp *c)
{
    const int64_t top = hole;
    int64_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (lt_(c, f[child], f[child - 1])) child--;
        f[hole] = f[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        f[hole] = f[child - 1];
        hole = child - 1;
    }
    int64_t parent = (hole - 1) / 2;
    while (hole > top && lt_(c, f[parent], value)) {
        f[hole] = f[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    f[hole] = value;
}

static void heap_select(int64_t *first, int64_t *middle, int64_t *last, const KeyCmp *c)
{
    const int64_t len = middle - first;
    if (len >= 2)
        for (int64_t parent = (len - 2) / 2;; parent--) {
            adjust_heap(first, parent, len, first[parent], c);
            if (parent == 0) break;
        }
    for (int64_t *i = middle; i < last; i++)
        if (lt_(c, *i, *first)) {
            int64_t v = *i;
            *i = *first;
            adjust_heap(first, 0, len, v, c);
        }
}

static void nth_element_(int64_t *first, int64_t *nth, int64_t *last, const KeyCmp *c)
{
    if (first == last || nth == last) return;
    int64_t n = last - first, lg = 0;
    while (n > 1) { n >>= 1; lg++; }
    int64_t depth = 2 * lg;
    while (last - first > 3) {
        if (depth == 0) {
            heap_select(first, nth + 1, last, c);
            int64_t t = *first; *first = *nth; *nth = t;
            return;
        }
        depth--;
        int64_t *mid = first + (last - first) / 2;
        int64_t *a = first + 1, *b = mid, *cc = last - 1, *s;          /* __move_median_to_first */
        if (lt_(c, *a, *b)) s = lt_(c, *b, *cc) ? b : (lt_(c, *a, *cc) ? cc : a);
        else s = lt_(c, *a, *cc) ? a : (lt_(c, *b, *cc) ? cc : b);
        { int64_t t = *first; *first = *s; *s = t; }
        int64_t *f = first + 1, *l = last;                              /* __unguarded_partition */
        const int64_t piv = *first;
        for (;;) {
            while (lt_(c, *f, piv)) f++;
            l--;
            while (lt_(c, piv, *l)) l--;
            if (!(f < l)) break;
            int64_t t = *f; *f = *l; *l = t;
            f++;
        }
        if (f <= nth) first = f; else last = f;
    }
    for (int64_t *i = first + 1; i < last; i++) {                       /* __insertion_sort */
        int64_t v = *i;
        if (lt_(c, v, *first)) { memmove(first + 1, first, (size_t)(i - first) * sizeof(int64_t)); *first = v; }
        else { int64_t *j = i; while (lt_(c, v, j[-1])) { *j = j[-1]; j--; } *j = v; }
    }
}

/* ------------------------------------------------------------------ scipy cKDTree (2-D) */
typedef struct { int split_dim; double split; int64_t start, end; int less, greater; } KNode;
typedef struct { KNode *nodes; int n, cap; const double *data; int64_t *idx; int leafsize; } KTree;

static int knew(KTree *t)
{
    if (t->n == t->cap) { t->cap = t->cap ? 2 * t->cap : 64; t->nodes = (KNode *)realloc(t->nodes, sizeof(KNode) * (size_t)t->cap); }
    return t->n++;
}

static int kbuild(KTree *t, int64_t start, int64_t end)
{
    const int me = knew(t);
    KNode nd; nd.start = start; nd.end = end; nd.less = nd.greater = -1; nd.split_dim = -1; nd.split = 0;
    if (end - start > t->leafsize) {
        double mx[2], mn[2];                                            /* compact_nodes: bounds of the node's points */
        for (int k = 0; k < 2; k++) mx[k] = mn[k] = t->data[t->idx[start] * 2 + k];
        for (int64_t j = start + 1; j < end; j++)
            for (int k = 0; k < 2; k++) {
                double v = t->data[t->idx[j] * 2 + k];
                mx[k] = mx[k] > v ? mx[k] : v;
                mn[k] = mn[k] < v ? mn[k] : v;
            }
        int d = 0; double size = 0;
        for (int k = 0; k < 2; k++) if (mx[k] - mn[k] > size) { d = k; size = mx[k] - mn[k]; }
        if (mx[d] != mn[d]) {
            KeyCmp c = {t->data, 2, d};
            int64_t i = (end - start) / 2;
            nth_element_(t->idx + start, t->idx + start + i, t->idx + end, &c);
            double split = t->data[t->idx[start + i] * 2 + d];
            int64_t p = start, q = end - 1;
            while (p <= q) {
                if (t->data[t->idx[p] * 2 + d] < split) p++;
                else if (t->data[t->idx[q] * 2 + d] >= split) q--;
                else { int64_t s = t->idx[p]; t->idx[p] = t->idx[q]; t->idx[q] = s; p++; q--; }
            }
            if (p == start) {                                           /* slide midpoint: no point below the split */
                int64_t j = start; split = t->data[t->idx[j] * 2 + d];
                for (int64_t k = start + 1; k < end; k++) if (t->data[t->idx[k] * 2 + d] < split) { j = k; split = t->data[t->idx[j] * 2 + d]; }
                int64_t s = t->idx[start]; t->idx[start] = t->idx[j]; t->idx[j] = s;
                p = start + 1;
            } else if (p == end) {
                int64_t j = end - 1; split = t->data[t->idx[j] * 2 + d];
                for (int64_t k = start; k < end - 1; k++) if (t->data[t->idx[k] * 2 + d] > split) { j = k; split = t->data[t->idx[j] * 2 + d]; }
                int64_t s = t->idx[end - 1]; t->idx[end - 1] = t->idx[j]; t->idx[j] = s;
                p = end - 1;
            }
            nd.split_dim = d; nd.split = split;
            t->nodes[me] = nd;
            int l = kbuild(t, start, p);
            int g = kbuild(t, p, end);
            nd.less = l; nd.greater = g;
        }
    }
    t->nodes[me] = nd;
    return me;
}

/* RectRectDistanceTracker<MinkowskiDistP2>: squared min / max distance between the two current boxes */
typedef struct { int which, dim; double mind, maxd, lo, hi; } TItem;
typedef struct { double r1[2][2], r2[2][2]; double mind, maxd, ub, limit; TItem st[256]; int sp; } Tracker;

static void tdim(const Tracker *t, int k, double *mn, double *mx)
{
    const double a0 = t->r1[0][k], a1 = t->r1[1][k], b0 = t->r2[0][k], b1 = t->r2[1][k];
    double lo = a0 - b1 > b0 - a1 ? a0 - b1 : b0 - a1;
    if (lo < 0) lo = 0;
    double hi = a1 - b0 > b1 - a0 ? a1 - b0 : b1 - a0;
    *mn = lo * lo; *mx = hi * hi;
}
static void tfull(Tracker *t)
{
    double mn = 0, mx = 0, a, b;
    for (int k = 0; k < 2; k++) { tdim(t, k, &a, &b); mn += a; mx += b; }
    t->mind = mn; t->maxd = mx;
}
static void tpush(Tracker *t, int which, int less, int dim, double split)
{
    double (*rect)[2] = which == 1 ? t->r1 : t->r2;
    TItem *it = &t->st[t->sp++];
    it->which = which; it->dim = dim; it->mind = t->mind; it->maxd = t->maxd; it->lo = rect[0][dim]; it->hi = rect[1][dim];
    double min1, max1, min2, max2;
    tdim(t, dim, &min1, &max1);
    if (less) rect[1][dim] = split; else rect[0][dim] = split;
    tdim(t, dim, &min2, &max2);
    const double L = t->limit;
    if (t->mind < L || t->maxd < L || (min1 != 0 && min1 < L) || max1 < L || (min2 != 0 && min2 < L) || max2 < L) tfull(t);
    else { t->mind += (min2 - min1); t->maxd += (max2 - max1); }
}
static void tpop(Tracker *t)
{
    TItem *it = &t->st[--t->sp];
    double (*rect)[2] = it->which == 1 ? t->r1 : t->r2;
    t->mind = it->mind; t->maxd = it->maxd; rect[0][it->dim] = it->lo; rect[1][it->dim] = it->hi;
}

typedef struct { int64_t *p; int64_t n, cap; } Pairs;
static void padd(Pairs *o, int64_t i, int64_t j)
{
    if (o->n == o->cap) { o->cap = o->cap ? 2 * o->cap : 1024; o->p = (int64_t *)realloc(o->p, sizeof(int64_t) * 2 * (size_t)o->cap); }
    if (i > j) { int64_t s = i; i = j; j = s; }
    o->p[2 * o->n] = i; o->p[2 * o->n + 1] = j; o->n++;
}

static void knocheck(const KTree *t, Pairs *o, int a, int b)
{
    const KNode *n1 = &t->nodes[a], *n2 = &t->nodes[b];
    if (n1->split_dim == -1) {
        if (n2->split_dim == -1) {
            for (int64_t i = n1->start; i < n1->end; i++)
                for (int64_t j = (a == b ? i + 1 : n2->start); j < n2->end; j++) padd(o, t->idx[i], t->idx[j]);
        } else { knocheck(t, o, a, n2->less); knocheck(t, o, a, n2->greater); }
    } else if (a == b) {
        knocheck(t, o, n1->less, n2->less); knocheck(t, o, n1->less, n2->greater); knocheck(t, o, n1->greater, n2->greater);
    } else { knocheck(t, o, n1->less, b); knocheck(t, o, n1->greater, b); }
}

static void kcheck(const KTree *t, Tracker *tr, Pairs *o, int a, int b)
{
    const KNode *n1 = &t->nodes[a], *n2 = &t->nodes[b];
    if (tr->mind > tr->ub) return;
    if (tr->maxd < tr->ub) { knocheck(t, o, a, b); return; }
    if (n1->split_dim == -1) {
        if (n2->split_dim == -1) {
            for (int64_t i = n1->start; i < n1->end; i++) {
                const double *pi = t->data + t->idx[i] * 2;
                for (int64_t j = (a == b ? i + 1 : n2->start); j < n2->end; j++) {
                    const double *pj = t->data + t->idx[j] * 2;
                    double d = (pi[0] - pj[0]) * (pi[0] - pj[0]);
                    d += (pi[1] - pj[1]) * (pi[1] - pj[1]);
                    if (d <= tr->ub) padd(o, t->idx[i], t->idx[j]);
                }
            }
        } else {
            tpush(tr, 2, 1, n2->split_dim, n2->split); kcheck(t, tr, o, a, n2->less); tpop(tr);
            tpush(tr, 2, 0, n2->split_dim, n2->split); kcheck(t, tr, o, a, n2->greater); tpop(tr);
        }
    } else if (n2->split_dim == -1) {
        tpush(tr, 1, 1, n1->split_dim, n1->split); kcheck(t, tr, o, n1->less, b); tpop(tr);
        tpush(tr, 1, 0, n1->split_dim, n1->split); kcheck(t, tr, o, n1->greater, b); tpop(tr);
    } else {
        tpush(tr, 1, 1, n1->split_dim, n1->split);
        tpush(tr, 2, 1, n2->split_dim, n2->split); kcheck(t, tr, o, n1->less, n2->less); tpop(tr);
        tpush(tr, 2, 0, n2->split_dim, n2->split); kcheck(t, tr, o, n1->less, n2->greater); tpop(tr);
        tpop(tr);
        tpush(tr, 1, 0, n1->split_dim, n1->split);
        if (a != b) { tpush(tr, 2, 1, n2->split_dim, n2->split); kcheck(t, tr, o, n1->greater, n2->less); tpop(tr); }
        tpush(tr, 2, 0, n2->split_dim, n2->split); kcheck(t, tr, o, n1->greater, n2->greater); tpop(tr);
        tpop(tr);
    }
}

/* pts (n,2) f64 -> pairs in the order scipy.spatial.cKDTree(pts).query_pairs(r) emits them.
 * *pairs_out is malloc'ed ((count,2) int64); idx_out (n, optional) receives cKDTree.indices. */
int64_t oracle_ckdtree_pairs(const double *pts, int64_t n, double r, int64_t **pairs_out, int64_t *idx_out)
{
    *pairs_out = NULL;
    if (n <= 0) return 0;
    KTree t; memset(&t, 0, sizeof(t));
    t.data = pts; t.leafsize = 16;
    t.idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; i++) t.idx[i] = i;
    kbuild(&t, 0, n);
    Tracker tr; memset(&tr, 0, sizeof(tr));
    for (int k = 0; k < 2; k++) {
        double mn = pts[k], mx = pts[k];
        for (int64_t i = 1; i < n; i++) { double v = pts[2 * i + k]; mn = mn < v ? mn : v; mx = mx > v ? mx : v; }
        tr.r1[0][k] = tr.r2[0][k] = mn; tr.r1[1][k] = tr.r2[1][k] = mx;
    }
    tr.ub = r * r;
    tfull(&tr);
    tr.limit = tr.maxd;                       /* inaccurate_distance_limit */
    Pairs o = {NULL, 0, 0};
    kcheck(&t, &tr, &o, 0, 0);
    if (idx_out) memcpy(idx_out, t.idx, sizeof(int64_t) * (size_t)n);
    free(t.idx); free(t.nodes);
    *pairs_out = o.p;
    return o.n;
}

void oracle_free(void *p) { free(p); }

/* ------------------------------------------------------------------ CPython set of (i, j) tuples */
static int64_t tuple_hash2(uint64_t i, uint64_t j)
{
    const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
    uint64_t acc = P5, lane[2] = {i, j};                 /* hash(int) = the int for 0 <= v < 2^61 - 1 */
    for (int k = 0; k < 2; k++) {
        acc += lane[k] * P2;
        acc = (acc << 31) | (acc >> 33);
        acc *= P1;
    }
    acc += 2ULL ^ (P5 ^ 3527539ULL);
    if (acc == (uint64_t)-1) return 1546275796;
    return (int64_t)acc;
}

typedef struct { int64_t key, hash; } SEnt;             /* key = index of the pair, -1 = empty */

static void sinsert_clean(SEnt *tab, uint64_t mask, int64_t key, int64_t hash)
{
    uint64_t perturb = (uint64_t)hash, i = (uint64_t)hash & mask;
    for (;;) {
        int probes = (i + 9 <= mask) ? 9 : 0;
        SEnt *e = &tab[i];
        do { if (e->key < 0) { e->key = key; e->hash = hash; return; } e++; } while (probes--);
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & mask;
    }
}

/* order_out (n) = indices into `pairs` in the iteration order of the set built by adding them in order.
 * returns the number of DISTINCT pairs (query_pairs never emits duplicates, so = n). */
int64_t oracle_pyset_order(const int64_t *pairs, int64_t n, int64_t *order_out)
{
    uint64_t mask = 7;
    int64_t fill = 0;
    SEnt *tab = (SEnt *)malloc(sizeof(SEnt) * 8);
    for (int k = 0; k < 8; k++) tab[k].key = -1;
    for (int64_t p = 0; p < n; p++) {
        const int64_t h = tuple_hash2((uint64_t)pairs[2 * p], (uint64_t)pairs[2 * p + 1]);
        uint64_t perturb = (uint64_t)h, i = (uint64_t)h & mask;
        int done = 0;
        while (!done) {
            int probes = (i + 9 <= mask) ? 9 : 0;
            SEnt *e = &tab[i];
            do {
                if (e->key < 0) { e->key = p; e->hash = h; fill++; done = 1; break; }
                if (e->hash == h && pairs[2 * e->key] == pairs[2 * p] && pairs[2 * e->key + 1] == pairs[2 * p + 1]) { done = 1; break; }
                e++;
            } while (probes--);
            if (done) break;
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
        if ((uint64_t)fill * 5 >= mask * 3) {
            const int64_t minused = fill > 50000 ? fill * 2 : fill * 4;
            uint64_t newsize = 8;
            while ((int64_t)newsize <= minused) newsize <<= 1;
            SEnt *nt = (SEnt *)malloc(sizeof(SEnt) * newsize);
            for (uint64_t k = 0; k < newsize; k++) nt[k].key = -1;
            for (uint64_t k = 0; k <= mask; k++) if (tab[k].key >= 0) sinsert_clean(nt, newsize - 1, tab[k].key, tab[k].hash);
            free(tab); tab = nt; mask = newsize - 1;
        }
    }
    int64_t m = 0;
    for (uint64_t k = 0; k <= mask; k++) if (tab[k].key >= 0) order_out[m++] = tab[k].key;
    free(tab);
    return m;
}

/* ------------------------------------------------------------------ skimage _blob_overlap / _prune_blobs (2-D) */
static double blob_overlap(const double *b1, const double *b2)
{
    const double root2 = sqrt(2.0);
    double r1, r2, ms;
    if (b1[2] == 0 && b2[2] == 0) return 0.0;
    if (b1[2] > b2[2]) { ms = b1[2]; r1 = 1.0; r2 = b2[2] / b1[2]; }
    else { ms = b2[2]; r2 = 1.0; r1 = b1[2] / b2[2]; }
    const double p0 = b1[0] / (ms * root2), p1 = b1[1] / (ms * root2), q0 = b2[0] / (ms * root2), q1 = b2[1] / (ms * root2);
    const double d = sqrt((q0 - p0) * (q0 - p0) + (q1 - p1) * (q1 - p1));
    if (d > r1 + r2) return 0.0;
    if (d <= fabs(r1 - r2)) return 1.0;
    double ratio1 = (d * d + r1 * r1 - r2 * r2) / (2 * d * r1);
    ratio1 = ratio1 < -1 ? -1 : (ratio1 > 1 ? 1 : ratio1);
    const double acos1 = acos(ratio1);
    double ratio2 = (d * d + r2 * r2 - r1 * r1) / (2 * d * r2);
    ratio2 = ratio2 < -1 ? -1 : (ratio2 > 1 ? 1 : ratio2);
    const double acos2 = acos(ratio2);
    const double a = -d + r2 + r1, b = d - r2 + r1, c = d + r2 - r1, dd = d + r2 + r1;
    const double area = r1 * r1 * acos1 + r2 * r2 * acos2 - 0.5 * sqrt(fabs(a * b * c * dd));
    const double rmin = r1 < r2 ? r1 : r2;
    return area / (3.141592653589793 * (rmin * rmin));
}

/* blobs (n,3) f64 rows [row, col, sigma] in peak_local_max order (highest response first); sigmas of the
 * pruned blobs are zeroed IN PLACE exactly as _prune_blobs does.  returns the number of survivors. */
int64_t oracle_prune_blobs(double *blobs, int64_t n, double overlap)
{
    if (n <= 0) return 0;
    double smax = blobs[2];
    for (int64_t i = 1; i < n; i++) smax = smax > blobs[3 * i + 2] ? smax : blobs[3 * i + 2];
    const double distance = 2 * smax * sqrt(2.0);
    double *pts = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    for (int64_t i = 0; i < n; i++) { pts[2 * i] = blobs[3 * i]; pts[2 * i + 1] = blobs[3 * i + 1]; }
    int64_t *pairs = NULL;
    const int64_t np_ = oracle_ckdtree_pairs(pts, n, distance, &pairs, NULL);
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(np_ > 0 ? np_ : 1));
    const int64_t m = oracle_pyset_order(pairs, np_, order);
    for (int64_t k = 0; k < m; k++) {
        double *b1 = blobs + 3 * pairs[2 * order[k]], *b2 = blobs + 3 * pairs[2 * order[k] + 1];
        if (blob_overlap(b1, b2) > overlap) {
            if (b1[2] > b2[2]) b2[2] = 0; else b1[2] = 0;
        }
    }
    free(order); free(pairs); free(pts);
    int64_t keep = 0;
    for (int64_t i = 0; i < n; i++) keep += blobs[3 * i + 2] > 0;
    return keep;
}

/* ------------------------------------------------------------------ numpy 1.22 npy_aquicksort (float64) */
static void aheapsort_f64(const double *v, int64_t *tosort, int64_t n)
{
    int64_t *a = tosort - 1, i, j, l, tmp;                /* 1-based like numpy */
    for (l = n >> 1; l > 0; --l) {
        tmp = a[l];
        for (i = l, j = l << 1; j <= n;) {
            if (j < n && v[a[j]] < v[a[j + 1]]) j += 1;
            if (v[tmp] < v[a[j]]) { a[i] = a[j]; i = j; j += j; } else break;
        }
        a[i] = tmp;
    }
    for (; n > 1;) {
        tmp = a[n]; a[n] = a[1]; n -= 1;
        for (i = 1, j = 2; j <= n;) {
            if (j < n && v[a[j]] < v[a[j + 1]]) j++;
            if (v[tmp] < v[a[j]]) { a[i] = a[j]; i = j; j += j; } else break;
        }
        a[i] = tmp;
    }
}

/* tosort (n) receives np.argsort(v) as numpy 1.22.3 computes it for a float64 vector without NaNs */
void oracle_aquicksort_f64(const double *v, int64_t num, int64_t *tosort)
{
    for (int64_t i = 0; i < num; i++) tosort[i] = i;
    if (num < 2) return;
    int64_t *pl = tosort, *pr = tosort + num - 1, *stack[128], **sptr = stack, *pm, *pi, *pj, *pk, vi, tmp;
    int depth[128], *psdepth = depth, cdepth = 0;
    for (int64_t k = num; k > 1; k >>= 1) cdepth++;
    cdepth *= 2;
    double vp;
    for (;;) {
        if (cdepth < 0) { aheapsort_f64(v, pl, pr - pl + 1); goto stack_pop; }
        while ((pr - pl) > 16) {
            pm = pl + ((pr - pl) >> 1);
            if (v[*pm] < v[*pl]) { tmp = *pm; *pm = *pl; *pl = tmp; }
            if (v[*pr] < v[*pm]) { tmp = *pr; *pr = *pm; *pm = tmp; }
            if (v[*pm] < v[*pl]) { tmp = *pm; *pm = *pl; *pl = tmp; }
            vp = v[*pm];
            pi = pl; pj = pr - 1;
            tmp = *pm; *pm = *pj; *pj = tmp;
            for (;;) {
                do ++pi; while (v[*pi] < vp);
                do --pj; while (vp < v[*pj]);
                if (pi >= pj) break;
                tmp = *pi; *pi = *pj; *pj = tmp;
            }
            pk = pr - 1;
            tmp = *pi; *pi = *pk; *pk = tmp;
            if (pi - pl < pr - pi) { *sptr++ = pi + 1; *sptr++ = pr; pr = pi - 1; }
            else { *sptr++ = pl; *sptr++ = pi - 1; pl = pi + 1; }
            *psdepth++ = --cdepth;
        }
        for (pi = pl + 1; pi <= pr; ++pi) {
            vi = *pi; vp = v[vi]; pj = pi; pk = pi - 1;
            while (pj > pl && vp < v[*pk]) *pj-- = *pk--;
            *pj = vi;
        }
    stack_pop:
        if (sptr == stack) break;
        pr = *(--sptr);
        pl = *(--sptr);
        cdepth = *(--psdepth);
    }
}
