"""The reference's maximum-clique TIE-BREAK (outlierRejection.py:63-75: the first strictly-largest clique in
networkx.find_cliques order) as restated in oracle/c/clique.c: CPython's set (probing, growth, copy, &, -, pop, dummies)
and networkx's iterative Bron-Kerbosch, checked against the LIVE interpreter / networkx of this image and against the
masks the reference itself produced on its real fixtures (tests/golden/outliers.npz)."""
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _adj_words(A):
    K = A.shape[0]
    nw = max(1, (K + 63) // 64)
    bits = np.zeros((K, nw * 64), np.uint8)
    bits[:, :K] = A & ~np.eye(K, dtype=bool)
    return np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(K, nw)


def _nx_first_largest(A):
    """rejectOutliers' construct, verbatim in behaviour: nx.Graph(int8 matrix with a unit diagonal) -> find_cliques -> first
    strictly-largest clique"""
    nx = pytest.importorskip("networkx")
    G = nx.Graph((A | np.eye(A.shape[0], dtype=bool)).astype(np.int8))
    best = []
    for c in nx.find_cliques(G):
        if len(c) > len(best):
            best = c
    mask = np.zeros(A.shape[0], bool)
    mask[np.array(best, int)] = True
    return mask


def _random_graphs(rng, n_graphs, kmax):
    for t in range(n_graphs):
        K = int(rng.integers(1, kmax))
        kind = t % 4
        if kind == 0:                                            # sparse
            A = rng.random((K, K)) < rng.choice([0.05, 0.15, 0.3])
        elif kind == 1:                                          # dense, many ties
            A = rng.random((K, K)) < rng.choice([0.7, 0.85, 0.93])
        elif kind == 2:                                          # a planted clique + exchangeable pairs: tie-heavy
            A = rng.random((K, K)) < 0.3
            core = rng.permutation(K)[:max(2, K // 2)]
            A[np.ix_(core, core)] = True
            for a, b in zip(core[::3], core[1::3]):
                A[a, b] = A[b, a] = False
        else:                                                    # like a scan pair: one big clique, sparse conflicts
            A = np.ones((K, K), bool)
            for _ in range(int(K * rng.uniform(0.2, 0.6))):
                a, b = rng.integers(0, K, 2)
                A[a, b] = A[b, a] = False
        A = np.triu(A, 1)
        yield (A | A.T)


def scan_pair_like(rng, K):
    """static features form one big clique, movers / mistracks are adjacent to random parts of it, and a handful of
    borderline pairs of static features fail the threshold - the structure that makes ties (the number of maximal cliques
    stays in the thousands, so that networkx can enumerate it)"""
    A = np.ones((K, K), bool)
    out = rng.permutation(K)[:int(K * rng.uniform(0.1, 0.35))]
    for o in out:
        row = rng.random(K) < rng.uniform(0.2, 0.7)
        A[o, :] = row
        A[:, o] = row
    inl = np.setdiff1d(np.arange(K), out)
    for _ in range(int(rng.integers(2, 9))):
        a, b = rng.choice(inl, 2, replace=False)
        A[a, b] = A[b, a] = False
    A &= ~np.eye(K, dtype=bool)
    return A & A.T


def test_set_restatement_matches_the_live_interpreter():
    """programs of set operations (build by insertion, copy, &, -, discard, pop, iteration) on ints 0..1023: the C tables
    iterate in CPython's own order"""
    rng = np.random.default_rng(7)
    for trial in range(300):
        n = int(rng.choice([3, 6, 9, 20, 33, 70, 130, 200, 400, 600, 1024]))
        live, ops, keys, spans, want = {}, [], [], [], []

        def new(a):
            ks = rng.permutation(n)[:int(rng.integers(0, n + 1))] if rng.random() < 0.5 else np.sort(rng.permutation(n)[:int(rng.integers(0, n + 1))])
            spans.append((len(keys), len(keys) + len(ks)))
            keys.extend(int(k) for k in ks)
            ops.append((0, a, len(spans) - 1))
            s = set()
            for k in ks.tolist():
                s.add(k)
            live[a] = s
        for a in range(4):
            new(a)
        for step in range(40):
            op = int(rng.integers(1, 7))
            a, b, c = (int(v) for v in rng.integers(0, 4, 3))
            if op == 1:
                live[a] = live[b].copy(); ops.append((1, a, b))
            elif op == 2 and b != c:
                live[a] = live[b] & live[c]; ops.append((2, a | (c << 8), b))
            elif op == 3 and b != c:
                live[a] = live[b] - live[c]; ops.append((3, a | (c << 8), b))
            elif op == 4:
                for k in rng.integers(0, n, 5).tolist():
                    live[a].discard(k); ops.append((4, a, k))
            elif op == 5:
                want.append(live[a].pop() if live[a] else -1); ops.append((5, a, 0))
            elif op == 6:
                want.extend(list(live[a]) + [-1]); ops.append((6, a, 0))
        for a in range(4):
            want.extend(list(live[a]) + [-1]); ops.append((6, a, 0))
        got = oracle.pyset_program(ops, keys, spans if spans else [(0, 0)])
        assert got.tolist() == want, trial


def test_first_largest_clique_matches_live_networkx():
    """plain enumeration (prune=0) and the bounded walk (prune=1) both return networkx's first strictly-largest clique"""
    rng = np.random.default_rng(11)
    n = 0
    for A in _random_graphs(rng, 240, 34):
        want = _nx_first_largest(A)
        adj = _adj_words(A)
        s0, m0, _ = oracle.max_clique_nx(adj, prune=False)
        s1, m1, st = oracle.max_clique_nx(adj, prune=True)
        assert np.array_equal(m0, want) and s0 == want.sum(), n
        assert np.array_equal(m1, want) and s1 == want.sum(), n
        n += 1


def test_bounded_walk_on_scan_pair_like_graphs_matches_live_networkx():
    """larger tie-heavy graphs of the scan-pair kind (one big clique, sparse conflicts): only the bounded walk is feasible in
    C; networkx still enumerates them in seconds"""
    rng = np.random.default_rng(5)
    ties = 0
    for t in range(40):
        A = scan_pair_like(rng, int(rng.integers(40, 140)))
        want = _nx_first_largest(A)
        size, mask, st = oracle.max_clique_nx(_adj_words(A))
        assert size == want.sum() and np.array_equal(mask, want), t
        lsize, lmask, _ = oracle.max_clique_lex(_adj_words(A))
        assert lsize == size
        ties += not np.array_equal(lmask, mask)
    assert ties >= 10, ties                                     # the tie-break matters on most of them


def test_the_reference_masks_on_its_own_fixtures():
    """tests/golden/outliers.npz holds the masks the reference's rejectOutliers returned (make_goldens.py imported it):
    its outlier_test.npz in both directions, its real 95-pair fixture (16 maximum cliques of 67) and four unique-clique
    sets - the restatement returns every one of them, ties included"""
    g = np.load(os.path.join(GOLD, "outliers.npz"))
    tags = sorted({k[:-5] for k in g.files if k.endswith("_mask")})
    assert {"npz139", "npz139b", "real95"} <= set(tags)
    for tag in tags:
        prev, new = g[f"{tag}_prev"], g[f"{tag}_new"]
        _, _, mask = oracle.rejectOutliers(prev, new)
        assert np.array_equal(mask, g[f"{tag}_mask"]), tag
