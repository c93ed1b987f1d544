"""one-off assurance run: the committed max_clique_kernel against the oracle's networkx-order clique on many random
consistency graphs (scan-pair-like, K = 3..330) - prints the mismatches (none expected) and the count"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from radarslampy_amd import _ffi
ctx = _ffi.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
bad = 0; t0 = time.time(); ties = 0
for t in range(N):
    K = int(rng.integers(3, 330))
    p = rng.uniform(100, 1900, size=(K, 2)).astype(np.float32)
    n = (p + rng.normal(0, rng.choice([0.8, 1.6, 2.4, 3.5]), size=(K, 2))).astype(np.float32)
    movers = rng.permutation(K)[:int(K * rng.uniform(0.0, 0.5))]
    n[movers] += rng.normal(0, rng.choice([4, 12, 40]), size=(len(movers), 2)).astype(np.float32)
    mask, n_in, flags, adj = ctx.reject_outliers(p, n, oracle.DIST_THRESHOLD_PX, want_adj=True)
    size, omask, st = oracle.max_clique_nx(adj)
    if not (flags & 1) or n_in != size or not np.array_equal(mask, omask):
        bad += 1
        print("MISMATCH case", t, "K", K, "gpu", n_in, flags, "oracle", size, flush=True)
    ties += not np.array_equal(omask, oracle.max_clique_lex(adj)[1])
    if time.time() - t0 > 500: N = t + 1; break
print("cases", N, "mismatches", bad, "differ from the lexicographic rule", ties)
