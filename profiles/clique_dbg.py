import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from radarslampy_amd import _ffi
ctx = _ffi.Context(0)
rng = np.random.default_rng(77)
for t in range(40):
    K = int(rng.integers(3, 40 + 8 * t))
    p = rng.uniform(100, 1900, size=(K, 2)).astype(np.float32)
    n = (p + rng.normal(0, rng.choice([0.8, 1.6, 2.4]), size=(K, 2))).astype(np.float32)
    movers = rng.permutation(K)[:int(K * rng.uniform(0.05, 0.4))]
    n[movers] += rng.normal(0, 12, size=(len(movers), 2)).astype(np.float32)
    adj = oracle.consistency_graph(p, n)
    size, omask, st = oracle.max_clique_nx(adj)
    print(t, K, size, st, flush=True)
    mask, n_in, flags, _ = ctx.reject_outliers(p, n, oracle.DIST_THRESHOLD_PX)
    print("   gpu", n_in, flags, np.array_equal(mask, omask), flush=True)
