"""one case of tests/test_gpu_properties.py::test_clique_tie_break_in_every_table_regime per process (a hanging kernel must not take the others down)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
rng = np.random.default_rng(2024)
cases = []
for K in (9, 33, 70, 129, 200, 300, 520):
    p = rng.uniform(60, 1960, size=(K, 2)).astype(np.float32)
    n = (p + rng.normal(0, 1.9, size=(K, 2))).astype(np.float32)
    mv = rng.permutation(K)[:max(1, K // 4)]
    n[mv] += rng.normal(0, 15, size=(len(mv), 2)).astype(np.float32)
    cases.append((p, n))
for K in (40, 150, 400):
    p = rng.uniform(60, 1960, size=(K, 2)).astype(np.float32)
    n = (p + rng.normal(0, 40, size=(K, 2))).astype(np.float32)
    for g in range(K // 12):
        idx = rng.permutation(K)[:6]
        n[idx] = p[idx] + rng.normal(0, 25, size=2).astype(np.float32) + rng.normal(0, 0.8, size=(6, 2)).astype(np.float32)
    cases.append((p, n))
if len(sys.argv) > 1:
    i = int(sys.argv[1])
    from radarslampy_amd import _ffi
    ctx = _ffi.Context(0)
    p, n = cases[i]
    adj = oracle.consistency_graph(p, n)
    size, omask, st = oracle.max_clique_nx(adj)
    print("case", i, "K", len(p), "omega", size, "oracle stats", st, flush=True)
    mask, n_in, flags, _ = ctx.reject_outliers(p, n, oracle.DIST_THRESHOLD_PX)
    print("   gpu", n_in, flags, np.array_equal(mask, omask), flush=True)
else:
    import subprocess
    for i in range(len(cases)):
        r = subprocess.run(["timeout", "25", sys.executable, __file__, str(i)], capture_output=True, text=True)
        print(r.stdout.strip(), "| rc", r.returncode, flush=True)
