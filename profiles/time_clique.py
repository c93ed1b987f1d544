#!/usr/bin/env python3
"""max_clique_kernel on the reference's own hard graphs (outlier_test.npz K=139: 4 maximum cliques; the 95-pair fixture of
archive/testTransformKLT2.py: 16 maximum cliques; reference cost 0.13-1.33 s in networkx) and on a 240-feature synthetic pair
right after a re-detection (maximum clique 155), each replicated over 4096 problems like an engine step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ctx = _ffi.Context(0)
o = np.load(os.path.join(G, "outliers.npz"))
h = np.load(os.path.join(G, "clique_hard240.npz"))
sets = {"npz139": (o["npz139_prev"], o["npz139_new"]), "real95": (o["real95_prev"], o["real95_new"]), "u256": (o["u256_prev"], o["u256_new"]),
        "synthetic240": (h["go"], h["gn"])}
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, (p, n) in sets.items():
    g, q, ni, pr = ctx.time_reject_outliers(p, n, 0.5 / 0.0864, copies=copies, reps=3)
    print(f"{name}: K={len(p)} copies={copies}: graph {g:.3f} ms, max clique {q:.3f} ms per launch ({q*1e3/copies:.2f} us per problem), clique size {ni}, proven {pr}")
