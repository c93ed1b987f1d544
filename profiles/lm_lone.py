#!/usr/bin/env python3
"""ONE motion-distortion solve per launch (the stage API on the goldens of tests/golden/mds.npz): run under
rocprofv3 --kernel-trace --stats to read the latency of mds_lm_kernel alone.  usage: python profiles/lm_lone.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mds.npz"))
ctx = _ffi.Context(0)
sigma5 = np.array([4, 4, 1, 1, (5 * np.pi / 180) ** 2], np.float64)
for tag in ["n8", "n60", "n150", "n250"]:
    for rep in range(10):
        out = ctx.mds_solve(g[f"{tag}_T0"], g[f"{tag}_p_w"], g[f"{tag}_p_jt"], g[f"{tag}_Tinit"], sigma5)
    print(tag, "N", len(g[f"{tag}_p_w"]), "nfev", out[1], "info", out[2])
ctx.close()
