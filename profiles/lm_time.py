import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from radarslampy_amd import _ffi
g = np.load("tests/golden/mds.npz")
ctx = _ffi.Context(0)
sigma5 = np.array([4, 4, 1, 1, (5 * np.pi / 180) ** 2], np.float64)
for tag in ["n8", "n60", "n150", "n250"]:
    for rep in range(3):
        out = ctx.mds_solve(g[f"{tag}_T0"], g[f"{tag}_p_w"], g[f"{tag}_p_jt"], g[f"{tag}_Tinit"], sigma5, want_debug=True)
    print(tag, "N", len(g[f"{tag}_p_w"]), "nfev", out[1], "info", out[2], "ticks(setup,jac,qrfac,qtf,lmpar,resid)", [int(v) for v in out[3]], "sum", int(sum(out[3])))
ctx.close()
