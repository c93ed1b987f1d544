import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
import oracle
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
ctx = _ffi.Context(0)
o = np.load(os.path.join(G, "outliers.npz")); h = np.load(os.path.join(G, "clique_hard240.npz"))
sets = {"npz139": (o["npz139_prev"], o["npz139_new"]), "real95": (o["real95_prev"], o["real95_new"]), "u256": (o["u256_prev"], o["u256_new"]), "synthetic240": (h["go"], h["gn"])}
for name, (p, n) in sets.items():
    mask, n_in, flags, _ = ctx.reject_outliers(p, n, 0.5 / 0.0864)
    print(name, "K", len(p), "omega", n_in & 0xffff, "phase1 nodes", n_in >> 16, "walk queries", (flags >> 8) & 255, "walk nodes", flags >> 16)
# bench-like pairs: synthetic sequence with movers
recs, poses, feat = synth.make_sequence(5, 7, n_movers=120, distortion=True)
det = lambda c: oracle.getFeatures(c)[0]
cart0 = oracle.convertPolarImageToCartesian(recs[0][:, 11:11 + 2025].astype(np.float32) / np.float32(255.))
feat0 = oracle.append_dedupe(np.empty((0, 2)), det(cart0))
pipe = oracle.OdometryPipeline(recs[0], feat0, poses[0], detect=det)
real = oracle.rejectOutliers
def rej(prev, new):
    mask, n_in, flags, _ = ctx.reject_outliers(prev, new, 0.5 / 0.0864)
    print("  pair K", len(prev), "omega", n_in & 0xffff, "phase1 nodes", n_in >> 16, "walk queries", (flags >> 8) & 255, "walk nodes", flags >> 16)
    return real(prev, new)
oracle.rejectOutliers = rej
for t in range(1, 7):
    pipe.step(recs[t])
