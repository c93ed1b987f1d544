#!/usr/bin/env python3
"""ONE outlier-rejection problem at a time (the single-sequence configs: the clique stage is the largest part of a steady pair):
latency of max_clique_kernel alone on the correspondence sets of real and bench-like scan pairs.  The sets are made by the oracle's
loop on (i) the reference's 11 real data/tiny scans and (ii) a synthetic bench sequence, saved once under gpurun_out/ and re-read.
usage: python profiles/clique_lone.py [make]      (ROAM_LIB = an A/B build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = os.path.join(ROOT, "tests", "golden", "clique_lone_sets.npz")
if len(sys.argv) > 1 and sys.argv[1] == "make":                       # CPU only
    import oracle
    from radarslampy_amd import synth
    out = {}
    det = lambda c: oracle.getFeatures(c)[0]                            # noqa: E731
    real = oracle.rejectOutliers
    tag = [""]
    def rej(prev, new):
        out[tag[0] + "_prev"] = prev.copy(); out[tag[0] + "_new"] = new.copy()
        return real(prev, new)
    oracle.rejectOutliers = rej
    pay = np.load(os.path.join(ROOT, "tests", "golden", "tiny_track.npz"))["payload"]
    cart0 = oracle.convertPolarImageToCartesian(pay[0].astype(np.float32) / np.float32(255.))
    pipe = oracle.OdometryPipeline(np.ascontiguousarray(pay[0]), oracle.append_dedupe(np.empty((0, 2)), det(cart0)), np.zeros(3), detect=det, payload_off=0, clip=pay.shape[2])
    for t in range(1, 11):
        tag[0] = "tiny%02d" % t
        pipe.step(np.ascontiguousarray(pay[t]))
    recs, poses, feat = synth.make_sequence(5, 7, n_movers=120, distortion=True)
    cart0 = oracle.convertPolarImageToCartesian(recs[0][:, 11:11 + 2025].astype(np.float32) / np.float32(255.))
    pipe = oracle.OdometryPipeline(recs[0], oracle.append_dedupe(np.empty((0, 2)), det(cart0)), poses[0], detect=det)
    for t in range(1, 7):
        tag[0] = "synth%02d" % t
        pipe.step(recs[t])
    np.savez_compressed(SETS, **out)
    print("saved", sorted(k for k in out if k.endswith("_prev")))
    sys.exit(0)
from radarslampy_amd import _ffi
ctx = _ffi.Context(0)
z = np.load(SETS)
tot = 0.0
for k in sorted(k[:-5] for k in z.files if k.endswith("_prev")):
    p, n = z[k + "_prev"], z[k + "_new"]
    g, q, ni, pr = ctx.time_reject_outliers(p, n, 0.5 / 0.0864, copies=1, reps=20)
    mask, n_in, flags, _ = ctx.reject_outliers(p, n, 0.5 / 0.0864)
    tot += q
    print("%-8s K %3d omega %3d: graph %6.1f us, clique %7.1f us%s" % (k, len(p), ni & 0xffff, g * 1e3, q * 1e3,
          ("  [phase1 nodes %d, walk queries %d, walk nodes %d]" % (n_in >> 16, (flags >> 8) & 255, flags >> 16)) if os.environ.get("STATS") else ""))
print("sum of clique latencies: %.1f us over the sets" % (tot * 1e3))
ctx.close()
