"""Determinism / pipeline-safety check on the GPU: the same 3-sequence, 384-lane job is run several times with the host
reading results after every step (the two pipeline stages never overlap) and several times with all steps enqueued
back to back (front end of step N+1 overlaps the back end of step N).  Every run must leave bit-identical final
poses, counts, features, peak lists and pyramids."""
import sys, zlib
sys.path.insert(0, '.')
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
ctx = _ffi.Context(0)
B, T = 384, 6
seqs = [synth.make_sequence(100 + d, T, n_movers=8, distortion=True) for d in range(3)]


def run(sync):
    eng = Engine(B, 3 * T, ctx=ctx)
    for d in range(3):
        for t in range(T):
            eng.upload_scan(d * T + t, seqs[d][0][t])
    for b in range(B):
        eng.init_lane(b, (b % 3) * T, seqs[b % 3][2][:150 + (b % 7) * 40], seqs[b % 3][1][0])
    for t in range(1, T):
        eng.step([(b % 3) * T + t for b in range(B)])
        if sync:
            eng.results()
    res = eng.results()
    h = zlib.crc32(np.array([r["pose"] for r in res]).tobytes())
    h = zlib.crc32(np.array([[r["n_tracked"], r["n_good"], r["n_inliers"], r["n_peaks"], r["lm_nfev"]] for r in res]).tobytes(), h)
    for b in (0, 1, 2, 50, 95, 200, 383):
        h = zlib.crc32(eng.lane_peaks(b).tobytes(), h)
        h = zlib.crc32(eng.lane_features(b).tobytes(), h)
        for lvl in range(4):
            h = zlib.crc32(eng.lane_image(b, lvl).tobytes(), h)
    eng.close()
    return h


hs = [run(True) for _ in range(3)] + [run(False) for _ in range(5)]
print("checksums", hs, "DETERMINISTIC" if len(set(hs)) == 1 else "MISMATCH")
sys.exit(0 if len(set(hs)) == 1 else 1)
