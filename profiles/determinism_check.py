import sys, zlib
sys.path.insert(0, '.')
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
ctx = _ffi.Context(0)
B, T = 96, 4
seqs = [synth.make_sequence(100 + d, T, n_movers=8, distortion=True) for d in range(3)]
def run():
    eng = Engine(B, 3 * T, ctx=ctx)
    for d in range(3):
        for t in range(T): eng.upload_scan(d * T + t, seqs[d][0][t])
    for b in range(B): eng.init_lane(b, (b % 3) * T, seqs[b % 3][2], seqs[b % 3][1][0])
    h = 0
    for t in range(1, T):
        eng.step([(b % 3) * T + t for b in range(B)])
        res = eng.results()
        for b in (0, 1, 2, 50, 95):
            h = zlib.crc32(eng.lane_peaks(b).tobytes(), h); h = zlib.crc32(eng.lane_features(b).tobytes(), h)
            h = zlib.crc32(eng.lane_image(b, 3).tobytes(), h)
        h = zlib.crc32(np.array([r["pose"] for r in res]).tobytes(), h)
        h = zlib.crc32(np.array([[r["n_inliers"], r["n_peaks"], r["lm_nfev"]] for r in res]).tobytes(), h)
    eng.close()
    return h
hs = [run() for _ in range(6)]
print("checksums", hs, "DETERMINISTIC" if len(set(hs)) == 1 else "MISMATCH")
