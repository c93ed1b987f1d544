#!/usr/bin/env python3
"""digest of a 1 536-lane job in which every lane re-detects in every step (set_retrack(2)): the same digest must come out with
ROAM_DET_SIDE=0 and with ROAM_DET_SIDE=512 (determinants of a chunk on a second stream beside the next chunk's integral images)"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
ctx = _ffi.Context(0)
B, T = 1536, 4
seqs = [synth.make_sequence(200 + d, T, n_movers=30, distortion=True, scintillation=0.5) for d in range(3)]
eng = Engine(B, 3 * T, ctx=ctx, retrack_on_device=True)
for d in range(3):
    for t in range(T):
        eng.upload_scan(d * T + t, seqs[d][0][t])
for b in range(B):
    eng.init_lane(b, (b % 3) * T, seqs[b % 3][2][:60 + (b % 5) * 30], seqs[b % 3][1][0])
h = 0
for t in range(1, T):
    eng.set_retrack(2)
    eng.step([(b % 3) * T + t for b in range(B)])
    res = eng.results()
    h = zlib.crc32(np.array([r["pose"] for r in res]).tobytes(), h)
    h = zlib.crc32(np.array([[r["n_tracked"], r["n_good"], r["n_inliers"], r["n_after_retrack"]] for r in res]).tobytes(), h)
    for b in (0, 1, 2, 511, 512, 1023, 1024, 1535):
        h = zlib.crc32(eng.lane_features(b).tobytes(), h)
print("digest %08x" % h, "env", os.environ.get("ROAM_DET_SIDE", "-"))
eng.close(); ctx.close()
