#!/usr/bin/env python3
"""the motion-distortion solve of a whole batch ALONE: synchronous steps of a B-lane engine, the LM stage's event time
(ROAM_LM_BLOCK=1 / ROAM_LM_WPE=n choose the kernel form).  usage: python profiles/lm_batch.py [lanes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = 5
recs, poses, feat = synth.make_sequence(5, T, n_movers=60, distortion=True)
ctx = _ffi.Context(0)
eng = Engine(B, T * B, ctx=ctx)
for t in range(T):
    eng.upload_scan(t, recs[t])
for b in range(1, B):
    for t in range(T):
        eng.copy_scan(b * T + t, t)
eng.synchronize()
for b in range(B):
    eng.init_lane(b, b * T, feat, poses[0])
out = []
for t in range(1, T):
    eng.step(np.arange(B, dtype=np.int32) * T + t)
    eng.synchronize()
    st = eng.stage_times()
    r = eng.results()
    out.append((round(st["mds_lm"], 3), int(np.median([x["n_inliers"] for x in r]))))
print("lanes", B, "mds_lm stage ms / median inliers per step:", out)
eng.close(); ctx.close()
