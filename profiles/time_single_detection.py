#!/usr/bin/env python3
"""latency of ONE device-side detection (init_lane_detect on a 1-lane engine) and of small chunks.
usage: python profiles/time_single_detection.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
recs, poses, feat = synth.make_sequence(5, 2, n_static=460, n_movers=24, distortion=True)
ctx = _ffi.Context(0)
for B in (1, 8, 64, 128):
    eng = Engine(B, 2 * B, ctx=ctx, retrack_on_device=True, retrack_slots=max(B, 1))
    for t in range(2):
        eng.upload_scan(t, recs[t])
    for b in range(1, B):
        for t in range(2):
            eng.copy_scan(b * 2 + t, t)
    eng.synchronize()
    for b in range(B):
        eng.init_lane(b, b * 2, feat[:40], poses[0])
    eng.synchronize()
    best = 1e9
    for rep in range(3):
        eng.set_retrack(2)                       # every lane re-detects in the next step
        t0 = time.perf_counter()
        eng.step(np.arange(B, dtype=np.int32) * 2 + 1)
        eng.synchronize()
        dt = time.perf_counter() - t0
        st = eng.stage_times()
        best = min(best, st["retrack"])
        eng.set_retrack(1)
    print(f"{B} lanes: retrack stage {best:.3f} ms = {best*1e3/B:.1f} us per detection")
    eng.close()
ctx.close()
