#!/usr/bin/env python3
"""cost of the device-side retrack (retrack.hip): B lanes that ALL run out of features in one step.
usage: python profiles/time_retrack.py [lanes] [slots]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 0
recs, poses, feat = synth.make_sequence(5, 3, n_static=460, n_movers=24, distortion=True)
ctx = _ffi.Context(0)
eng = Engine(B, 3 * B, ctx=ctx, retrack_on_device=True, retrack_slots=slots)
for t in range(3):
    eng.upload_scan(t, recs[t])
for b in range(1, B):
    for t in range(3):
        eng.copy_scan(b * 3 + t, t)
eng.synchronize()
for b in range(B):
    eng.init_lane(b, b * 3, feat[:40], poses[0])
for t in (1, 2):
    t0 = time.perf_counter()
    eng.step(np.arange(B, dtype=np.int32) * 3 + t)
    eng.synchronize()
    dt = time.perf_counter() - t0
    r = eng.results_array()
    st = eng.stage_times()
    print(f"step {t}: {dt*1e3:.2f} ms wall, retrack stage {st['retrack']:.2f} ms for {int((r['flags'] & 8 != 0).sum())} of {B} lanes "
          f"({st['retrack']*1e3/max(1,int((r['flags'] & 8 != 0).sum())):.1f} us per lane), features after: {r['n_after_retrack'][:4]}, "
          f"overflow flags {int(((r['flags'] >> 8) & 15).max())}")
    print("   stages:", {k: round(v, 2) for k, v in st.items()}, "inliers", r['n_inliers'][:3], "tracked", r['n_tracked'][:3])
    # force every lane to retrack again in the next step
    for b in range(B):
        pass
eng.close(); ctx.close()
