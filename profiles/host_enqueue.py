#!/usr/bin/env python3
"""host time of ONE roam_engine_step call of a 1-lane engine (the enqueue of ~25 kernels, events and the result copy) against the
device time of the step: a single sequence cannot go faster than the slower of the two.  usage: python profiles/host_enqueue.py [--no-md]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
md = "--no-md" not in sys.argv
T = 8
recs, poses, feat = synth.make_sequence(5, T, n_movers=20, distortion=True)
ctx = _ffi.Context(0)
for ev in (True, False):
    eng = Engine(1, T, ctx=ctx, retrack_on_device=True, motion_distortion=md, stage_events=ev)
    for t in range(T):
        eng.upload_scan(t, recs[t])
    eng.init_lane_detect(0, 0, poses[0])
    order = (list(range(1, T)) + list(range(T - 2, -1, -1))) * 12
    eng.step([order[0]]); eng.synchronize()
    t0 = time.perf_counter()
    for t in order[1:]:
        eng.step([t])
    t1 = time.perf_counter()
    eng.synchronize()
    t2 = time.perf_counter()
    n = len(order) - 1
    print(f"stage_events={ev} md={md}: host enqueue {1e6 * (t1 - t0) / n:.1f} us per step, whole {1e6 * (t2 - t0) / n:.1f} us per step ({n} steps)")
    eng.close()
ctx.close()
