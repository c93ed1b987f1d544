#!/usr/bin/env python3
"""what a single sequence pays once: engine creation, pinned ring, first uploads, first-frame detection, tear-down (ms)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
recs, poses, feat = synth.make_sequence(5, 6, n_movers=20, distortion=True)
ctx = _ffi.Context(0)
for rep in range(3):
    t = [time.perf_counter()]
    eng = Engine(1, 8, ctx=ctx, retrack_on_device=True, stage_events=False); t.append(time.perf_counter())
    pinned = ctx.host_alloc((8, 400, 3779)); t.append(time.perf_counter())
    for i in range(4):
        pinned[i] = recs[i]; eng.upload_scans_async(i, pinned[i], n=1)
    eng.synchronize(); t.append(time.perf_counter())
    eng.init_lane_detect(0, 0, poses[0]); t.append(time.perf_counter())
    eng.step([1]); eng.synchronize(); t.append(time.perf_counter())
    eng.close(); t.append(time.perf_counter())
    ctx.host_free(pinned); t.append(time.perf_counter())
    names = ["engine create", "pinned ring alloc (12 MB)", "4 uploads + sync", "init_lane_detect", "first step + sync", "engine close", "pinned ring free"]
    print("  ".join(f"{n} {1e3 * (b - a):.2f}" for n, a, b in zip(names, t[:-1], t[1:])))
ctx.close()
