#!/usr/bin/env python3
"""isolated timing of the image-scale kernels of the device-side detection (roam_engine_time_kernel): `slots` detections per launch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi, synth
if os.environ.get("ROAM_LIB"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["ROAM_LIB"])      # an A/B build (profiles/build_variant.py)
from radarslampy_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
recs, poses, feat = synth.make_sequence(5, 2, n_static=460, n_movers=24, distortion=True)
ctx = _ffi.Context(0)
eng = Engine(B, 2 * B, ctx=ctx, retrack_on_device=True, retrack_slots=B)
for t in range(2):
    eng.upload_scan(t, recs[t])
for b in range(1, B):
    for t in range(2):
        eng.copy_scan(b * 2 + t, t)
eng.synchronize()
for b in range(B):
    eng.init_lane(b, b * 2, feat[:40], poses[0])
eng.step(np.arange(B, dtype=np.int32) * 2 + 1)
eng.synchronize()
for k in os.environ.get("KERNELS", "doh_integral,doh_det_maxima").split(","):
    ms, by = eng.time_kernel(k, 3)
    print(f"{k}: {ms:.2f} ms per {B} detections = {ms*1e3/B:.1f} us each, {by/ms/1e6:.0f} GB/s algorithmic")
eng.close(); ctx.close()
