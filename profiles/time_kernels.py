#!/usr/bin/env python3
"""isolated timing of the per-scan image kernels (roam_engine_time_kernel) at a lane count with lane-private scans.
usage: python profiles/time_kernels.py [lanes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
recs, poses, feat = synth.make_sequence(5, 2, n_movers=16, distortion=True)
ctx = _ffi.Context(0)
eng = Engine(B, 2 * B, ctx=ctx)
for t in range(2):
    eng.upload_scan(t, recs[t])
for b in range(1, B):
    for t in range(2):
        eng.copy_scan(b * 2 + t, t)
eng.synchronize()
for b in range(B):
    eng.init_lane(b, b * 2, feat, poses[0])
eng.step(np.arange(B, dtype=np.int32) * 2 + 1)
eng.synchronize()
for name in ("ingest_peaks", "warp_quantise", "pyramid"):
    ms, by = eng.time_kernel(name, 10)
    print(f"{name}: {ms:.3f} ms per {B} scans = {ms*1e3/B:.2f} us each, {by/ms/1e6:.0f} GB/s algorithmic")
eng.close(); ctx.close()
