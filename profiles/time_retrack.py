"""Cost of one retrack (appendNewFeatures on a resident scan): DoH maxima on the device + host
bookkeeping (response order, _prune_blobs) + SSC-ANMS on the device + dedupe + keyframe refresh."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
ctx = _ffi.Context(0)
recs, poses, feat = synth.make_sequence(5, 2, n_static=460)
eng = Engine(1, 2, ctx=ctx)
eng.upload_scan(0, recs[0]); eng.upload_scan(1, recs[1])
eng.init_lane(0, 0, feat[:70], poses[0])
eng.step([1]); eng.results()
for rep in range(3):
    t0 = time.perf_counter(); new = eng.detect_features(1); t1 = time.perf_counter()
    pts = eng.retrack_lane(0, 1); t2 = time.perf_counter()
    print(f"detect_features {1e3*(t1-t0):.2f} ms ({len(new)} features) | retrack_lane total {1e3*(t2-t1):.2f} ms -> {len(pts)} features")
