import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
recs, poses, feat = synth.make_sequence(5, 2, n_static=460, n_movers=24, distortion=True)
ctx = _ffi.Context(0)
eng = Engine(1, 2, ctx=ctx, retrack_on_device=True, retrack_slots=1)
for t in range(2): eng.upload_scan(t, recs[t])
eng.init_lane(0, 0, feat[:40], poses[0])
for rep in range(3):
    eng.set_retrack(2); eng.step(np.array([1], np.int32)); eng.synchronize(); eng.set_retrack(1)
eng.close(); ctx.close()
