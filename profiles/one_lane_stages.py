import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
recs, poses, feat = synth.make_sequence(5, 7, n_movers=120, distortion=True, scintillation=0.6) if "scintillation" in synth.make_sequence.__code__.co_varnames else synth.make_sequence(5, 7, n_movers=120, distortion=True)
ctx = _ffi.Context(0)
T = len(recs)
eng = Engine(1, T, ctx=ctx, retrack_on_device=True)
for t in range(T):
    eng.upload_scan(t, recs[t])
eng.init_lane_detect(0, 0, poses[0])
acc = {}
order = list(range(1, T)) + list(range(T - 2, -1, -1))
for rep in range(3):
    for t in order:
        eng.step([t]); eng.synchronize()
        r = eng.results()[0]
        st = eng.stage_times()
        print(t, r["n_tracked"], r["n_good"], r["n_inliers"], "retrack" if r["retrack"] else "", {k: round(v, 3) for k, v in st.items() if v > 0.02})
eng.close(); ctx.close()
