#!/usr/bin/env python3
"""where a column wave and the row wave of rt_integral_kernel spend a phase (a -DRI_PROF build: ROAM_LIB=variants/libroam_riprof.so):
s_memtime ticks (100 MHz) summed over the workgroups of 512 detections, wave 0 and the row wave of each"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
os.environ["ROAM_LIB_PARTIAL"] = "1"
N = 512
recs, poses, feat = synth.make_sequence(5, 2, n_static=460, n_movers=24, distortion=True)
ctx = _ffi.Context(0)
eng = Engine(N, 2 * N, ctx=ctx, retrack_on_device=True, retrack_slots=N)
for t in range(2): eng.upload_scan(t, recs[t])
for b in range(1, N):
    for t in range(2): eng.copy_scan(b * 2 + t, t)
for b in range(N): eng.init_lane(b, b * 2, feat[:40], poses[0])
eng.set_retrack(2); eng.step(np.arange(N, dtype=np.int32) * 2 + 1); eng.synchronize()
lib = _ffi.load_library()
lib.roam_debug_integral_prof.argtypes = [C.c_void_p, C.c_int]
out = (C.c_ulonglong * 16)()
lib.roam_debug_integral_prof(out, 1)
ms, by = eng.time_kernel("doh_integral", 3)
lib.roam_debug_integral_prof(out, 0)
v = np.array(list(out), float)
names = ["col: loop head", "col: barrier wait", "col: C (tile -> HBM)", "col: A1 (box + taps)", "col: A2 (column sums -> tile)", "row: loop head", "row: barrier wait", "row: B = rest"]
print(f"doh_integral {ms:.2f} ms per {N}")
tot_c = v[:5].sum()
for k in range(5): print(f"  {names[k]:32s} {100 * v[k] / tot_c:5.1f} %")
tot_r = v[8 + 5] + v[8 + 6]
print(f"  row wave: loop body (B + head) {100 * v[8 + 5] / tot_r:5.1f} %   barrier wait {100 * v[8 + 6] / tot_r:5.1f} %")
eng.close(); ctx.close()
