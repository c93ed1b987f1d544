"""time kernels of several library builds at bench scale (lane-private scans): python profiles/time_variants_big.py a.so b.so"""
import sys, subprocess, os
sys.path.insert(0, '.')
if len(sys.argv) > 2:
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, __file__, lib])
    sys.exit(0)
from radarslampy_amd import _ffi
_ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np
from radarslampy_amd import synth
from radarslampy_amd.engine import Engine
ctx = _ffi.Context(0)
B = int(os.environ.get("LANES", "1024")); T = 2
recs, poses, feat = synth.make_sequence(5, T, n_movers=16, distortion=True)
eng = Engine(B, B * T, ctx=ctx)
for t in range(T): eng.upload_scan(t, recs[t])
for b in range(1, B):
    for t in range(T): eng.copy_scan(b * T + t, t)
for b in range(B): eng.init_lane(b, b * T, feat, poses[0])
eng.step(np.arange(B, dtype=np.int32) * T + 1); eng.synchronize()
for name in os.environ.get("KERNELS", "warp_quantise").split(","):
    ms, by = eng.time_kernel(name, 10)
    print(os.path.basename(sys.argv[1]), name, round(ms, 4), 'ms', round(by / ms / 1e6, 1), 'GB/s')
