import sys, json
sys.path.insert(0,'.')
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
ctx=_ffi.Context(0)
B=256; T=3
recs,poses,feat=synth.make_sequence(5,T,n_movers=16,distortion=True)
eng=Engine(B,T,ctx=ctx)
for t in range(T): eng.upload_scan(t,recs[t])
for b in range(B): eng.init_lane(b,0,feat,poses[0])
eng.step(np.full(B,1,np.int32)); eng.synchronize()
for name in ("ingest_peaks","warp_quantise","pyramid"):
    ms,by=eng.time_kernel(name,10); print(name, round(ms,4),'ms', round(by/ms/1e6,1),'GB/s')
