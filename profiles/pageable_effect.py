#!/usr/bin/env python3
"""does a pageable host->device copy earlier in the process slow a later single-sequence run down?  (bench.py's default run measures
the single-sequence segment after a 4 096-lane run whose set-up uploads records from pageable memory)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
from radarslampy_amd.RawROAMSystem import stream_records
recs, poses, feat = synth.make_sequence(5, 40, n_movers=20, distortion=True)
recs = recs + recs[-2::-1] + recs[1:] + recs[-2::-1]
ctx = _ffi.Context(0)
def rate(md):
    flags = dict(rejectOutliers=True, correctMotionDistortion=md)
    stream_records(iter(recs[:12]), 12, poses[0], flags, ctx)
    best = 0
    for rep in range(2):
        t0 = time.perf_counter()
        stream_records(iter(recs), len(recs), poses[0], flags, ctx)
        best = max(best, (len(recs) - 1) / (time.perf_counter() - t0))
    return round(best, 1)
print("fresh process:            md on", rate(True), "md off", rate(False))
eng = Engine(64, 128, ctx=ctx)
for i in range(128):
    eng.upload_scan(i, recs[i % len(recs)])            # pageable source
eng.close()
print("after pageable uploads:   md on", rate(True), "md off", rate(False))
big = Engine(4096, 8192, ctx=ctx, retrack_on_device=True)
big.close()
print("after a 4096-lane engine: md on", rate(True), "md off", rate(False))
ctx.close()
