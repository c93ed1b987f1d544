#!/bin/bash
# kernel trace of ONE device-side detection (a 1-lane engine, three forced re-detections): what a lone retrack pair waits for
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/one; mkdir -p gpurun_out/one
cat > gpurun_out/one/run.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from radarslampy_amd import _ffi, synth
from radarslampy_amd.engine import Engine
recs, poses, feat = synth.make_sequence(5, 2, n_static=460, n_movers=120, scintillation=0.6, distortion=True)
ctx = _ffi.Context(0)
eng = Engine(1, 2, ctx=ctx, retrack_on_device=True)
for t in range(2): eng.upload_scan(t, recs[t])
eng.init_lane(0, 0, feat[:40], poses[0])
for rep in range(3):
    eng.set_retrack(2); eng.step([1]); eng.synchronize(); print("retrack stage ms", eng.stage_times()["retrack"])
eng.close(); ctx.close()
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/one/prof -- python3 gpurun_out/one/run.py > gpurun_out/one/log.txt 2>&1
grep "retrack stage" gpurun_out/one/log.txt
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/one/prof/*/*kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f))))
# the last detection: from the last g4_update_kernel on (its last block lists the lanes that re-detect)
i0 = max(i for i, r in enumerate(rows) if r[2].startswith("g4_update"))
t0 = rows[i0][0]
for s, e, n in rows[i0:]:
    if e - s > 3000 or n.startswith("rt_") or n.startswith("ssc"):
        print("%9.1f %9.1f %8.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n[:50]))
PY
