#!/bin/bash
# kernel trace of the single-sequence driver (bench.py --stream --no-md): where a scan pair's 0.5 ms goes - kernel time against the gaps between
# dependent launches.  Prints, for the steady pairs of the awaited (synchronous) pass, the dispatches of a few consecutive pairs.
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/stream_tl; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --stream ${MD:---no-md} --stream-frames 120 > $O/bench.json 2> $O/err.log
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/stream_tl/prof/*/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
# steps are delimited by the peak kernel; take the last 40 steps (the synchronous pass) and keep those without a detection
idx = [i for i, r in enumerate(rows) if name(r).startswith("klt_kernel")]
steps = []
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b]
    if any(name(r).startswith("rt_blobs") and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 20000 for r in seg): continue
    steps.append(seg)
steps = steps[-30:]
busy = collections.Counter(); n = 0; span = 0.0; ksum = 0.0
for seg in steps:
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    span += (int(rows[rows.index(seg[-1]) + 1]["Start_Timestamp"]) - t0) / 1e3 if rows.index(seg[-1]) + 1 < len(rows) else (t1 - t0) / 1e3
    for r in seg:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        busy[name(r)] += d; ksum += d
    n += 1
print("steady pairs averaged: %d; klt-to-klt span %.1f us per pair, sum of kernel durations %.1f us (kernels overlap across streams)" % (n, span / n, ksum / n))
for k, v in busy.most_common(24):
    print("  %-36s %8.1f us per pair" % (k, v / n))
seg = steps[-1]; t0 = int(seg[0]["Start_Timestamp"])
print("one pair, dispatch by dispatch (start, end, duration us):")
for r in seg:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("  %8.1f %8.1f %7.1f  %s" % (s, e, e - s, name(r)))
PY
rm -rf $O/prof
