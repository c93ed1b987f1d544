#!/usr/bin/env python3
"""Timeline of the last two timed steps of a bench run recorded with rocprofv3 --kernel-trace: start / end / duration (ms) of every
dispatch longer than 0.15 ms, relative to the first one shown - what overlaps what when re-detections are part of the step.
usage: timeline_mix.py <rocprof output dir>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 150000]
# full-batch klt dispatches mark the steps; keep the span of the last "forced" step out: take the klt launches of the timed loop
ik = [i for i, r in enumerate(big) if r["Kernel_Name"].startswith("klt_kernel")]
sel = int(sys.argv[2]) if len(sys.argv) > 2 else -8
lo, hi = ik[sel], ik[sel + 2]
t0 = int(big[lo]["Start_Timestamp"])
for r in big[lo:hi + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{s:9.3f} {e:9.3f} {e - s:8.3f}  {r['Kernel_Name'].split('(')[0][:36]:36s} q{r.get('Queue_Id', '?')}")
