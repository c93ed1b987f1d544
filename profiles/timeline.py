#!/usr/bin/env python3
"""Timeline of the last full-batch step in a rocprofv3 --kernel-trace CSV: start / end (ms, relative to the
step's first kernel) of every dispatch, to see what overlaps what."""
import csv, glob, sys
root = sys.argv[1]
f = sorted(glob.glob(root + "/*/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last full-batch klt dispatch belongs to the last step; the step starts with the peak kernel before it
ik = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("klt_kernel")][-1]
j0 = [i for i, r in enumerate(rows[:ik]) if r["Kernel_Name"].startswith("peaks_rows")][-1]
t0 = int(rows[j0]["Start_Timestamp"])
for r in rows[j0:ik + 14]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{s:8.3f} {e:8.3f} {e - s:8.3f}  {r['Kernel_Name'].split('(')[0][:40]}")
