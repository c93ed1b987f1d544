#!/usr/bin/env python3
"""Per-(kernel, grid) duration summary from a rocprofv3 --kernel-trace CSV.  The --stats table
averages over ALL dispatches of a kernel name; the engine launches the same kernels with a
1-lane grid while initialising lanes and with the full-batch grid inside roam_engine_step, so the
full-batch rows below are the ones bench.py's HIP-event timing must agree with: phase "steps" = launches inside
the warm-up / timed steps (roofline.avg_launch_ms averages the timed ones), phase "isolated" = the re-launches
after the timed region (roofline.isolated_kernel_ms).
The detection kernels (rt_*) are launched once per chunk of `retrack_slots` lanes whatever the number of lanes that really
re-detect - only the device knows it - and a launch beyond that number returns at once (a few microseconds): the last two columns
count and average the launches that did work (> 1 ms).  bench.py's live figure for a detection kernel is the FIRST chunk of every
timed step (always a full one on the default workload): compare it with `avg_us_busy`."""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof3"
f = sorted(glob.glob(root + "/*/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
# bench.py re-launches the front-end kernels alone AFTER the timed steps (roofline `isolated_*` fields): every
# dispatch that starts after the last tracker (klt_kernel) dispatch has ended belongs to that phase
klt_end = max((int(r["End_Timestamp"]) for r in rows if r["Kernel_Name"].startswith("klt_kernel")), default=None)
d = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    grid = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])) if "Grid_Size_X" in r else (int(r["Grid_Size"]),)
    phase = "isolated" if klt_end is not None and int(r["Start_Timestamp"]) > klt_end + 2_000_000 else "steps"
    d[(name, grid, phase)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel,grid_threads,phase,calls,avg_us,min_us,max_us,calls_busy,avg_us_busy")
for (name, grid, phase), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    busy = [x for x in v if x > 1000.0]
    print(f"{name},{'x'.join(map(str, grid))},{phase},{len(v)},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{len(busy)},{(sum(busy)/len(busy) if busy else 0.0):.2f}")
