#!/usr/bin/env python3
"""Per-(kernel, grid) duration summary from a rocprofv3 --kernel-trace CSV.  The --stats table
averages over ALL dispatches of a kernel name; the engine launches the same kernels with a
1-lane grid while initialising lanes and with the full-batch grid inside roam_engine_step, so the
full-batch rows below are the ones bench.py's HIP-event timing must agree with."""
import collections
import csv
import glob
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof3"
f = sorted(glob.glob(root + "/*/*kernel_trace.csv"))[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    grid = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])) if "Grid_Size_X" in r else (int(r["Grid_Size"]),)
    d[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel,grid_threads,calls,avg_us,min_us,max_us")
for (name, grid), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name},{'x'.join(map(str, grid))},{len(v)},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f}")
