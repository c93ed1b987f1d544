#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc/*/runc/*_counter_collection.csv):
per kernel name, mean counter value per dispatch (largest-grid dispatches only)."""
import csv, glob, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/*/runc/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        grid = int(r["Grid_Size"])
        agg[k][(r["Counter_Name"])].append((grid, float(r["Counter_Value"])))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        gmax = max(g for g, _ in v)
        vals = [x for g, x in v if g == gmax]
        print(f"    {c:42s} n={len(vals):3d} mean={sum(vals)/len(vals):16.1f}")
