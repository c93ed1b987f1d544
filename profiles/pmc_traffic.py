#!/usr/bin/env python3
"""Build profiles/r01_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of profiles/pmc_run.sh
(gpurun_out/pmc/C and /D): per kernel, mean KB per launch over the largest-grid dispatches and the
resulting HBM bytes per scan (lanes = scans per launch).  usage: pmc_traffic.py <pmc dir> <lanes> <out.json>"""
import csv, glob, json, sys, collections
root, lanes, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]


def collect(sub, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{root}/{sub}/runc/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            agg[k].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    res = {}
    for k, v in agg.items():
        g = max(x for x, _ in v)
        vals = [y for x, y in v if x == g]
        res[k] = sum(vals) / len(vals)
    return res


fetch, write = collect("C", "FETCH_SIZE"), collect("D", "WRITE_SIZE")
kernels = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    kernels[k] = {"fetch_KB_per_launch": round(f, 1), "write_KB_per_launch": round(w, 1),
                  "traffic_bytes_per_scan": int((f + w) * 1024 / lanes)}
json.dump({"lanes": lanes,
           "source": "profiles/pmc_run.sh passes C (FETCH_SIZE) and D (WRITE_SIZE), rocprofv3 --pmc, largest-grid dispatches",
           "note": "units as reported by rocprofv3 (KB). gfx950 FETCH_SIZE under-reports wide coalesced reads by up to 2x and is "
                   "uncalibrated for byte/dword accesses (MI355X_MICROARCH.md HBM section); WRITE_SIZE is exact for 16-B stores only",
           "kernels": kernels}, open(out, "w"), indent=1)
print(json.dumps(kernels.get("warp_gather_kernel")))
