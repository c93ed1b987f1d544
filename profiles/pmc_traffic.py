#!/usr/bin/env python3
"""Build profiles/rNN_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes of profiles/pmc_run.sh
(gpurun_out/pmc/A, C, D): per kernel, the launch with the largest grid - and, for the persistent-grid detection kernels whose
work per launch varies, the busiest of those - gives HBM bytes and VALU wave-instructions per launch.
usage: pmc_traffic.py <pmc dir> <lanes> <out.json>"""
import collections, csv, glob, json, sys
root, lanes, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]


def collect(sub, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{root}/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            agg[k].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return {k: max(y for x, y in v if x == max(g for g, _ in v)) for k, v in agg.items()}


fetch, write, valu = collect("C", "FETCH_SIZE"), collect("D", "WRITE_SIZE"), collect("A", "SQ_INSTS_VALU")
kernels = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    kernels[k] = {"fetch_KB_per_launch": round(f, 1), "fetch_KB_per_launch_x2_gfx950_wide_reads": round(2 * f, 1), "write_KB_per_launch": round(w, 1),
                  "traffic_bytes_per_launch": int((f + w) * 1024), "traffic_bytes_per_scan": int((f + w) * 1024 / lanes),
                  "valu_wave_insts_per_launch": int(valu.get(k, 0))}
json.dump({"lanes": lanes,
           "source": "profiles/pmc_run.sh passes A (SQ_INSTS_VALU), C (FETCH_SIZE) and D (WRITE_SIZE), rocprofv3 --pmc, busiest largest-grid dispatch",
           "note": "units as reported by rocprofv3 (KB). On gfx950 FETCH_SIZE reports exactly half of the bytes of wide coalesced streaming reads "
                   "(16 B per lane; MI355X_MICROARCH.md HBM section) - the x2 column applies that correction, which holds for such reads only; "
                   "traffic_bytes_* use the uncorrected counter.  The detection kernels (rt_*) process `retrack_slots` detections per launch, not `lanes`.",
           "kernels": kernels}, open(out, "w"), indent=1)
for k in ("warp_gather_kernel", "rt_det_mask_kernel", "rt_integral_kernel"):
    print(k, json.dumps(kernels.get(k)))
