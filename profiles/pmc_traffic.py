#!/usr/bin/env python3
"""Build profiles/rNN_pmc_traffic.json from the PMC passes of profiles/pmc_det.sh (gpurun_out/pmc_det/{A..E}): per detection
kernel and launch of `units` detections - memory-side bytes (gfx950-corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE
tallies 128-byte requests at 64 bytes, x 2; cross-check TCC_MISS x 128 B), VALU wave-instructions, LDS array cycles - together
with the hash of the kernel's source file, so that bench.py can tell whether the numbers still describe the code it runs.
The hash is the one profiles/pmc_det.sh wrote NEXT TO the counters when it measured (source_fingerprint.txt: the translation unit,
every header of csrc/ and the build flags - radarslampy_amd/build.py fingerprint), not one taken when this script runs.
Round 6: the integral image counts with the tiles that exist (the ones rt_det_strip_kernel loads and rt_integral_kernel therefore
writes: 87.7 % of a 2024 x 2024 image, retrack_build_phases) - IMAGE_FRACTION below.
usage: pmc_traffic.py <pmc dir> <units per launch> <out.json>"""
import collections, csv, glob, hashlib, json, os, sys
root, units, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{root}/*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
    return {k: max(v) for k, v in agg.items()}          # the busiest launch (= the full chunk)


C = {c: collect(c) for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_MISS_sum", "TCC_HIT_sum", "TCC_EA0_RDREQ_sum", "TCP_TCC_READ_REQ_sum",
                             "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")}
IMAGE_FRACTION = 0.8770                       # needed tiles / all tiles of the 2024 x 2024 image (engine: rt_image_px; 0.8789 with the 32-row tiles of -DRI_ROWS=32)
ALGO = {"rt_det_strip_kernel": ("retrack.hip", IMAGE_FRACTION * 2024 * 2024 * 8.0,
                                "float64 integral image marched in 62-column strips through an LDS ring: memory-side reads {ratio:.2f} x the algorithmic bytes "
                                "(L2 hit rate {hit:.0%} on the lines neighbouring strips share); VALU issue, LDS array and HBM each about half busy - "
                                "a step's phases (staging, per-wave skip test, 16 box pairs, maxima, barrier) run one after the other inside a workgroup "
                                "and only two workgroups fit a CU (74 KB of LDS each)"),
        "rt_integral_kernel": ("retrack.hip", 400 * 2025 + IMAGE_FRACTION * 2024 * 2024 * 8.0,
                               "one sweep, both float64 prefix sums in NumPy's sequential order, the tiles the determinant kernel reads written once (28.7 MB per detection; algorithmic "
                               "bytes = polar payload + that image, the sampling-map words it also reads are the same table for every detection and mostly L2 hits: "
                               "memory-side {ratio:.2f} x algorithmic, L2 hit rate {hit:.0%}); latency chain of ~955 phases per detection, two workgroups per CU")}
fp_file = os.path.join(root, "source_fingerprint.txt")
FP = open(fp_file).read().strip() if os.path.exists(fp_file) else None
kernels = {}
for k, (src, algo, note) in ALGO.items():
    if k not in C["FETCH_SIZE"]:
        continue
    f2 = 2.0 * C["FETCH_SIZE"][k] * 1024.0
    w = C["WRITE_SIZE"].get(k, 0.0) * 1024.0
    miss = C["TCC_MISS_sum"].get(k, 0.0) * 128.0
    hit = C["TCC_HIT_sum"].get(k, 0.0)
    req = hit + C["TCC_MISS_sum"].get(k, 0.0)
    kernels[k] = {"source_file": src, "source_fingerprint": FP,
                  "algorithmic_bytes_per_launch": int(algo * units),
                  "fetch_bytes_uncorrected": int(C["FETCH_SIZE"][k] * 1024), "fetch_bytes_x2_gfx950": int(f2), "tcc_miss_x_128B": int(miss),
                  "write_bytes": int(w), "hbm_bytes_per_launch_corrected": int(f2 + w),
                  "l2_hit_rate": round(hit / req, 4) if req else None,
                  "valu_wave_insts_per_launch": int(C["SQ_INSTS_VALU"].get(k, 0)), "lds_wave_insts_per_launch": int(C["SQ_INSTS_LDS"].get(k, 0)),
                  "lds_active_cycles_per_launch": int(C["SQ_LDS_IDX_ACTIVE"].get(k, 0)), "lds_bank_conflict_cycles": int(C["SQ_LDS_BANK_CONFLICT"].get(k, 0)),
                  "note": note.format(ratio=(f2 + w) / (algo * units), hit=hit / req if req else 0.0)}
json.dump({"units_per_launch": units,
           "source": "profiles/pmc_det.sh: rocprofv3 --pmc passes A-E on profiles/time_doh.py (the detection kernels alone, `units` detections per launch), busiest launch",
           "kernels": kernels}, open(out, "w"), indent=1)
for k, v in kernels.items():
    print(k, json.dumps({x: v[x] for x in ("hbm_bytes_per_launch_corrected", "algorithmic_bytes_per_launch", "tcc_miss_x_128B", "l2_hit_rate")}))
