#!/bin/bash
# PMC passes for the determinant kernel of the device-side detection alone (profiles/time_doh.py: 512 detections per launch).
# One rocprofv3 invocation per counter group; --pmc is never combined with sys / hip / hsa trace domains.
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_det
rm -rf $OUT; mkdir -p $OUT
# what is being measured: the hash of retrack.hip + every header + the build flags, taken NOW (bench.py compares it with the tree it runs)
python3 radarslampy_amd/build.py --fingerprint retrack.hip > $OUT/source_fingerprint.txt
ARGS="profiles/time_doh.py ${LANES:-512}"
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1; }
run A SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run B TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run C FETCH_SIZE
run D WRITE_SIZE
run E SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SMEM
python3 - <<'PY'
import csv, glob, collections, os
agg = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmc_det/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith(os.environ.get("KFILTER", "rt_")): continue
        d = agg[k].setdefault(r["Counter_Name"], [])
        d.append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-36s max-launch %.6g  (launches %d)" % (c, max(vals), len(vals)))
PY
