#!/bin/bash
# PMC passes for the warp kernel (run on the GPU box through gpurun), 512 lanes with lane-private scans
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcw
rm -rf $OUT; mkdir -p $OUT
ARGS="profiles/time_kernels.py ${LANES:-512}"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1; }
run A SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD
run E SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmcw/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("warp_gather"): continue
        agg[k].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-36s max-launch %.4g  (launches %d)" % (c, max(vals), len(vals)))
PY
