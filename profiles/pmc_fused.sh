#!/bin/bash
# PMC passes (instruction counts, waits, LDS) of the three image-scale detection kernels side by side: rt_integral_kernel,
# rt_det_strip_kernel, rt_fused_kernel (profiles/time_doh.py, 512 detections per launch).  --pmc is never combined with trace domains.
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_fused
rm -rf $OUT; mkdir -p $OUT
ARGS="profiles/time_doh.py ${LANES:-512}"
export KERNELS=doh_integral,doh_det_maxima,doh_fused
run() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1; }
run A SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run E SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SMEM
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmc_fused/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("rt_"): continue
        agg[k].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-28s max-launch %.6g  (launches %d)" % (c, max(vals), len(vals)))
PY
