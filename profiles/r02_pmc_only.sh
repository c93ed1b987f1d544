#!/bin/bash
# the PMC part of r02_profile.sh alone
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; OUT=$O/pmc; rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --lanes 512 --retrack-slots 512 --steps 4 --warmup 2 --cpu-pairs 0 --kernel-reps 3"
run() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1; }
run A SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD
run C FETCH_SIZE
run D WRITE_SIZE
python3 profiles/pmc_traffic.py $OUT 512 $O/pmc_traffic.json
