#!/bin/bash
# round-2 artefacts (run on the GPU box through gpurun): default bench unprofiled, the same command under
# rocprofv3 --kernel-trace --stats, and the PMC passes at the benched lane count.
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --cpu-pairs 0 > $O/bench_profiled.json 2> $O/prof.log
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats.csv
python3 profiles/trace_summary.py $O/prof > $O/trace_summary.csv 2>/dev/null
rm -f $O/prof/*/*kernel_trace.csv          # tens of MB; the summary above is what gets committed
OUT=$O/pmc; mkdir -p $OUT
# PMC serialises every dispatch: at 4096 lanes the 70 000 dispatches of the lane set-up alone take longer than a GPU slot lasts (rocprofv3 aborted);
# the detection kernels work on `retrack_slots` = 512 detections per launch whatever the lane count, so 512 lanes measure the same launches
ARGS="bench.py --lanes 512 --retrack-slots 512 --steps 4 --warmup 2 --cpu-pairs 0 --kernel-reps 3"
run() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1; }
run A SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD
run C FETCH_SIZE
run D WRITE_SIZE
python3 profiles/pmc_traffic.py $OUT 512 $O/pmc_traffic.json
cut -c1-300 $O/bench_unprofiled.json; head -14 $O/kernel_stats.csv | cut -d, -f1-5
