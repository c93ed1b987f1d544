set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/rt_trace; mkdir -p gpurun_out/rt_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rt_trace -- python3 profiles/time_retrack.py 1024 512 > gpurun_out/rt_trace/log.txt 2>&1
f=$(ls gpurun_out/rt_trace/*/*kernel_stats.csv | head -1); head -25 $f | cut -d, -f1-8
rm -f gpurun_out/rt_trace/*/*kernel_trace.csv
