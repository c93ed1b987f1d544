#!/bin/bash
# round-6 artefacts (run on the GPU box through gpurun; the PNG-fed stream now through the native decoder, the isolated detection timings, slots A/B): PMC passes of the detection kernels (-> r06_pmc_traffic.json, with the
# fingerprint of the measured sources), the default bench at the driver's arguments (now with the endless / stream segments and the
# strict roofline accounting) and at 50 / 3, the same command under rocprofv3 --kernel-trace --stats, the single-sequence mode, the
# config-5 loop at world 1, clique timings, a one-detection trace.
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; rm -rf $O; mkdir -p $O
timeout 900 bash profiles/pmc_det.sh > $O/pmc_det_kernels.txt 2>&1
python3 profiles/pmc_traffic.py gpurun_out/pmc_det 512 $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
cp $O/pmc_traffic.json profiles/r06_pmc_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default_20_5.json 2> $O/bench_default_20_5.err
timeout 600 python bench.py --steps 50 --warmup 3 --cpu-pairs 0 --cpu-procs 0 --no-segments > $O/bench_default_50_3.json 2> $O/bench_default_50_3.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --cpu-pairs 0 --cpu-procs 0 --no-segments --render-procs 1 > $O/bench_default_under_rocprof.json 2> $O/prof.log
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats.csv
python3 profiles/trace_summary.py $O/prof > $O/kernel_trace_summary.csv 2>/dev/null
python3 profiles/timeline_mix.py $O/prof > $O/step_timeline.txt 2>/dev/null
rm -f $O/prof/*/*kernel_trace.csv
timeout 600 python bench.py --stream --png > $O/bench_stream_md_on.json 2> $O/bench_stream_md_on.err
timeout 600 python bench.py --stream --png --no-md > $O/bench_stream_md_off.json 2> $O/bench_stream_md_off.err
timeout 900 python bench.py --config5 > $O/bench_config5_world1.json 2> $O/bench_config5.err
timeout 600 python bench.py --h2d --lanes 1024 --cpu-pairs 0 --cpu-procs 0 > $O/bench_h2d_streaming.json 2> $O/bench_h2d.err
timeout 300 python profiles/time_clique.py > $O/time_clique.txt 2>&1
timeout 300 python profiles/time_kernels.py 4096 > $O/time_kernels.txt 2>&1
timeout 300 python profiles/time_doh.py 512 >> $O/time_kernels.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-pairs 0 --cpu-procs 0 --no-segments --retrack-slots 512 > $O/bench_default_slots512.json 2> $O/bench_default_slots512.err
bash profiles/one_detection_trace.sh > $O/one_detection_trace.txt 2>&1
LANES=512 timeout 600 bash profiles/pmc_warp.sh > $O/pmc_warp.txt 2>&1
for f in $O/bench_default_20_5.json $O/bench_default_50_3.json $O/bench_config5_world1.json $O/bench_h2d_streaming.json; do cut -c1-220 $f; done
cat $O/time_clique.txt $O/time_kernels.txt; head -12 $O/kernel_stats.csv | cut -d, -f1-5
# only the summaries travel back (gpurun merges at most 64 MiB): the raw counter / trace CSVs stay on the box
rm -rf gpurun_out/pmc_det gpurun_out/pmc_fused gpurun_out/pmcw $O/prof gpurun_out/pmc_rt 2>/dev/null
du -sh gpurun_out | tail -1
