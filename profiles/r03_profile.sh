#!/bin/bash
# round-3 artefacts (run on the GPU box through gpurun): the default bench unprofiled at the driver's arguments and at 50 / 3, the
# same command under rocprofv3 --kernel-trace --stats, the PMC passes of the detection kernels (-> r03_pmc_traffic.json), the
# round-2 workload for comparison, the single-sequence mode, and the two full-length streaming runs.
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; rm -rf $O; mkdir -p $O
# (every command under its own timeout; the profiled run renders its sequences in-process: rocprofv3 preloads itself into child
# processes too, and its SIGTERM handler there once turned the end of a multiprocessing pool into a hang)
# PMC first: bench.py reads profiles/r03_pmc_traffic.json
timeout 900 bash profiles/pmc_det.sh > $O/pmc_det_kernels.txt 2>&1
python3 profiles/pmc_traffic_r03.py gpurun_out/pmc_det 512 $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
cp $O/pmc_traffic.json profiles/r03_pmc_traffic.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default_20_5.json 2> $O/bench_default_20_5.err
timeout 600 python bench.py --steps 50 --warmup 3 --cpu-pairs 0 > $O/bench_default_50_3.json 2> $O/bench_default_50_3.err
timeout 600 python bench.py --endless --preroll 0 --distinct 4 --steps 20 --warmup 5 --cpu-pairs 0 > $O/bench_round2_workload_20_5.json 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --cpu-pairs 0 --render-procs 1 > $O/bench_default_under_rocprof.json 2> $O/prof.log
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats.csv
python3 profiles/trace_summary.py $O/prof > $O/kernel_trace_summary.csv 2>/dev/null
python3 profiles/timeline_mix.py $O/prof > $O/step_timeline.txt 2>/dev/null
rm -f $O/prof/*/*kernel_trace.csv          # tens of MB; the summaries above are what gets committed
timeout 600 python bench.py --stream > $O/bench_stream_md_on.json 2> $O/bench_stream_md_on.err
timeout 600 python bench.py --stream --no-md > $O/bench_stream_md_off.json 2> $O/bench_stream_md_off.err
timeout 600 python bench.py --h2d --lanes 1024 --cpu-pairs 0 > $O/bench_h2d_streaming.json 2> $O/bench_h2d.err
timeout 600 python profiles/run_full_seq.py 8866 1 > $O/full_seq_1_md_on.json 2> $O/full_seq_on.err
timeout 600 python profiles/run_full_seq.py 8866 0 > $O/full_seq_1_md_off.json 2> $O/full_seq_off.err
bash profiles/one_detection_trace.sh > $O/one_detection_trace.txt 2>&1
for f in $O/bench_default_20_5.json $O/bench_default_50_3.json $O/bench_round2_workload_20_5.json $O/bench_h2d_streaming.json; do cut -c1-200 $f; done
cat $O/full_seq_1_md_on.json $O/full_seq_1_md_off.json; head -12 $O/kernel_stats.csv | cut -d, -f1-5
