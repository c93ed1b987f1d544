#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final2; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $O/gpu_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/gpu_tests.log 2>&1
python bench.py > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
python bench.py > $O/bench_unprofiled2.json 2>> $O/bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py > $O/bench_profiled.json 2> $O/prof.log
python3 profiles/trace_summary.py $O/prof > $O/trace_summary.csv
python3 profiles/timeline.py $O/prof > $O/timeline.txt
cp $O/prof/*/*kernel_stats.csv $O/kernel_stats.csv
python bench.py --h2d > $O/bench_h2d.json 2> $O/bench_h2d.err
tail -3 $O/gpu_tests.log; cut -c1-160 $O/bench_unprofiled.json; cut -c1-160 $O/bench_unprofiled2.json; cut -c1-160 $O/bench_profiled.json; cut -c1-160 $O/bench_h2d.json
