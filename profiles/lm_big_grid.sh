cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for g in 64 512 4096; do
O=gpurun_out/tl; rm -rf $O; mkdir -p $O
ROAM_LM_BIG_GRID=$g timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --steps 12 --warmup 5 --cpu-pairs 0 --cpu-procs 0 --no-segments --render-procs 1 > $O/bench.json 2> $O/prof.log
python3 profiles/timeline_mix.py $O/prof > $O/step_timeline.txt 2>/dev/null; rm -rf $O/prof
echo "== grid $g"; sed -n 18,27p $O/step_timeline.txt; cut -c60-130 $O/bench.json
done
