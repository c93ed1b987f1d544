cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03b
( time python bench.py --steps 20 --warmup 5 --cpu-pairs 0 ) > gpurun_out/r03b/bench_20_5.json 2> gpurun_out/r03b/bench_20_5.err
( time python bench.py --steps 50 --warmup 3 --cpu-pairs 0 ) > gpurun_out/r03b/bench_50_3.json 2> gpurun_out/r03b/bench_50_3.err
tail -5 gpurun_out/r03b/bench_20_5.err
python - <<'PY'
import json
for f in ("bench_20_5","bench_50_3"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r03b/{f}.json") if l.startswith("{")][0])
        c=d["config"]
        print(f, d["value"], d["ms_per_step"], "rtfrac", c["retrack_fraction"], c["retracks_per_step"], "rej", c["rejected_fraction"], "tracked", c["mean_tracked"], "steady", c.get("steady_pairs_per_s"), c.get("steady_mean_tracked"), "rt_us", c.get("retrack_us_per_lane"))
        print("   ", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["isolated_frac"], d["roofline"]["isolated_kernel_ms"], c.get("stage_ms_last_mix_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
