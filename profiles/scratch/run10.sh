cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03b
( time python bench.py --stream ) > gpurun_out/r03b/stream_md_on.json 2> gpurun_out/r03b/stream_md_on.err; tail -3 gpurun_out/r03b/stream_md_on.err; cat gpurun_out/r03b/stream_md_on.json
( time python bench.py --stream --no-md ) > gpurun_out/r03b/stream_md_off.json 2> gpurun_out/r03b/stream_md_off.err; cat gpurun_out/r03b/stream_md_off.json
python bench.py --endless --steps 20 --warmup 5 --cpu-pairs 0 > gpurun_out/r03b/bench_endless.json 2>&1; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r03b/bench_endless.json") if l.startswith("{")][0]); c=d["config"]
print("endless", d["value"], d["ms_per_step"], c["retrack_fraction"], c["retracks_per_step"], c["rejected_fraction"])
PY
