#!/bin/bash
# copies what r03_profile.sh left under gpurun_out/r03 into the tracked profiles/r03_* files and prints the headline numbers
cd "$(dirname "$0")/.."
O=gpurun_out/r03
for f in bench_default_20_5.json bench_default_50_3.json bench_round2_workload_20_5.json bench_default_under_rocprof.json bench_stream_md_on.json bench_stream_md_off.json bench_h2d_streaming.json full_seq_1_md_on.json full_seq_1_md_off.json kernel_stats.csv kernel_trace_summary.csv step_timeline.txt one_detection_trace.txt; do cp $O/$f profiles/r03_$f; done
cp $O/pmc_det_kernels.txt profiles/r03_pmc_det_kernels.txt; cp $O/pmc_traffic.json profiles/r03_pmc_traffic.json
python3 - <<'PY'
import json
for n in ("default_20_5","default_50_3","round2_workload_20_5","default_under_rocprof","h2d_streaming","stream_md_on","stream_md_off"):
    d=json.loads(open(f"profiles/r03_bench_{n}.json").read().strip().splitlines()[-1])
    c=d["config"]; r=d.get("roofline") or {}
    print(n, d["value"], d["ms_per_step"], c.get("retrack_fraction"), c.get("retrack_us_per_lane"), c.get("steady_pairs_per_s"), c.get("steady_mean_tracked"))
    if n=="default_20_5":
        print({k:r.get(k) for k in ("kernel","achieved","frac","traffic","traffic_over_algorithmic","isolated_busy_fractions","avg_launch_ms","units_per_launch","isolated_frac","kernel_ms_per_step_alone","isolated_kernel_ms","in_step_kernel_ms")}); print(d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"]["value"])
    if n.startswith("stream"): print(c["latency_ms_per_pair"])
PY
