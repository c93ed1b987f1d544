cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k mds 2>&1 | tail -3
for mode in 0 1; do
  rm -rf gpurun_out/lmab$mode; 
  ROAM_LM_BLOCK=$mode rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lmab$mode -- python3 profiles/lm_lone.py > gpurun_out/lmab$mode.log 2>&1
  echo "== ROAM_LM_BLOCK=$mode"; cat gpurun_out/lmab$mode.log | tail -4
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/lmab$mode/*/*kernel_trace.csv")[0]
rows = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "mds_lm" in r["Kernel_Name"]]
# 10 reps per tag, 4 tags
for k in range(4):
    d = [x[1] for x in rows[10 * k + 2: 10 * k + 10]]
    print(rows[10 * k][0][:40], "median us", sorted(d)[len(d) // 2] / 1e3)
PY
  rm -rf gpurun_out/lmab$mode
done
