#!/bin/bash
# kernel trace of single-sequence streaming (bench.py --stream): every kernel of two consecutive steady pairs with its start and end
# usage (on the GPU box): bash profiles/stream_pair_trace.sh [--no-md]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/spt; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --stream --cpu-pairs 0 --cpu-procs 0 --stream-frames 120 "$@" > $O/bench.json 2> $O/prof.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/spt/prof/*/*kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44], r["Queue_Id"]) for r in csv.DictReader(open(f))))
# the pipelined pass is the last one: take two consecutive klt launches near its middle
klt = [i for i, r in enumerate(rows) if r[2].startswith("klt_kernel")]
i0 = klt[len(klt) - 40]; i2 = klt[len(klt) - 38]
t0 = rows[i0][0]
busy = 0
for s, e, n, q in rows[i0:i2]:
    print("%9.1f %9.1f %7.1f us  q%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
print("two pairs span us", (rows[i2][0] - t0) / 1e3)
PY
rm -rf $O/prof
