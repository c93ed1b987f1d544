set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03a
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r03a/pytest_all.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r03a/pytest_all.log
timeout 300 python profiles/time_retrack.py 1024 512 2>&1 | tail -4
timeout 300 python profiles/time_retrack.py 1024 1024 2>&1 | tail -4
