cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_retrack.py -x -q -m gpu 2>&1 | tail -3
bash profiles/scratch/run7.sh
