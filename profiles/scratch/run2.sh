set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_gpu_retrack.py tests/test_gpu_reference_dump.py -x -q -m gpu > gpurun_out/r03a/pytest_retrack.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r03a/pytest_retrack.log
timeout 300 python profiles/time_doh.py 512 > gpurun_out/r03a/doh_strip.log 2>&1; cat gpurun_out/r03a/doh_strip.log
for v in variants/libroam_*.so; do echo $v; ROAM_LIB=$v timeout 900 python -m pytest tests/test_gpu_retrack.py tests/test_gpu_reference_dump.py -x -q -m gpu 2>&1 | tail -2;  ROAM_LIB=$v timeout 300 python profiles/time_doh.py 512 2>&1 | grep det_maxima; done
