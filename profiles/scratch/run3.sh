cd "$GRAFT_REPO_ROOT"
echo base; timeout 300 python profiles/time_doh.py 512 2>&1 | grep det_maxima
for v in variants/libroam_*.so; do echo $v; ROAM_LIB=$v timeout 300 python profiles/time_doh.py 512 2>&1 | grep det_maxima; done
