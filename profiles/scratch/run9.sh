cd "$GRAFT_REPO_ROOT"
echo base 512; timeout 300 python profiles/time_doh.py 512 2>&1 | grep doh_
echo base 1024; timeout 300 python profiles/time_doh.py 1024 2>&1 | grep doh_
for v in variants/libroam_*.so; do echo $v 512; ROAM_LIB=$v timeout 300 python profiles/time_doh.py 512 2>&1 | grep doh_; echo $v 1024;  ROAM_LIB=$v timeout 300 python profiles/time_doh.py 1024 2>&1 | grep doh_; done
