cd "$GRAFT_REPO_ROOT"
ROAM_LIB=variants/libroam_prof.so timeout 300 python profiles/time_doh.py 512 2>&1 | sort | uniq -c | sort -k3,3n -k5,5n | tail -40
