for n in "$@"; do L=$GRAFT_REPO_ROOT/build/ab/lib_$n.so; [ $n = default ] && L=""; echo "== $n"; MOCOFLOW_HIP_LIB=$L timeout 300 python tools/time_dump_pass.py 1024 2>&1 | grep "bf16x3"; done
