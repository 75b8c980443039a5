cd ${GRAFT_REPO_ROOT:-.}
for nb in 1 2; do echo "=== blocks=$nb"; MF_BF16_BLOCKS=$nb MOCOFLOW_HIP_LIB=build/ab/lib_tl.so timeout 120 python3 tools/timeline.py C3 2>&1 | grep -A 60 "wave 0" | head -75; done
