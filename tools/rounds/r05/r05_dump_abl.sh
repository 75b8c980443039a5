# x3 training forward with the dump stores ablated (build/ab/lib_d1.so: one of the four row stores of a tile; lib_d2.so: all four to
# the first one's address -- same store count, a quarter of the bytes); kernel time from rocprofv3
cd /tmp && export TMPDIR=/tmp
for n in default d1 d2; do L=$GRAFT_REPO_ROOT/build/ab/lib_$n.so; [ $n = default ] && L=""
  rm -rf /tmp/pr_$n; MOCOFLOW_HIP_LIB=$L MF_TRAIN_FWD=bf16x3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$n -- python3 $GRAFT_REPO_ROOT/tools/time_moco_step.py 1024 > /tmp/pr_$n.log 2>&1
  echo "== $n"; python3 - /tmp/pr_$n <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel_bf16" in r["Name"] or "nerf_backward_kernel_x3" in r["Name"]:
            print(r["Name"][:60], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1))
PY
done
