cd /tmp && export TMPDIR=/tmp
for n in default grow; do L=$GRAFT_REPO_ROOT/build/ab/lib_$n.so; [ $n = default ] && L=""
  rm -rf /tmp/pg_$n; MOCOFLOW_HIP_LIB=$L MF_ONLY=step timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg_$n -- python3 $GRAFT_REPO_ROOT/tools/time_train_step.py 5120 > /tmp/pg_$n.log 2>&1
  echo "== $n"; python3 - /tmp/pg_$n <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nerf_backward_kernel_x3" in r["Name"] or "wgrad_kernel<true>" in r["Name"]:
            print(r["Name"][:60], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1))
PY
done
