#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export MF_ONLY=${MF_ONLY:-hipbwd}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_train -- python3 ${MF_TOOL:-$REPO/tools/time_train_step.py} ${1:-5120} > $REPO/gpurun_out/prof_train.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$REPO/gpurun_out/prof_train/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:30]:
        print(r["Name"][:110].ljust(110), r["Calls"].rjust(6), f'{float(r["TotalDurationNs"])/1e6:9.1f} ms', r["Percentage"])
PY
tail -6 $REPO/gpurun_out/prof_train.log
