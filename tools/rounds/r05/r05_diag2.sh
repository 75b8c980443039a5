#!/bin/bash
# timeline + MFMA-busy / clock counters of one library variant: tools/r05_diag2.sh <tag> <cfgs...>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$REPO/gpurun_out/r05_diag2_$TAG
mkdir -p $OUT
cd $REPO
for c in "$@"; do
  MOCOFLOW_HIP_LIB=$REPO/build/ab/lib_tl.so timeout 300 python3 tools/timeline.py $c > $OUT/timeline_$c.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  for L in base default; do
    LIB=$REPO/build/ab/lib_$L.so; [ $L = default ] && LIB=""
    MOCOFLOW_HIP_LIB=$LIB timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_${c}_$L -- python3 $REPO/bench.py --config $c --steps 30 --warmup 10 --no-cpu-baseline --no-train-leg --no-extra-legs > $OUT/pmc_${c}_$L.log 2>&1
    MOCOFLOW_HIP_LIB=$LIB timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr_${c}_$L -- python3 $REPO/bench.py --config $c --steps 30 --warmup 10 --no-cpu-baseline --no-train-leg --no-extra-legs > $OUT/tr_${c}_$L.log 2>&1
    python3 - $OUT $c $L <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, c, L = sys.argv[1:4]
tot = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, f"pmc_{c}_{L}/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, f"tr_{c}_{L}/**/*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
dom = max(dur, key=lambda k: sum(dur[k]))
cn = {k: sum(v) / len(v) for k, v in tot[dom].items()}
d = dur[dom]
avg = sum(d) / len(d) / 1e3
cyc = cn["GRBM_GUI_ACTIVE"] / 8
print(f"{c} {L}: {dom[:60]} avg {avg:.1f} us min {min(d)/1e3:.1f}  cycles/XCD {cyc:.4g}  MFMA busy {cn['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f}  "
      f"GHz(pmc run) {cyc / (avg * 1e3):.3f}  wait_inst/wave_cycles {cn['SQ_WAIT_INST_ANY'] / cn['SQ_WAVE_CYCLES']:.3f}  VALU insts {cn['SQ_INSTS_VALU']:.4g}")
PY
  done
done
