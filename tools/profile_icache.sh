#!/bin/bash
# Run ON THE GPU BOX: instruction-cache counters of the dominant kernel.   usage: tools/profile_icache.sh <tag> [bench args]
set -u
TAG=${1:-ic}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-extra-legs $*"
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py $ARGS > $OUT/p1.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
tot = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dom = max(tot, key=lambda k: sum(tot[k].get("SQ_WAVE_CYCLES", [0])))
print("kernel:", dom[:100])
for k, v in sorted(tot[dom].items()):
    print(f"{k:28s} {sum(v)/len(v):.6g}")
PY
