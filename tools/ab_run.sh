#!/bin/bash
# Interleaved same-box timing of library variants (run ON THE GPU BOX): tools/ab_run.sh "<configs>" <name> ...
# name = "default" (the in-tree library) or build/ab/lib_<name>.so; prints kernel ms (graph replay), frac and step ms per
# (config, variant, repetition).
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CFGS=$1; shift
for rep in 1 2 3; do
  for cfg in $CFGS; do
    for name in "$@"; do
      L=$REPO/build/ab/lib_$name.so; [ $name = default ] && L=""
      MOCOFLOW_HIP_LIB=$L python3 $REPO/bench.py --config $cfg --steps 200 --warmup 100 --no-cpu-baseline --no-train-leg --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$cfg $name rep$rep kernel_ms %.4f frac %.3f step_ms %.4f' % (r['kernel_ms'], r['frac'], d['ms_per_step']))"
    done
  done
done
