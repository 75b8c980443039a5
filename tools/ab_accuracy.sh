#!/bin/bash
# Accuracy + kernel time of library variants built by tools/ab_lib.sh (build/ab/lib_<name>.so), per bench config, with
# the CPU oracle comparison on.   usage: tools/ab_accuracy.sh "<name> <name> ..." ["<config> ..."]
NAMES=${1:?usage: ab_accuracy.sh "name ..." ["config ..."]}
CFGS=${2:-C2b C3 C5}
for l in $NAMES; do for c in $CFGS; do
MOCOFLOW_HIP_LIB=build/ab/lib_$l.so python3 bench.py --config $c --steps 20 --warmup 3 --no-train-leg --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; e=d.get('error_vs_cpu',{})
print('$l $c kernel_ms %.4f frac %.3f' % (r['kernel_ms'], r['frac']), 'psnr', e.get('psnr_equiv_db'), 'l2', {k: round(v,5) for k,v in e.get('l2_rel',{}).items()})"
done; done
