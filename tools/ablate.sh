#!/bin/bash
# timing ablations of the render kernel (results are WRONG with flags != 0; timing only).
# Needs an ablation build:  bash tools/build_ablate.sh  then  MOCOFLOW_HIP_LIB=moco_flow_amd/libmocoflow_flags.so
for f in ${FLAGS:-0 1 2 3 4 8 15}; do
  echo -n "MF_DEBUG_FLAGS=$f: "
  MF_DEBUG_FLAGS=$f python bench.py --no-cpu-baseline --steps 30 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))"
done
