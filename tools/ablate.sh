#!/bin/bash
# timing ablations of the fp32 render kernel (results are WRONG with flags != 0; timing only).
# Needs an ablation build:  bash tools/build_ablate.sh  then  MOCOFLOW_HIP_LIB=moco_flow_amd/libmocoflow_flags.so
# flags: 1 no barrier, 2 no LDS-DMA, 4 no sincos, 8 no composite, 64 no half-panel stagger
for rep in 1 2; do
for f in ${FLAGS:-0 1 2 3 4 8 15}; do
  echo -n "MF_DEBUG_FLAGS=$f: "
  MF_DEBUG_FLAGS=$f python bench.py --no-cpu-baseline --no-train-leg --no-extra-legs --steps 30 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('kernel ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))"
done; done
