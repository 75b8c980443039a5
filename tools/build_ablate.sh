#!/bin/bash
# Timing-ablation builds of the library (results are WRONG with flags != 0, durations are real):
#   libmocoflow_flags.so  : -DMF_TIMING_FLAGS=1, the kernels honour MF_DEBUG_FLAGS (production compiles them out)
#   libmocoflow_ablate.so : the same + -DMF_ABLATE_NOLDS (fragment ds_reads of the fp32 core removed)
# Use with  MOCOFLOW_HIP_LIB=moco_flow_amd/libmocoflow_flags.so FLAGS="0 1 2 3" bash tools/ablate.sh
# (profiles/README.md, "Where the last 15 % ... goes").
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO/moco_flow_amd/csrc
for variant in flags ablate; do
  DEFS="-DMF_TIMING_FLAGS=1"; [ $variant = ablate ] && DEFS="$DEFS -DMF_ABLATE_NOLDS=1"
  mkdir -p /tmp/abl_$variant
  for f in *.hip; do
    X=$(make -s unitflags UNIT=$f)       # per-unit flags of the shipped build (csrc/Makefile)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $DEFS $X -c $f -o /tmp/abl_$variant/${f%.hip}.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/abl_$variant/*.o -o ../libmocoflow_$variant.so
  echo built ../libmocoflow_$variant.so
done
