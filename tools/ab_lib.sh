#!/bin/bash
# A/B of kernel build variants.  Step 1 (here, CPU): build/ab/lib_<name>.so per "<name>=<flags>" spec, recompiling only
# the translation unit under test and linking the other objects of the in-tree build.  Step 2 (GPU box, through
# gpurun): tools/ab_lib.sh run "<bench args>" name...   -> interleaved timing of each variant on the same box.
# usage: tools/ab_lib.sh build <unit.hip> "<name>=<flags>" ...
#        tools/ab_lib.sh run "<bench args>" <name> ...
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
AB=$REPO/build/ab
mkdir -p $AB
mode=$1; shift
if [ "$mode" = build ]; then
  unit=$1; shift
  make -C $REPO/moco_flow_amd/csrc -j8 > /dev/null || exit 1
  others=$(ls $REPO/moco_flow_amd/csrc/*.o | grep -v "/${unit%.hip}.o")
  # (the variant carries the unit's own flags of the shipped build -- csrc/Makefile, `unitflags` -- or it measures them too)
  unitflags=$(make -s -C $REPO/moco_flow_amd/csrc unitflags UNIT=$unit)
  for spec in "$@"; do
    name=${spec%%=*}; flags="${spec#*=} $unitflags"
    ( cd $REPO/moco_flow_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $flags -c $unit -o $AB/${unit%.hip}_$name.o \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $AB/${unit%.hip}_$name.o -o $AB/lib_$name.so ) || echo "build $name FAILED" &
  done
  wait
  ls -la $AB/*.so
else
  BARGS=$1; shift
  for rep in 1 2 3; do
    for name in "$@"; do
      L=$AB/lib_$name.so; [ $name = default ] && L=""       # "default": the in-tree library
      MOCOFLOW_HIP_LIB=$L python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-leg --no-extra-legs $BARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$name rep$rep kernel_ms %.4f frac %.3f step %.3f' % (r['kernel_ms'], r['frac'], d['ms_per_step']))"
    done
  done
fi
