#!/bin/bash
# build/ab/lib_tl.so: the library with the phase timeline compiled in (-DMF_TIMELINE; tools/timeline.py reads it).
# Other whole-library variants: MF_VARIANT_FLAGS=-DMF_DBG_JITTER MF_VARIANT_NAME=jit tools/build_timeline.sh (race screen)
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/tlobj $REPO/build/ab
cd $REPO/moco_flow_amd/csrc
rm -f /tmp/tlobj/*.o
for f in $(make -s units); do        # the library's translation units (csrc/Makefile)
  X=$(make -s unitflags UNIT=$f)         # per-unit flags of the shipped build (csrc/Makefile)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off ${MF_VARIANT_FLAGS:--DMF_TIMELINE} $X "$@" -c $f.hip -o /tmp/tlobj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/tlobj/*.o -o $REPO/build/ab/lib_${MF_VARIANT_NAME:-tl}.so
ls -la $REPO/build/ab/lib_${MF_VARIANT_NAME:-tl}.so
