#!/bin/bash
# Device ISA of every translation unit of the library (the unit's own flags of the shipped build), comments and
# directives that carry paths / line numbers stripped -> <outdir>/<unit>.s ; `diff -r` of two snapshots says whether an edit
# changed the generated code (round 5: pruning the experiment switches must not).  usage: tools/isa_snapshot.sh <outdir> [unit ...]
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$1; shift
mkdir -p $OUT
cd $REPO/moco_flow_amd/csrc
UNITS=${@:-$(make -s units)}
n=0
for f in $UNITS; do
  X=$(make -s unitflags UNIT=$f)
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --cuda-device-only -S $X $f.hip -o - 2>/dev/null \
      | grep -v -E '^\s*(;|\.file|\.loc|\.ident|\.section\s+\.debug|\.Lfunc|\.cfi)' | sed -E 's/;.*$//' > $OUT/$f.s ) &
  n=$((n+1)); if [ $((n % 4)) -eq 0 ]; then wait; fi
done
wait
wc -l $OUT/*.s | tail -1
