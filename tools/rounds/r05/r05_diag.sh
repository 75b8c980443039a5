#!/bin/bash
# Round 5, diagnosis pass of the bf16-pipe render kernels (run ON THE GPU BOX through gpurun):
#   1. tools/proto/duty_power: what the chip gives an MFMA-dense wave program with REAL operands, with injected idle time and
#      with the fragment stream from LDS -- is the pass issue-bound or limiter-bound?
#   2. the phase timeline (MF_TIMELINE build) of C3 and C3x,
#   3. instruction mix / wave-state counters of C3 and C3x (tools/profile_mix.sh).
# -> gpurun_out/r05_diag/
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05_diag
mkdir -p $OUT
cd $REPO
timeout 300 build/proto/duty_power > $OUT/duty_power.txt 2>&1
for c in C3 C3x C2b; do
  MOCOFLOW_HIP_LIB=$REPO/build/ab/lib_tl.so timeout 300 python3 tools/timeline.py $c > $OUT/timeline_$c.txt 2>&1
done
bash tools/profile_mix.sh r05_c3 --config C3 > $OUT/mix_c3.log 2>&1
bash tools/profile_mix.sh r05_c3x --config C3x > $OUT/mix_c3x.log 2>&1
cp $REPO/gpurun_out/mix_r05_c3/summary.txt $OUT/mix_c3_summary.txt
cp $REPO/gpurun_out/mix_r05_c3x/summary.txt $OUT/mix_c3x_summary.txt
tail -40 $OUT/duty_power.txt
