#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + PMC passes of one training-step tool.
# usage: tools/profile_train.sh <tag> <tool.py> <n_rays>     -> gpurun_out/prof_<tag>/summary.txt
set -u
TAG=${1:-train}; TOOL=${2:-tools/time_train_step.py}; N=${3:-5120}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MF_ONLY=${MF_ONLY:-step}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/$TOOL $N > $OUT/trace.log 2>&1
# PMC passes: own runs, kernel-trace only (never with sys/hip/hsa tracing)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $REPO/$TOOL $N > $OUT/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/$TOOL $N > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/$TOOL $N > $OUT/pmc_write.log 2>&1
python3 $REPO/tools/summarize_train.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
