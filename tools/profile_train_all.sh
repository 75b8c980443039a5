#!/bin/bash
# The training-step profiles of a round (run ON THE GPU BOX; ROUND=r06 by default): joint MoCo stage and stage 1, default fp32 forward
# and the opt-in three-product forward -> gpurun_out/$ROUND/${ROUND}_train_<name>_summary.txt (+ _traffic_entry.json: HBM GB per step)
set -u
R=${ROUND:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO; mkdir -p gpurun_out/$R
one() {   # name, tool, rays, [env]
  env $4 MF_ONLY=step bash tools/profile_train.sh ${R}_$1 $2 $3 > /dev/null 2>&1
  cp gpurun_out/prof_${R}_$1/summary.txt gpurun_out/$R/${R}_train_$1_summary.txt
  cp gpurun_out/prof_${R}_$1/traffic_entry.json gpurun_out/$R/${R}_train_$1_traffic_entry.json 2>/dev/null
  rm -rf gpurun_out/prof_${R}_$1
}
one joint tools/time_moco_step.py 1024 MF_X=0
one stage1 tools/time_train_step.py 5120 MF_X=0
if [ "${OPTIN:-1}" = 1 ]; then
  one joint_optin tools/time_moco_step.py 1024 MF_TRAIN_FWD=bf16x3
  one stage1_optin tools/time_train_step.py 5120 MF_TRAIN_FWD=bf16x3
fi
for f in joint stage1 joint_optin stage1_optin; do [ -f gpurun_out/$R/${R}_train_${f}_summary.txt ] && { echo "== $f"; sed -n 3,12p gpurun_out/$R/${R}_train_${f}_summary.txt | cut -c1-110; grep "HBM traffic" gpurun_out/$R/${R}_train_${f}_summary.txt; }; done
