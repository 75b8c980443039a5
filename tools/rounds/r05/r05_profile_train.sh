#!/bin/bash
# Round 5: the training-step profiles (run ON THE GPU BOX): joint MoCo stage and stage 1, default fp32 forward and the opt-in
# three-product forward -> gpurun_out/r05/r05_train_<name>_summary.txt
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO; mkdir -p gpurun_out/r05
MF_ONLY=step bash tools/profile_train.sh r05_joint tools/time_moco_step.py 1024 > /dev/null 2>&1;  cp gpurun_out/prof_r05_joint/summary.txt gpurun_out/r05/r05_train_joint_summary.txt
MF_TRAIN_FWD=bf16x3 MF_ONLY=step bash tools/profile_train.sh r05_joint_optin tools/time_moco_step.py 1024 > /dev/null 2>&1; cp gpurun_out/prof_r05_joint_optin/summary.txt gpurun_out/r05/r05_train_joint_optin_summary.txt
MF_ONLY=step bash tools/profile_train.sh r05_stage1 tools/time_train_step.py 5120 > /dev/null 2>&1; cp gpurun_out/prof_r05_stage1/summary.txt gpurun_out/r05/r05_train_stage1_summary.txt
MF_TRAIN_FWD=bf16x3 MF_ONLY=step bash tools/profile_train.sh r05_stage1_optin tools/time_train_step.py 5120 > /dev/null 2>&1; cp gpurun_out/prof_r05_stage1_optin/summary.txt gpurun_out/r05/r05_train_stage1_optin_summary.txt
rm -rf gpurun_out/prof_r05_joint* gpurun_out/prof_r05_stage1*
for f in joint joint_optin stage1 stage1_optin; do echo "== $f"; sed -n 3,12p gpurun_out/r05/r05_train_${f}_summary.txt | cut -c1-110; done
