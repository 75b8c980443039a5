#!/bin/bash
# Timing-ablation build of the library (results are WRONG, durations are real): fragment ds_reads of the
# fp32 core removed (-DMF_ABLATE_NOLDS).  Use with  MOCOFLOW_HIP_LIB=moco_flow_amd/libmocoflow_ablate.so
# FLAGS="0 1 2 3" bash tools/ablate.sh   (profiles/README.md, "Where the last 15 % ... goes").
set -e
cd "$(dirname "$0")/../moco_flow_amd/csrc"
OBJS=""
for f in mf_abi mf_pack mf_forward mf_render mf_backward mf_wgrad mf_nofgrad mf_composite mf_sample mf_aux; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMF_ABLATE_NOLDS=1 -c $f.hip -o /tmp/abl_$f.o
  OBJS="$OBJS /tmp/abl_$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ../libmocoflow_ablate.so
echo built ../libmocoflow_ablate.so
