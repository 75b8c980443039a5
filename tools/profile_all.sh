#!/bin/bash
# The committed profile set of a round (ROUND=r06 by default) (run ON THE GPU BOX): kernel-trace stats + PMC passes per BASELINE config, summaries
# under gpurun_out/${ROUND:-r06}/ -- copied to profiles/ by hand afterwards together with traffic_entry.json of each.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for cfg in "$@"; do
  bash tools/profile_gpu.sh ${ROUND:-r06}_$cfg --config $cfg > /dev/null 2>&1
  mkdir -p gpurun_out/${ROUND:-r06}
  cp gpurun_out/prof_${ROUND:-r06}_$cfg/summary.txt gpurun_out/${ROUND:-r06}/${ROUND:-r06}_${cfg}_summary.txt
  cp gpurun_out/prof_${ROUND:-r06}_$cfg/traffic_entry.json gpurun_out/${ROUND:-r06}/${ROUND:-r06}_${cfg}_traffic_entry.json 2>/dev/null
  f=$(find gpurun_out/prof_${ROUND:-r06}_$cfg/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${ROUND:-r06}/${ROUND:-r06}_${cfg}_kernel_stats.csv
  rm -rf gpurun_out/prof_${ROUND:-r06}_$cfg/pmc_* gpurun_out/prof_${ROUND:-r06}_$cfg/trace
  grep -E "dominant|dispatches|hipcc resource|traffic entry" gpurun_out/${ROUND:-r06}/${ROUND:-r06}_${cfg}_summary.txt | cut -c1-400
done
