#!/bin/bash
# Round 5: the committed profile set (run ON THE GPU BOX): kernel-trace stats + PMC passes per BASELINE config, summaries
# under gpurun_out/r05/ -- copied to profiles/ by hand afterwards together with traffic_entry.json of each.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for cfg in "$@"; do
  bash tools/profile_gpu.sh r05_$cfg --config $cfg > /dev/null 2>&1
  mkdir -p gpurun_out/r05
  cp gpurun_out/prof_r05_$cfg/summary.txt gpurun_out/r05/r05_${cfg}_summary.txt
  cp gpurun_out/prof_r05_$cfg/traffic_entry.json gpurun_out/r05/r05_${cfg}_traffic_entry.json 2>/dev/null
  f=$(find gpurun_out/prof_r05_$cfg/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r05/r05_${cfg}_kernel_stats.csv
  rm -rf gpurun_out/prof_r05_$cfg/pmc_* gpurun_out/prof_r05_$cfg/trace
  grep -E "dominant|dispatches|hipcc resource|traffic entry" gpurun_out/r05/r05_${cfg}_summary.txt | cut -c1-400
done
