#!/bin/bash
# Same-box A/B of the fast bf16 mode's two kernel families (run ON THE GPU BOX): MF_BF16_BLOCKS=1 (8 waves x 32 samples) vs 2
# (4 waves x 2 x 32 samples, mf_bf16_2b.hpp) -- outputs bitwise, then interleaved timing.  usage: tools/ab_blocks.sh "<configs>"
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $REPO
MF_BF16_BLOCKS=1 timeout 200 python3 tools/blocks_dump.py /tmp/blocks1.pt 2>&1 | tail -1
MF_BF16_BLOCKS=2 timeout 200 python3 tools/blocks_dump.py /tmp/blocks2.pt 2>&1 | tail -1
timeout 100 python3 tools/blocks_dump.py --compare /tmp/blocks1.pt /tmp/blocks2.pt 2>&1 | tail -12
for rep in 1 2 3; do
  for cfg in $1; do
    for nb in 1 2; do
      MF_BF16_BLOCKS=$nb timeout 200 python3 bench.py --config $cfg --steps 200 --warmup 100 --no-cpu-baseline --no-train-leg --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$cfg blocks=$nb rep$rep kernel_ms %.4f frac %.3f step_ms %.4f' % (r['kernel_ms'], r['frac'], d['ms_per_step']))"
    done
  done
done
