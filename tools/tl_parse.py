#!/usr/bin/env python3
"""One line per block mode out of tools/tl_blocks.sh's output: the phase deltas of wave 0's second tile."""
import re, sys
txt = open(sys.argv[1]).read()
for part in txt.split('=== blocks=')[1:]:
    rows = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r'tag\s+(\d+)\s+t\s+[\d.]+\s+\+\s*([\d.]+)', part)]
    idx = [i for i, (t, _) in enumerate(rows) if t == 1]
    if len(idx) >= 3:
        seg = rows[idx[1]:idx[2] + 1]
        print('blocks', part[0], ' '.join(f"{t}:{int(d)}" for t, d in seg), ' total', int(sum(d for _, d in seg[1:])))
