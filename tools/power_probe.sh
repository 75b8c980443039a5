#!/bin/bash
# sample power / clocks while a bench config loops
cfg=$1
python3 bench.py --config $cfg --steps ${2:-6000} --warmup 10 --no-cpu-baseline --no-train-leg --no-extra-legs > /tmp/b_$cfg.json 2>/dev/null &
BP=$!
sleep 9
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo; sleep 0.4; done
wait $BP
python3 -c "
import json; d=json.loads(open('/tmp/b_$cfg.json').read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
