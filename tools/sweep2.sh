#!/bin/bash
# GPU box: bench.py under combinations "VAR1=a VAR2=b" given as quoted arguments
for combo in "$@"; do
  env $combo python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$combo', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"
done
