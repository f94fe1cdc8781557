#!/bin/bash
# GPU box: bench.py under several values of one environment variable:  tools/sweep_env.sh PHX_LANES 1 4 6 8
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$var=$v', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"
done
