#!/bin/bash
run() {
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config']['launches_per_step'])"
}
run A=0
run PHX_PRIOR_BW_FIRST=1
run PHX_PRIOR_BW_FIRST=1 PHX_DEFER_EARLY=1
run PHX_PRIOR_BW_FIRST=1 PHX_DEFER_EARLY=1 PHX_POLITE_LDS=84000
run PHX_DEFER_EARLY=1 PHX_POLITE_LDS=84000
run A=0
