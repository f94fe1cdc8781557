#!/bin/bash
run() {
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config']['launches_per_step'], d['config']['final_loss'])"
}
for i in 1 2; do
run PHX_BN_MID_MAXP=0
run PHX_BN_MID_MAXP=4096
run PHX_BN_MID_MAXP=16384
run PHX_BN_MID_MAXP=65536
done
