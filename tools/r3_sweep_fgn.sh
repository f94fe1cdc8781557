#!/bin/bash
run() {
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads --norm group 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config']['launches_per_step'], d['config']['final_loss'])"
}
for i in 1 2; do
run PHX_FGN=0
run PHX_FGN=1
done
env PHX_FGN=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads --norm instance 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('instance FGN=1', round(d['value'],1), round(d['ms_per_step'],3), d['config']['launches_per_step'])"
env PHX_FGN=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads --norm instance 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('instance FGN=0', round(d['value'],1), round(d['ms_per_step'],3), d['config']['launches_per_step'])"
