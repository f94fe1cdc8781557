#!/bin/bash
run() {
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config']['launches_per_step'])"
}
for i in 1 2; do
run PHX_FBN_MAXP=0
run PHX_FBN_MAXP=4096
run PHX_FBN_MAXP=4096 PHX_FBN_MAXK=64
run PHX_FBN_MAXP=16384 PHX_FBN_MAXK=64
done
