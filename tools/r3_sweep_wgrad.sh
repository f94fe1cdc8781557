#!/bin/bash
run() {
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config']['launches_per_step'])"
}
run A=0
for b in 16 24 32 48 64 128 192; do run PHX_WGRAD_DEFER_BLOCKS=$b; done
run A=0
run PHX_WGRAD_DEFER_TILES=4096
run PHX_WGRAD_DEFER_TILES=256
run PHX_NREP=2
run PHX_NREP=8
run A=0
