#!/bin/bash
run() {
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config']['launches_per_step'])"
}
run A=0
for v in 128 384 512; do run PHX_BN32_MAXBLOCKS=$v; done
run A=0
for v in 32 96 128; do run PHX_FWD_SPLITK_BLOCKS=$v; done
run A=0
for v in 2 8; do run PHX_NREP=$v; done
run PHX_WGRAD_DEFER_BLOCKS=64
run PHX_WGRAD_DEFER_BLOCKS=128
run A=0
