#!/bin/bash
# GPU box: deferred filter gradients beside the latency-bound backward chains, with "polite" occupancy (PHX_POLITE_LDS)
run() {
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config']['launches_per_step'])"
}
run A=0
run PHX_DEFER_LANE2=1 GPU_MAX_HW_QUEUES=4
run PHX_DEFER_LANE2=1 GPU_MAX_HW_QUEUES=4 PHX_POLITE_LDS=84000
run PHX_DEFER_LANE2=1 GPU_MAX_HW_QUEUES=3 PHX_POLITE_LDS=84000
run PHX_DEFER_LANE2=1 GPU_MAX_HW_QUEUES=8 PHX_POLITE_LDS=84000
run PHX_DEFER_EARLY=1 PHX_POLITE_LDS=84000
run PHX_DEFER_EARLY=1
run PHX_DEFER_LANE2=1 GPU_MAX_HW_QUEUES=4 PHX_POLITE_LDS=84000 PHX_WGRAD_DEFER_BLOCKS=48
run A=1
