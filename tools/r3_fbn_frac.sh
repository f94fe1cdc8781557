#!/bin/bash
# GPU box: headline value and convolution-family fraction with / without the one-launch conv + batch-norm layers
for v in 4096 0 4096 0; do
PHX_FBN_MAXP=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('FBN_MAXP=$v', round(d['value'],1), round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'conv ms', round(sum(v['ms_per_step'] for k,v in r['families'].items() if 'wgrad' not in k),3), d['config']['launches_per_step'])"
done
