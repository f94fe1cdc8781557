#!/bin/bash
# GPU box: same-box A/B of staged builds (_old_ab/: round-3 baseline) against the working tree
R=$GRAFT_REPO_ROOT
b() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config']['launches_per_step'])"; }
for d in _old_ab .; do
echo "== trace $d"; (cd $R/$d && python tools/trace_small.py 2>&1 | grep fwd | cut -c1-70)
done
echo "== trace . dense"; (cd $R && PHX_LDS_PAD=0 python tools/trace_small.py 2>&1 | grep fwd | cut -c1-70)
for i in 1 2; do
echo "== bench old"; (cd $R/_old_ab && b)
echo "== bench new"; (cd $R && b)
echo "== bench new dense"; (cd $R && PHX_LDS_PAD=0 b)
echo "== bench new, no dual no fbn"; (cd $R && PHX_DUAL=0 PHX_FBN_MAXP=0 b)
done
