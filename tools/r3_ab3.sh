#!/bin/bash
# GPU box: same-box A/B of a staged build (_old_ab/) against the working tree; optional argument: an environment setting for a third leg
R=$GRAFT_REPO_ROOT
b() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config']['launches_per_step'])"; }
for i in 1 2 3; do
echo "== bench old"; (cd $R/_old_ab && b)
echo "== bench new"; (cd $R && b)
if [ -n "$1" ]; then echo "== bench new, $1"; (cd $R && export $1 && b); fi
done
