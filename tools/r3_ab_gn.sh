#!/bin/bash
# GPU box: same-box A/B of a staged build (_old_ab/) against the working tree on the group-norm workload (extra bench.py flags: "$@")
R=$GRAFT_REPO_ROOT
b() { python bench.py --norm group "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config']['launches_per_step'])"; }
for i in 1 2 3; do
echo "== GN old"; (cd $R/_old_ab && b "$@")
echo "== GN new"; (cd $R && b "$@")
done
