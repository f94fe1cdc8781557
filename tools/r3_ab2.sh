#!/bin/bash
R=$GRAFT_REPO_ROOT
b() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config']['launches_per_step'])"; }
for d in _old_ab .; do
echo "== trace $d"; (cd $R/$d && python tools/trace_small.py 2>&1 | grep fwd | cut -c1-70)
done
for i in 1 2; do
echo "== bench old"; (cd $R/_old_ab && b)
echo "== bench new"; (cd $R && b)
echo "== bench new, db kernel"; (cd $R && PHX_FWD_DB=1 b)
done
cd $R; BENCH_SHAPES=short python tools/bench_fwd_db.py 2>&1 | tail -6
