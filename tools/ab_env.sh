#!/bin/bash
# GPU box, same-box A/B of one environment switch: bench.py (40 steps) per setting, base first and last.  usage: ab_env.sh NAME V1 [V2 ...]
R=$GRAFT_REPO_ROOT
n=$1; shift
run() { timeout 200 python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-other-workloads 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s  %.3f ms' % (d['value'], d['ms_per_step']))"; }
echo "base: $(run)"
for v in "$@"; do echo "$n=$v: $(env $n=$v bash -c "$(declare -f run); R=$R; run")"; done
echo "base: $(run)"
