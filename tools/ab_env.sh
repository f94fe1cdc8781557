#!/bin/bash
# GPU box, same-box A/B of environment switches: bench.py (40 steps) per NAME=V setting, base first and last.  usage: ab_env.sh NAME=V [NAME=V ...]
R=$GRAFT_REPO_ROOT
run() { timeout 200 python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-other-workloads 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s  %.3f ms' % (d['value'], d['ms_per_step']))" 2>&1 | tail -1; }
echo "base: $(run)"
for kv in "$@"; do echo "$kv: $(export $kv; run)"; done
echo "base: $(run)"
