#!/bin/bash
# GPU box, same box A/B: bench.py (40 steps) with one engine.py tuning constant changed at a time.  usage: const_sweep.sh NAME=V [NAME=V ...]
R=$GRAFT_REPO_ROOT
cp $R/phiseg_code_amd/engine.py /tmp/engine.py.orig
run() { timeout 200 python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-other-workloads 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f img/s  %.3f ms' % (d['value'], d['ms_per_step']))"; }
echo "base: $(run)"
for kv in "$@"; do
  n=${kv%%=*}; v=${kv#*=}
  cp /tmp/engine.py.orig $R/phiseg_code_amd/engine.py
  sed -i "s/^$n = [0-9]*/$n = $v/" $R/phiseg_code_amd/engine.py
  echo "$kv: $(run)"
done
cp /tmp/engine.py.orig $R/phiseg_code_amd/engine.py
echo "base: $(run)"
