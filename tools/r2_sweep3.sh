#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 60 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
echo -n "default "; run
for kv in "PHX_NREP=2" "PHX_NREP=4" "PHX_NREP=4 PHX_FWD_SPLITK_BLOCKS=32" "PHX_NREP=4 PHX_NORM_CAP=256" "PHX_NREP=4 PHX_NORM_CAP=1024" "PHX_NREP=1" "PHX_STREAM_FLOOR=512" "PHX_STREAM_FLOOR=2048" "PHX_STREAM_PPT=16" "PHX_STREAM_PPT=4" "PHX_NORM_PPT=8" "PHX_NORM_PPT=32" "PHX_STATS_FLOOR=64" "PHX_STATS_FLOOR=256" "PHX_FWD_N32=0" "PHX_NREP=4"; do
  echo -n "$kv "; env $kv bash -c "$(declare -f run); run"
done
echo -n "default "; run
