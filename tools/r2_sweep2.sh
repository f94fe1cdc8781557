#!/bin/bash
# GPU box: one-knob-at-a-time sweep of the engine's scheduling hooks around the current defaults (ms per step, 60 steps)
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --steps 60 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
echo -n "default "; run
for kv in PHX_BN_SPLITK=1 PHX_BN_SMALL=4096 PHX_BN_SMALL=256 PHX_NREP=4 PHX_NREP=16 PHX_WGRAD_DEFER_BLOCKS=64 PHX_WGRAD_DEFER_BLOCKS=128 PHX_WGRAD_DEFER_BLOCKS=192 PHX_EARLY_TOUCH=0 PHX_LANES=3 PHX_FWD_SPLITK_BLOCKS=128 PHX_FWD_SPLITK_BLOCKS=32 PHX_WGRAD_BLOCKS=256 PHX_WGRAD_BLOCKS=768 PHX_NREP_MINP=16384 PHX_WGRAD_DEFER_TILES=4096 PHX_WGRAD_DEFER_TILES=256; do
  echo -n "$kv "; env $kv bash -c "$(declare -f run); run"
done
echo -n "default "; run
