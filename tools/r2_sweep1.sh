#!/bin/bash
# round-2 first GPU call: baseline on this box + cheap policy sweeps
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
$B --profile-table > gpurun_out/s1/base.json 2> gpurun_out/s1/base.err
$B --no-roofline > gpurun_out/s1/base2.json 2>/dev/null
PHX_FUSE_BWS=1 $B --no-roofline > gpurun_out/s1/bws_all.json 2>/dev/null
PHX_FUSE_BWS=1 PHX_FUSE_BWS_MAXP=70000 $B --no-roofline > gpurun_out/s1/bws_70k.json 2>/dev/null
PHX_FUSE_BWS=1 PHX_FUSE_BWS_MAXP=300000 $B --no-roofline > gpurun_out/s1/bws_300k.json 2>/dev/null
PHX_BN_SMALL=4096 $B --no-roofline > gpurun_out/s1/bnsmall4096.json 2>/dev/null
PHX_BN_SPLITK=1 $B --no-roofline > gpurun_out/s1/bnsplitk.json 2>/dev/null
$B --no-roofline --workload generate --image-size 192 --nlabels 4 --batch 16 --steps 50 > gpurun_out/s1/gen192.json 2>/dev/null
$B --no-roofline --exp probunet > gpurun_out/s1/probunet.json 2>/dev/null
for f in gpurun_out/s1/*.json; do echo "$f $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['config'].get('final_loss'))")"; done
