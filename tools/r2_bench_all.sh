#!/bin/bash
# GPU box: round-2 measurement set -> gpurun_out/r02/ (copied into profiles/ afterwards)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02; mkdir -p $O
B="python bench.py --steps 20 --warmup 5"
$B --profile-table > $O/bench.json 2> $O/bench_table.txt
$B --no-cpu-baseline --norm group > $O/bench_gn.json 2>/dev/null
$B --no-cpu-baseline --norm instance > $O/bench_in.json 2>/dev/null
$B --no-cpu-baseline --exp probunet > $O/bench_probunet.json 2>/dev/null
$B --no-cpu-baseline --workload generate --image-size 192 --nlabels 4 --steps 100 > $O/bench_generate192.json 2>/dev/null
$B --no-cpu-baseline --workload generate --image-size 192 --nlabels 4 --steps 100 --samples-per-image 0 --batch 16 > $O/bench_generate192_unshared.json 2>/dev/null
$B --no-cpu-baseline --workload generate --image-size 192 --nlabels 4 --steps 50 --batch 8 > $O/bench_generate192_8img.json 2>/dev/null
for f in $O/bench*.json; do echo "$f $(tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; c=d.get('cpu_baseline') or {}; print(round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms; frac', r.get('frac'), 'final_loss', d['config'].get('final_loss'), 'cpu', c.get('value'), c.get('first_step_loss'), c.get('gpu_first_step_loss_same_inputs'))")"; done
