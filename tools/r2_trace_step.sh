#!/bin/bash
# GPU box: kernel trace of a few bench steps -> per-kernel summary (durations inside the replayed graph)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_$1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
shift
env "$@" rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline $BENCH_ARGS > $O/stats.log 2>&1
cd $R
db=$(find $O/stats -name "*results.db" | head -1)
python tools/prof_summary.py $db 11 60 > $O/kernel_stats.txt
rm -rf $O/stats
head -30 $O/kernel_stats.txt
