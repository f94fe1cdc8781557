#!/bin/bash
# GPU box: rocprofv3 kernel statistics of the default training step -> gpurun_out/stats_only/kernel_stats.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/stats_only
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads "$@" > $O/stats.log 2>&1
cd $R
db=$(find $O/stats -name "*results.db" | head -1)
python tools/prof_summary.py $db 11 > $O/kernel_stats.txt
rm -rf $O/stats
head -48 $O/kernel_stats.txt | cut -c1-175
