#!/bin/bash
# GPU box: rocprofv3 kernel statistics of the training step under another normalisation (--norm group|instance) -> gpurun_out/norm_$1/
N=${1:-group}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/norm_$N
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --norm $N --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > $O/stats.log 2>&1
cd $R
db=$(find $O/stats -name "*results.db" | head -1)
python tools/prof_summary.py $db 11 > $O/kernel_stats.txt
rm -rf $O/stats
head -40 $O/kernel_stats.txt | cut -c1-190
