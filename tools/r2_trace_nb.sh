R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_nb; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/stats -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/stats.log 2>&1
cd $R; db=$(find $O/stats -name "*results.db" | head -1); python tools/trace_neighbors.py $db copyBuffer; rm -rf $O/stats
