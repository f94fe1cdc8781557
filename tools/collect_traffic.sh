R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/bands
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-other-workloads > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/$c -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-other-workloads > $O/pmc_$c.log 2>&1
  f=$(find $O/pmc/$c -name "*counter_collection.csv" | head -1)
  cp "$f" $O/pmc/$c/p_counter_collection.csv
done
cd $R
db=$(find $O/stats -name "*results.db" | head -1)
python tools/prof_summary.py $db 11 40 $O/conv_in_situ.json > $O/kernel_stats.txt
python tools/pmc_summary.py $O/pmc 4 $O/pmc_hbm_traffic.txt $O/pmc_hbm_traffic.json > /dev/null
rm -rf $O/stats $O/pmc/*/runc $O/pmc/*/*/ $O/pmc 2>/dev/null
head -3 $O/kernel_stats.txt; tail -1 $O/pmc_hbm_traffic.txt
