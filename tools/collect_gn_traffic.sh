#!/bin/bash
# GPU box: HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes) and kernel statistics of the GROUP-NORM training step
# (BASELINE config 2 as named) -> gpurun_out/final_gn/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_gn
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--norm group --no-cpu-baseline --no-roofline --no-other-workloads"
rocprofv3 --kernel-trace --stats -d $O/stats -- python $R/bench.py --steps 8 --warmup 3 $A > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/$c -- python $R/bench.py --steps 2 --warmup 2 $A > $O/pmc_$c.log 2>&1
  f=$(find $O/pmc/$c -name "*counter_collection.csv" | head -1)
  cp "$f" $O/pmc/$c/p_counter_collection.csv
done
cd $R
db=$(find $O/stats -name "*results.db" | head -1)
python tools/prof_summary.py $db 11 40 > $O/kernel_stats.txt
python tools/pmc_summary.py $O/pmc 4 $O/pmc_hbm_traffic.txt $O/pmc_hbm_traffic.json > /dev/null
sed -i 's/phiseg_7_5 128x128 bf16 B=64, per training step/phiseg_7_5 128x128 bf16 B=64 GROUP NORM (bench.py --norm group), per training step/' $O/pmc_hbm_traffic.txt
rm -rf $O/stats $O/pmc/*/runc $O/pmc/*/*/ 2>/dev/null
tail -1 $O/stats.log | cut -c1-200; head -12 $O/kernel_stats.txt | cut -c1-160; tail -1 $O/pmc_hbm_traffic.txt
