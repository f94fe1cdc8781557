#!/bin/bash
# GPU box: HBM traffic per kernel of the training step (FETCH_SIZE / WRITE_SIZE, separate passes) -> gpurun_out/quick/pmc_hbm_traffic.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/quick
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc/$c -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-other-workloads > $O/pmc_$c.log 2>&1
  f=$(find $O/pmc/$c -name "*counter_collection.csv" | head -1)
  cp "$f" $O/pmc/$c/p_counter_collection.csv
done
cd $R
python tools/pmc_summary.py $O/pmc 4 $O/pmc_hbm_traffic.txt $O/pmc_hbm_traffic.json > /dev/null
rm -rf $O/pmc
grep -i "wgrad\|TOTAL" $O/pmc_hbm_traffic.txt
