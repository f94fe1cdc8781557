#!/bin/bash
# GPU box: the evidence behind bench.py's numbers -> gpurun_out/final/ (copy the summaries into profiles/ afterwards)
#   PHX_COMMIT=$(git rev-parse --short HEAD) in the gpurun command line: stamped into the summaries (the box has no .git)
#   1. bench.py (default flags) JSON line + per-layer conv table
#   2. rocprofv3 --kernel-trace --stats of the same command (rocpd database -> tools/prof_summary.py)
#   3. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, kernel trace only (-> tools/pmc_summary.py)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
[ -n "$SKIP_BENCH" ] || python bench.py --profile-table > $O/bench.json 2> $O/bench_table.txt
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
# MFMA / LDS utilisation of the convolution kernels (one more PMC pass, kernel trace only)
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc/MFMA -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-other-workloads > $O/pmc_MFMA.log 2>&1
cd $R
f=$(find $O/pmc/MFMA -name "*counter_collection.csv" | head -1)
python tools/pmc_mfma_summary.py "$f" 4 $O/pmc_mfma_lds_util.txt > /dev/null 2>&1 || true
rm -rf $O/stats $O/pmc/*/runc $O/pmc/*/*/ 2>/dev/null
# the fp32 parity path (csrc/conv_f32_mfma.hip): kernel statistics of the same step at --dtype f32, and the matrix pipe's busy fraction
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats_f32 -- python $R/bench.py --dtype f32 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-other-workloads > $O/stats_f32.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc/MFMA_f32 -- python $R/bench.py --dtype f32 --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-other-workloads > $O/pmc_MFMA_f32.log 2>&1
cd $R
db=$(find $O/stats_f32 -name "*results.db" | head -1)
python tools/prof_summary.py $db 8 30 > $O/kernel_stats_f32.txt
f=$(find $O/pmc/MFMA_f32 -name "*counter_collection.csv" | head -1)
python tools/pmc_mfma_summary.py "$f" 4 $O/pmc_mfma_lds_util_f32.txt > /dev/null 2>&1 || true
rm -rf $O/stats_f32 $O/pmc/*/runc $O/pmc/*/*/ 2>/dev/null
sed -i "1i # commit ${PHX_COMMIT:-?}" $O/kernel_stats.txt $O/kernel_stats_f32.txt $O/pmc_mfma_lds_util.txt $O/pmc_mfma_lds_util_f32.txt 2>/dev/null
head -5 $O/kernel_stats.txt; head -8 $O/kernel_stats_f32.txt; tail -1 $O/pmc_hbm_traffic.txt; tail -c 400 $O/bench.json
