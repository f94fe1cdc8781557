#!/bin/bash
# GPU box: MFMA / LDS utilisation counters of the bench step -> gpurun_out/final/pmc_mfma_util.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_mfma -- env PHX_LANES=1 python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $O/pmc_mfma.log 2>&1
cd $R
f=$(find $O/pmc_mfma -name "*counter_collection.csv" | head -1)
python tools/pmc_mfma_summary.py "$f" 4 $O/pmc_mfma_util.txt
rm -rf $O/pmc_mfma
