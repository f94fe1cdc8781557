#!/bin/bash
# GPU box: HBM / L2 counters of the forward conv micro-benchmark for the default kernel, dma128 (ws=5) and ping-pong
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_modes
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && BENCH_SHAPES=short rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python tools/bench_fwd_modes.py 0 5 pp > $out/p$i.log 2>&1)
done
cd $GRAFT_REPO_ROOT
python tools/pmc_conv_summary.py $out
