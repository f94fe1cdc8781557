#!/bin/bash
# GPU box: PMC passes over the standalone MFMA conv micro-benchmark (tools/bench_wgrad.py fwd|wgrad)
which=${1:-fwd}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$which
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/avail.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python tools/bench_wgrad.py $which > $out/p$i.log 2>&1)
done
cd $GRAFT_REPO_ROOT
python tools/pmc_conv_summary.py $out
