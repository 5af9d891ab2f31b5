#!/bin/bash
# usage (GPU box): bash tools/pmc_fill.sh  -- SQ counters of the two format-fill kernels (each alone on the chip under --pmc), two passes
cd $GRAFT_REPO_ROOT
for k in atd_fill quad_fill_staged; do
  echo "== $k"
  bash tools/pmc_kernel.sh $k c2 "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
  bash tools/pmc_kernel.sh $k c2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
done
