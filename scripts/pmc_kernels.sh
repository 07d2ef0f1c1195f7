#!/bin/bash
# Matrix-core and LDS counters of the main kernels (run on the GPU box): bash scripts/pmc_kernels.sh <outdir>
# one counter set per pass (no trace domains beside --pmc)
set -e
out=$1; export TMPDIR=/tmp; mkdir -p gpurun_out/$out
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/$out/set$i -- python3 scripts/kernels_once.py > gpurun_out/$out/set$i.log 2>&1
done
echo done
