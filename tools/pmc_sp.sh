#!/bin/bash
# Issue / LDS / wait counters of the split-precision product kernels on one long-K shape (separate --pmc passes).
# Usage (GPU box): bash tools/pmc_sp.sh  -> gpurun_out/pmc_sp/*.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_sp
rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
export SP_TIME_ONLY=1
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o t -- python3 tools/bench_gemm_sp.py 123008 768 2304 > /dev/null 2> $O/err$i.txt
  python3 tools/pmc_digest.py $O/p$i "gemm_sp" >> $O/counters.txt 2>> $O/err$i.txt
  rm -rf $O/p$i
done
cat $O/counters.txt
