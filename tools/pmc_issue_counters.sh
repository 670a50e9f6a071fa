#!/bin/bash
# LDS / issue counters of the MFMA kernels of one 2-D train step (separate --pmc passes, --kernel-trace only).
# Usage (on the GPU box): [CLX_PMC_WORKLOAD=train3d] bash tools/pmc_issue_counters.sh   -> gpurun_out/pmc_f32[_train3d]/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_f32${CLX_PMC_WORKLOAD:+_$CLX_PMC_WORKLOAD}
rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o t -- python3 bench.py --workload ${CLX_PMC_WORKLOAD:-train2d} --steps 1 --warmup 1 --no-infer --no-cpu-baseline --no-train3d > /dev/null 2> $O/err$i.txt
  python3 tools/pmc_digest.py $O/p$i "conv_" > $O/set${i}_f32.txt 2>> $O/err$i.txt; python3 tools/pmc_digest.py $O/p$i "wino_" > $O/set${i}_wino.txt 2>> $O/err$i.txt
  rm -rf $O/p$i
done
tail -n 3 $O/err*.txt
