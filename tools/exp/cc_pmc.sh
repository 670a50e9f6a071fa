#!/bin/bash
# issue / LDS counters of the connected-component kernels (separate --pmc passes)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/cc_pmc
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# the pixel-list reference the variants compare with: written here, OUTSIDE the profiler (see cc_variants.py)
CLX_CC_LABELS=0 python3 $R/tools/exp/cc_variants.py --dump /tmp/cc_ref_${CC_N:-4096}.npy
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o t -- python3 $R/tools/exp/cc_variants.py > /dev/null 2> $O/err$i.txt
  python3 $R/tools/pmc_digest.py $O/p$i "cc_" > $O/set$i.txt 2>> $O/err$i.txt
  rm -rf $O/p$i
done
cat $O/set*.txt | cut -c1-220
tail -n 2 $O/err*.txt | cut -c1-200
