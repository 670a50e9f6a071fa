#!/bin/bash
# What the chip sustains on the split-precision product's instruction mix alone (tools/exp/mfma_shapes_bf16.hip: six bf16
# MFMAs per fragment pair on operands held in registers, two waves per SIMD, every CU), on random and on constant operands,
# and the clock / matrix-pipe occupancy of those launches (one --pmc pass).
# Usage (GPU box): bash tools/mfma_ceiling.sh  -> gpurun_out/mfma_ceiling.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
B=tools/exp/mfma_shapes_bf16
[ -x $B ] || hipcc --offload-arch=gfx950 -O3 -o $B $B.hip
{
  $B
  $B const
  for mode in random const; do
    rm -rf $O/pmc_ceiling
    if [ $mode = const ]; then
      timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_ceiling -o t -- $B const > /dev/null 2>&1
    else
      timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_ceiling -o t -- $B > /dev/null 2>&1
    fi
    echo "counters, operands $mode (GRBM_GUI_ACTIVE sums the 8 XCDs; MHz = GRBM_GUI_ACTIVE / 8 / duration; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)):"
    python3 tools/pmc_digest.py $O/pmc_ceiling "loop"
    rm -rf $O/pmc_ceiling
  done
} > $O/mfma_ceiling.txt 2>&1
cat $O/mfma_ceiling.txt
