#!/bin/bash
# diagnostic variants of the fused Winograd kernel on two layers (what bounds it?), then PMC counters of the product
out=gpurun_out/$1; mkdir -p $out; rm -f $out/variants.txt $out/pmc.txt
for v in ${VARIANTS:-0 1 2 4 7}; do
  echo "== variant $v" >> $out/variants.txt
  CLX_FUSED_VARIANT=$v timeout 120 python tools/exp/fused_bench.py --only "infer l0.6" 2>&1 | grep -v amdgpu.ids | cut -c1-40,150-260 >> $out/variants.txt
  CLX_FUSED_VARIANT=$v timeout 120 python tools/exp/fused_bench.py --only "infer l1.6" 2>&1 | grep -v amdgpu.ids | cut -c1-40,150-260 >> $out/variants.txt
done
cat $out/variants.txt
[ -n "$NO_PMC" ] && exit 0
# PMC counters of the product kernel (separate passes, --kernel-trace only, each bounded)
export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -o t -- python3 tools/exp/fused_bench.py --only "infer l" --reps 2 > /dev/null 2> $out/err$i.txt
  python3 tools/pmc_digest.py $out/p$i "wino_fused" >> $out/pmc.txt 2>> $out/err$i.txt
  rm -rf $out/p$i
done
cut -c1-330 $out/pmc.txt
