export TMPDIR=/tmp SP_TIME_ONLY=1
O=gpurun_out/pmc16; rm -rf $O; mkdir -p $O
for m in 16 32; do
  export CLX_SP_MFMA=$m
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/p$m -o t -- python3 tools/bench_gemm_sp.py 123008 768 2304 > /dev/null 2> $O/err$m.txt
  echo "shape $m"; python3 tools/pmc_digest.py $O/p$m "gemm_sp"
  timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/q$m -o t -- python3 tools/bench_gemm_sp.py 123008 768 2304 > /dev/null 2> $O/err$m.txt
  python3 tools/pmc_digest.py $O/q$m "gemm_sp"
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/r$m -o t -- python3 tools/bench_gemm_sp.py 123008 768 2304 > /dev/null 2> $O/err$m.txt
  python3 tools/pmc_digest.py $O/r$m "gemm_sp"
done
