mkdir -p gpurun_out/r05f
timeout 280 python tools/exp/fused_e2e.py > gpurun_out/r05f/e2e.txt 2>&1; cat gpurun_out/r05f/e2e.txt | cut -c1-1500
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py tests/test_gpu_wino_fused.py -q -x 2>&1 | tail -15 > gpurun_out/r05f/test.log; cat gpurun_out/r05f/test.log
