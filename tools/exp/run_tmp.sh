mkdir -p gpurun_out/r05k
for f in 0 1 2 4; do
  echo "=== CLX_IGEMM_FLUSH=$f"
  CLX_IGEMM_FLUSH=$f timeout 600 python tools/parity_trained_scale.py 2d 2>&1 | grep -v amdgpu | grep "default\|three-launch\|from 128"
  CLX_IGEMM_FLUSH=$f timeout 600 python bench.py --no-infer --no-cpu-baseline --no-train-e2e --no-train3d > gpurun_out/r05k/bench_$f.json 2> gpurun_out/r05k/bench_$f.err; python - $f <<'PY'
import json, sys
d=json.load(open(f'gpurun_out/r05k/bench_{sys.argv[1]}.json'))
print(d['value'], d['ms_per_step'], d['roofline']['per_kernel']['conv_igemm_kernel<128,128,2,2>'])
PY
done > gpurun_out/r05k/flush.txt 2>&1
cat gpurun_out/r05k/flush.txt
echo "=== fused only for N = 64 (inference plan), flush 2"
CLX_WINO_FUSED_MAX_CHANNELS=0 timeout 600 python tools/parity_trained_scale.py 2d 2>&1 | grep -v amdgpu | grep "default"
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_chain.py tests/test_gpu_fastpath.py tests/test_gpu_wino_fused.py tests/test_gpu_train.py tests/test_gpu_fuzz.py -q 2>&1 | tail -4
