mkdir -p gpurun_out/r05n
for rep in 1 2; do
for lib in cellulus_amd/libclx.so cellulus_amd/libclx.so.scalarflush; do
  for f in 2 0; do
  CLX_IGEMM_FLUSH=$f python tools/bench_with_lib.py $lib --no-infer --no-cpu-baseline --no-train-e2e --no-train3d 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib flush=$f', d['value'], d['ms_per_step'], d['roofline']['per_kernel']['conv_igemm_kernel<128,128,2,2>']['tflops'])"
  done
done
done
CLX_IGEMM_FLUSH=2 timeout 300 python tools/parity_trained_scale.py 2d 2>&1 | grep "default"
