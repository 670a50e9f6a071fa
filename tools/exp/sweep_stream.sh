cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_infer.py -x -q -k "otsu" 2>&1 | tail -40
python -m pytest tests/test_gpu_fullsize.py -x -q -k "postprocessing" 2>&1 | tail -40
echo "== old f64 kernel"; CLX_MS_PREP_OLD=1 python tools/bench_stream.py 4096 2>/dev/null | grep "ms_prepare "
echo "== old f64 kernel + lb256"; CLX_MS_PREP_OLD=1 CLX_LIB=cellulus_amd/libclx.so.lb256 python tools/bench_stream.py 4096 2>/dev/null | grep "ms_prepare "
echo "== 8192 old f64 kernel"; CLX_MS_PREP_OLD=1 python tools/bench_stream.py 8192 2>/dev/null | grep "ms_prepare "
echo "== 8192 old f64 kernel + lb256"; CLX_MS_PREP_OLD=1 CLX_LIB=cellulus_amd/libclx.so.lb256 python tools/bench_stream.py 8192 2>/dev/null | grep "ms_prepare "
CLX_MS_PREP_OLD=1 CLX_LIB=cellulus_amd/libclx.so.lb256 python -m pytest tests/test_gpu_infer.py -x -q -k "mean_shift or handover" 2>&1 | tail -3
