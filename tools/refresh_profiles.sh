#!/bin/bash
# Regenerates the end-of-round evidence under gpurun_out/refresh (run on the GPU box through gpurun from
# the repo root: `gpurun -- bash tools/refresh_profiles.sh`); copy the results into profiles/ afterwards
# (names: profiles/README.md).  PMC passes are separate rocprofv3 runs with --kernel-trace only.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --streams 1 --no-infer --no-cpu-baseline --no-train-e2e > $O/bench_one_stream.json 2>/dev/null
CLX_BENCH_DETAIL=1 python bench.py --steps 6 --warmup 2 --no-infer --no-cpu-baseline --no-train-e2e > $O/per_layer.txt 2>&1
# kernel statistics, 2-D (+ the 3-D workload the default line also times): ONE stream (--streams 1: every kernel alone on
# the device — the pass bench.py takes its roofline numbers from; average durations must agree with roofline.avg_launch_ms)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 bench.py --streams 1 --steps 4 --warmup 2 --no-infer --no-cpu-baseline --no-train-e2e --no-train3d > $O/bench_under_rocprof.json 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3d -o t -- python3 bench.py --streams 1 --workload train3d --steps 4 --warmup 2 --no-infer --no-cpu-baseline --no-train-e2e > $O/bench3d_under_rocprof.json 2>/dev/null
# ... and the default command (two half batches on two streams; the trace holds both of bench.py's passes) with what
# shares the device when: tools/step_overlap.sh
bash tools/step_overlap.sh train2d > $O/overlap_train2d.txt 2>&1
bash tools/step_overlap.sh train3d > $O/overlap_train3d.txt 2>&1
export CLX_STREAMS=1      # every PMC pass and per-layer table below: kernels alone on the device
# matrix-pipe busy per kernel
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -o t -- python3 bench.py --steps 2 --warmup 1 --no-infer --no-cpu-baseline --no-train-e2e --no-train3d > /dev/null 2>&1
python3 tools/pmc_digest.py $O/pmc conv_ > $O/pmc_conv_kernels.txt
python3 tools/pmc_digest.py $O/pmc gemm_sp >> $O/pmc_conv_kernels.txt
python3 tools/pmc_digest.py $O/pmc chain64 >> $O/pmc_conv_kernels.txt
python3 tools/pmc_digest.py $O/pmc wino_ > $O/pmc_wino_kernels.txt
rm -rf $O/pmc
# HBM bytes per launch: training kernels ...
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_rd -o t -- python3 bench.py --steps 1 --warmup 1 --no-infer --no-cpu-baseline --no-train-e2e --no-train3d > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_wr -o t -- python3 bench.py --steps 1 --warmup 1 --no-infer --no-cpu-baseline --no-train-e2e --no-train3d > /dev/null 2>&1
python3 tools/hbm_traffic.py $O/pmc_rd $O/pmc_wr $O/hbm_traffic_train2d.json conv_ gemm_sp sp_split wino_ chain64 > $O/hbm_traffic_train2d.txt
rm -rf $O/pmc_rd $O/pmc_wr
# ... the 3-D workload ...
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_rd -o t -- python3 bench.py --workload train3d --steps 1 --warmup 1 --no-infer --no-cpu-baseline --no-train-e2e > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_wr -o t -- python3 bench.py --workload train3d --steps 1 --warmup 1 --no-infer --no-cpu-baseline --no-train-e2e > /dev/null 2>&1
CLX_TRAFFIC_CMD="bench.py --workload train3d --steps 1 --warmup 1 --no-infer --no-cpu-baseline --no-train-e2e" python3 tools/hbm_traffic.py $O/pmc_rd $O/pmc_wr $O/hbm_traffic_train3d.json conv_ wino_ chain64 > $O/hbm_traffic_train3d.txt
rm -rf $O/pmc_rd $O/pmc_wr
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -o t -- python3 bench.py --workload train3d --steps 2 --warmup 1 --no-infer --no-cpu-baseline --no-train-e2e > /dev/null 2>&1
python3 tools/pmc_digest.py $O/pmc conv_ > $O/pmc_conv_kernels_3d.txt
python3 tools/pmc_digest.py $O/pmc chain64 >> $O/pmc_conv_kernels_3d.txt
rm -rf $O/pmc
CLX_BENCH_DETAIL=1 python bench.py --workload train3d --steps 6 --warmup 2 --no-infer --no-cpu-baseline --no-train-e2e > $O/per_layer_3d.txt 2>&1
# ... and the streaming kernels of detect / segment (one 8192^2 image = 256 samples of 512^2 per launch)
python tools/bench_stream.py 4096 > $O/streaming_kernels_4096.txt 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream4k -o t -- python3 tools/bench_stream.py 4096 > /dev/null 2>&1
python tools/bench_stream.py 8192 > $O/streaming_kernels.txt 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stream -o t -- python3 tools/bench_stream.py 8192 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_rd -o t -- python3 tools/bench_stream.py 8192 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_wr -o t -- python3 tools/bench_stream.py 8192 > /dev/null 2>&1
CLX_TRAFFIC_CMD="tools/bench_stream.py 8192" python3 tools/hbm_traffic.py $O/pmc_rd $O/pmc_wr $O/hbm_traffic_streaming.json ms_ cc_ gs_ grow_shrink bucket_ histogram_kernel minmax_kernel noise_stats > $O/hbm_traffic_streaming.txt
rm -rf $O/pmc_rd $O/pmc_wr
for d in prof prof3d prof_stream prof_stream4k; do
  f=$(find $O/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv
done
rm -rf $O/prof $O/prof3d $O/prof_stream $O/prof_stream4k
# LDS bank conflicts / wait counters of the MFMA kernels
export CLX_STREAMS=1
bash tools/pmc_issue_counters.sh > /dev/null 2>&1
cat gpurun_out/pmc_f32/set1_f32.txt | grep -v "grey\|smallc" > $O/pmc_lds_conflicts_f32.txt
unset CLX_STREAMS
CLX_DETERMINISTIC=1 python bench.py --steps 10 --warmup 3 --no-infer --no-cpu-baseline --no-train-e2e > $O/bench_deterministic.json 2>/dev/null
ls -la $O
echo refresh done
# the inference tile (one stream: every kernel alone): kernel statistics, per-kernel table of one tile, HBM traffic, matrix-pipe busy
bash tools/infer_profile.sh > $O/infer_profile.log 2>&1
cp gpurun_out/infer_prof/* $O/ 2>/dev/null
# the reference's own GPU path (PyTorch + MIOpen) and the fused Winograd layer table
bash tools/gpu_library_kernels.sh > $O/gpu_library.log 2>&1
cp gpurun_out/gpu_library/kernels.txt $O/gpu_library_baseline_kernels.txt 2>/dev/null
cp gpurun_out/gpu_library/miopen_cmds.txt $O/gpu_library_baseline_miopen_cmds.txt 2>/dev/null
cp gpurun_out/gpu_library/line.json $O/gpu_library_baseline_line.json 2>/dev/null
timeout 300 python tools/exp/fused_bench.py 2>/dev/null | grep "^{" > $O/wino_fused_layers.txt
timeout 300 python tools/parity_trained_scale.py 2d 2>/dev/null | grep -v amdgpu > $O/parity_trained_scale_2d.txt
CLX_PRECISION=f32 timeout 300 python tools/parity_trained_scale.py 2d 2>/dev/null | grep -v amdgpu > $O/parity_trained_scale_2d_f32_mfma.txt
# issue / LDS / wait counters of the split-precision product on one long-K shape, and the product kernels alone
bash tools/pmc_sp.sh > /dev/null 2>&1
cp gpurun_out/pmc_sp/counters.txt $O/pmc_sp_counters.txt 2>/dev/null
python tools/bench_gemm_sp.py > $O/gemm_sp_shapes.txt 2>/dev/null
SP_WGRAD=1 SP_TIME_ONLY=1 python tools/bench_gemm_sp.py 256 128 128 2>/dev/null | grep wgrad >> $O/gemm_sp_shapes.txt
# the measured lines of the trained-network tests (error, instances, pixels that differ)
timeout 900 python -m pytest tests/test_gpu_trained_e2e.py -q -s 2>&1 | tr "\r" "\n" | grep -v "Warning\|warnings\|pin_memory\|^$\|it/s\]\|===> loss\|Checkpoint saved\|amdgpu.ids\|ExperimentConfig(\|Created logger" > $O/trained_e2e.txt
timeout 300 python tools/parity_trained_scale.py 3d 2>/dev/null | grep -v amdgpu > $O/parity_trained_scale_3d.txt
echo refresh complete
