#!/bin/bash
# copies gpurun_out/refresh/* (tools/refresh_profiles.sh) into profiles/ under the round's names: bash tools/copy_refresh.sh r02
set -e
P=${1:-r02}; R=gpurun_out/refresh
cp $R/bench_default.json profiles/${P}_bench_default.json
cp $R/bench_under_rocprof.json profiles/${P}_train2d_bench_under_rocprof.json
cp $R/bench3d_under_rocprof.json profiles/${P}_train3d_bench_under_rocprof.json
cp $R/prof_kernel_stats.csv profiles/${P}_train2d_kernel_stats.csv
cp $R/prof3d_kernel_stats.csv profiles/${P}_train3d_kernel_stats.csv
cp $R/prof_stream_kernel_stats.csv profiles/${P}_streaming_kernel_stats.csv
cp $R/per_layer.txt profiles/${P}_train2d_per_layer.txt
cp $R/per_layer_3d.txt profiles/${P}_train3d_per_layer.txt
cp $R/pmc_conv_kernels.txt profiles/${P}_pmc_conv_kernels.txt
cp $R/pmc_conv_kernels_3d.txt profiles/${P}_pmc_conv_kernels_3d.txt
cp $R/pmc_wino_kernels.txt profiles/${P}_pmc_wino_kernels.txt
cp $R/streaming_kernels.txt profiles/${P}_streaming_kernels.txt
cp $R/hbm_traffic_train2d.txt profiles/${P}_hbm_traffic_train2d.txt
cp $R/hbm_traffic_train3d.txt profiles/${P}_hbm_traffic_train3d.txt
cp $R/hbm_traffic_streaming.txt profiles/${P}_hbm_traffic_streaming.txt
cp $R/hbm_traffic_train2d.json $R/hbm_traffic_train3d.json $R/hbm_traffic_streaming.json profiles/
cp $R/pmc_lds_conflicts_f32.txt profiles/${P}_pmc_lds_conflicts_f32.txt
[ -f $R/streaming_kernels_4096.txt ] && cp $R/streaming_kernels_4096.txt profiles/${P}_streaming_kernels_4096.txt
[ -f $R/prof_stream4k_kernel_stats.csv ] && cp $R/prof_stream4k_kernel_stats.csv profiles/${P}_streaming_4096_kernel_stats.csv
[ -f $R/bench_one_stream.json ] && cp $R/bench_one_stream.json profiles/${P}_bench_one_stream.json
[ -f $R/overlap_train2d.txt ] && cp $R/overlap_train2d.txt profiles/${P}_two_streams_overlap_train2d.txt
[ -f $R/overlap_train3d.txt ] && cp $R/overlap_train3d.txt profiles/${P}_two_streams_overlap_train3d.txt
[ -f $R/bench_deterministic.json ] && cp $R/bench_deterministic.json profiles/${P}_bench_deterministic.json

[ -f $R/infer_kernel_stats.csv ] && cp $R/infer_kernel_stats.csv profiles/${P}_infer_kernel_stats.csv
[ -f $R/infer_tile_kernels.txt ] && cp $R/infer_tile_kernels.txt profiles/${P}_infer_tile_kernels.txt
[ -f $R/hbm_traffic_infer.json ] && cp $R/hbm_traffic_infer.json profiles/hbm_traffic_infer.json
[ -f $R/hbm_traffic_infer.txt ] && cp $R/hbm_traffic_infer.txt profiles/${P}_hbm_traffic_infer.txt
[ -f $R/pmc_infer_kernels.txt ] && cp $R/pmc_infer_kernels.txt profiles/${P}_pmc_infer_kernels.txt
[ -f $R/gpu_library_baseline_kernels.txt ] && cp $R/gpu_library_baseline_kernels.txt profiles/${P}_gpu_library_baseline_kernels.txt
[ -f $R/gpu_library_baseline_miopen_cmds.txt ] && cp $R/gpu_library_baseline_miopen_cmds.txt profiles/${P}_gpu_library_baseline_miopen_cmds.txt
[ -f $R/gpu_library_baseline_line.json ] && cp $R/gpu_library_baseline_line.json profiles/${P}_gpu_library_baseline_line.json
[ -f $R/wino_fused_layers.txt ] && cp $R/wino_fused_layers.txt profiles/${P}_wino_fused_layers.txt
[ -f $R/parity_trained_scale_2d.txt ] && cp $R/parity_trained_scale_2d.txt profiles/${P}_parity_trained_scale_2d.txt
[ -f $R/parity_trained_scale_3d.txt ] && cp $R/parity_trained_scale_3d.txt profiles/${P}_parity_trained_scale_3d.txt
[ -f $R/parity_trained_scale_2d_f32_mfma.txt ] && cp $R/parity_trained_scale_2d_f32_mfma.txt profiles/${P}_parity_trained_scale_2d_f32_mfma.txt
[ -f $R/pmc_sp_counters.txt ] && cp $R/pmc_sp_counters.txt profiles/${P}_pmc_sp_counters.txt
[ -f $R/gemm_sp_shapes.txt ] && cp $R/gemm_sp_shapes.txt profiles/${P}_gemm_sp_shapes.txt
[ -f $R/trained_e2e.txt ] && cp $R/trained_e2e.txt profiles/${P}_trained_e2e.txt

true
