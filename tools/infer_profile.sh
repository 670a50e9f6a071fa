#!/bin/bash
# The inference tile under rocprofv3 (run on the GPU box: `gpurun -- bash tools/infer_profile.sh`): kernel statistics of
# tools/infer_gaps.py (four infer_on_device calls of the 512^2 benchmark tile), the per-kernel table of ONE tile, and the
# HBM bytes per launch (separate --pmc passes).  Results under gpurun_out/infer_prof; copy into profiles/ as rNN_infer_*.
set -uo pipefail
R="${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
O="$R/gpurun_out/infer_prof"
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o t -- python3 tools/infer_gaps.py > /dev/null 2>&1
python3 tools/infer_gaps.py --digest "$O/prof" > "$O/infer_tile_kernels.txt"
f=$(find "$O/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$O/infer_kernel_stats.csv"
rm -rf "$O/prof"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_rd" -o t -- python3 tools/infer_gaps.py > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_wr" -o t -- python3 tools/infer_gaps.py > /dev/null 2>&1
CLX_TRAFFIC_CMD="tools/infer_gaps.py (four 512^2 tiles of 32 noisy forwards)" python3 tools/hbm_traffic.py "$O/pmc_rd" "$O/pmc_wr" "$O/hbm_traffic_infer.json" conv_ gemm_sp sp_split wino_ chain64 noise_stats > "$O/hbm_traffic_infer.txt"
rm -rf "$O/pmc_rd" "$O/pmc_wr"
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc" -o t -- python3 tools/infer_gaps.py > /dev/null 2>&1
python3 tools/pmc_digest.py "$O/pmc" conv_ > "$O/pmc_infer_kernels.txt"
python3 tools/pmc_digest.py "$O/pmc" gemm_sp >> "$O/pmc_infer_kernels.txt"
python3 tools/pmc_digest.py "$O/pmc" chain64 >> "$O/pmc_infer_kernels.txt"
python3 tools/pmc_digest.py "$O/pmc" wino_ >> "$O/pmc_infer_kernels.txt"
rm -rf "$O/pmc"
# the MFMA kernels of one tile as libclx's launch events see them (bench_infer.py: infer.roofline.all_mfma_kernels) against
# the same kernels' rows of the trace: within 3 % or the accounting is wrong
python3 tools/infer_mfma_check.py "$O/infer_tile_kernels.txt" >> "$O/infer_tile_kernels.txt" 2>&1
ls -la "$O"
cat "$O/infer_tile_kernels.txt"
