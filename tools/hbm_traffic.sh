#!/bin/bash
# HBM traffic per launch of the convolution kernels (run on the GPU box through gpurun from the repo root):
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes -> gpurun_out/refresh/hbm_traffic_train2d.{json,txt};
# copy the .json to profiles/ (bench.py reads roofline.traffic from it).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_rd -o t -- python3 $R/bench.py --steps 1 --warmup 1 --no-infer --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_wr -o t -- python3 $R/bench.py --steps 1 --warmup 1 --no-infer --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/hbm_traffic.py $O/pmc_rd $O/pmc_wr $O/hbm_traffic_train2d.json > $O/hbm_traffic_train2d.txt
rm -rf $O/pmc_rd $O/pmc_wr
echo traffic done
