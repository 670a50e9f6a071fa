#!/bin/bash
# Regenerates the end-of-round evidence under gpurun_out/ (run on the GPU box through gpurun from the repo
# root); copy the results into profiles/ afterwards (names: profiles/README.md).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
CLX_BENCH_DETAIL=1 python bench.py --steps 6 --warmup 2 --no-infer --no-cpu-baseline > $O/per_layer.txt 2>&1
python bench.py --workload train3d --steps 6 --warmup 2 --no-infer --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_train3d.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 $R/bench.py --steps 4 --warmup 2 --no-infer --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-infer --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/pmc_digest.py $O/pmc conv_ > $O/pmc_conv_kernels.txt
python3 $R/tools/pmc_digest.py $O/pmc wino_ > $O/pmc_wino_kernels.txt
rm -rf $O/pmc
echo refresh done
bash $R/tools/hbm_traffic.sh
