#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"; export TMPDIR=/tmp
O=gpurun_out/q1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p3 -o t -- python3 bench.py --workload train3d --steps 4 --warmup 2 --no-infer --no-cpu-baseline --no-train-e2e > $O/b3.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -o t -- python3 bench.py --steps 4 --warmup 2 --no-infer --no-cpu-baseline --no-train3d --no-train-e2e > $O/b2.json 2>/dev/null
for d in p2 p3; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; done
rm -rf $O/p2 $O/p3
