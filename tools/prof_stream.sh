#!/bin/bash
# per-kernel durations of the streaming kernels (tools/bench_stream.py SIZE) under rocprofv3 --kernel-trace --stats
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"; export TMPDIR=/tmp
S=${1:-4096}
O=gpurun_out/qs; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o t -- python3 tools/bench_stream.py $S > $O/stream_$S.txt 2>/dev/null
f=$(find $O/p -name "*kernel_stats.csv" | head -1); cp $f $O/stream_${S}_kernel_stats.csv
rm -rf $O/p
