#!/bin/bash
# Which vendor-library kernels the reference's own GPU path (plain PyTorch on "cuda:0": MIOpen) runs for the benchmark
# train step and inference tile: kernel statistics of bench.gpu_library_baseline alone + MIOpen's own command log.
#   gpurun -- bash tools/gpu_library_kernels.sh   -> gpurun_out/gpu_library/{kernels.txt,miopen_cmds.txt,line.json}
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/gpu_library
rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp
cat > $O/run.py <<'PY'
import json, sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
print(json.dumps(bench.gpu_library_baseline(bench.WORKLOADS["train2d"], dev, steps=2)))
PY
MIOPEN_ENABLE_LOGGING_CMD=1 timeout 600 python3 $O/run.py > $O/line.json 2> $O/miopen_raw.txt
grep -o "MIOpenDriver.*" $O/miopen_raw.txt | sort | uniq -c | sort -rn > $O/miopen_cmds.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 $O/run.py > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $O/kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"kernels of bench.gpu_library_baseline (warm-up + 2 train steps at batch 8 + 2 inference tiles), total {tot/1e6:.1f} ms")
for r in rows[:25]:
    print(f"{int(r['Calls']):6d} {int(r['TotalDurationNs'])/1e6:10.2f} ms {float(r['Percentage']):6.2f} %  {r['Name'][:150]}")
PY
rm -rf $O/prof $O/miopen_raw.txt
cat $O/line.json; head -30 $O/kernels.txt; head -20 $O/miopen_cmds.txt
