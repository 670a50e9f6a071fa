#!/bin/bash
# which torch fill kernels a train step still launches, with their sizes and neighbours in the trace
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/fills
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/fills -o t -- python3 $R/bench.py ${FILL_ARGS:---streams 1} --steps 2 --warmup 1 --no-infer --no-cpu-baseline --no-train-e2e --no-train3d > /dev/null 2>&1
f=$(find /tmp/fills -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'EOF'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
last = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
start = last[-2] if len(last) >= 2 else 0
for i in range(start, n):
    r = rows[i]
    k = r["Kernel_Name"]
    if "Fill" in k or "zero_many" in k or "elementwise" in k or "fillBuffer" in k:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        prev = rows[i - 1]["Kernel_Name"][:50] if i else ""
        nxt = rows[i + 1]["Kernel_Name"][:50] if i + 1 < n else ""
        print(f"{k[:70]:70s} grid={r['Grid_Size_X']:>10s} {dur:8.1f} us  after [{prev}] before [{nxt}]")
EOF
