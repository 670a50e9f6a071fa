cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_infer.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for sz in 4096 8192; do echo "== $sz"; python tools/bench_stream.py $sz 2>/dev/null | grep "ms_prepare"; done
for g in 1024 4096; do echo "== flags grid $g"; CLX_MS_FLAGS_GRID=$g python tools/bench_stream.py 4096 2>/dev/null | grep "ms_prepare"; done
for g in 1024 2048 4096; do echo "== scatter grid $g"; CLX_MS_SCATTER_GRID=$g python tools/bench_stream.py 4096 2>/dev/null | grep "ms_prepare"; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prep3 -o t -- python3 tools/bench_stream.py 4096 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prep3/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:50]
    if any(k in n for k in ("ms_", "minmax", "histogram")):
        print(f"{n:52s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  max {float(r['MaxNs'])/1e3:8.1f}")
PY
