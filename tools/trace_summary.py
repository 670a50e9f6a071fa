"""Per-(kernel, grid) average durations from a rocprofv3 --kernel-trace CSV (argument: directory)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if pat in n:
        short = n.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[-44:]
        per[f"{short} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(per.items()):
    print(f"{k:70s} n={len(v):3d} avg_us={sum(v) / len(v):9.1f}")
