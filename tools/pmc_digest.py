"""Digest of a rocprofv3 --pmc ... --kernel-trace --output-format csv run (argument: output directory):
per (kernel, grid): launches, mean duration, and counter means; MFMA busy fraction when
SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE were collected (busy / (GUI_ACTIVE/8 XCD * 1024 SIMDs))."""
import collections, csv, glob, sys
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "conv_"
f = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if pat not in n:
        continue
    key = n[n.index(pat):].split("(")[0][:48] + " grid=" + r["Grid_Size"]
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r and r["Start_Timestamp"]:
        acc[key]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, c in sorted(acc.items()):
    parts = [f"{key:64s}"]
    for name, vals in sorted(c.items()):
        parts.append(f"{name}={sum(vals) / len(vals):.4g}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        act = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
        parts.append(f"mfma_busy={busy / (act / 8 * 1024):.3f}")
    if "GRBM_GUI_ACTIVE" in c and "_us" in c:
        act = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
        parts.append(f"grbm_mhz={act / 8 / (sum(c['_us']) / len(c['_us'])):.0f}")
    print("  ".join(parts))
