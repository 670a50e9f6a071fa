"""Per region between two s_barrier instructions of one kernel in a -save-temps .s file: MFMAs, LDS reads, LDS-DMA loads,
scratch (spill) instructions, and the backward branches (loops).   python tools/exp/asm_regions.py file.s kernel_substring"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and ":" in l.split(";")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
labels = {}
for i in range(start, end):
    m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
    if m:
        labels[m.group(1)] = i
reg = dict(mfma=0, ds_read=0, glds=0, scratch=0, valu=0)
last = start
def flush(i, what):
    global reg, last
    print(f"{last - start:6d}..{i - start:6d} {what:10s} " + " ".join(f"{k}={v}" for k, v in reg.items()))
    reg = dict(mfma=0, ds_read=0, glds=0, scratch=0, valu=0)
    last = i
for i in range(start, end):
    l = lines[i]
    if "s_barrier" in l:
        flush(i, "barrier")
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        flush(i, f"loop->{labels[m.group(1)] - start}")
    if "v_mfma" in l: reg["mfma"] += 1
    elif "ds_read" in l: reg["ds_read"] += 1
    elif "global_load_lds" in l: reg["glds"] += 1
    elif "scratch_" in l: reg["scratch"] += 1
    elif re.match(r"\s+v_", l): reg["valu"] += 1
flush(end, "end")
