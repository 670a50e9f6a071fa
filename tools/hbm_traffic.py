"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE need separate
passes: MI355X_MICROARCH.md counter table).  Arguments: <fetch dir> <write dir> <out.json> [kernel pattern ...].

Units and correction as the guide's HBM section prescribes: both counters are KiB; on gfx950
FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B, so reads are DOUBLED;
WRITE_SIZE is exact for 16-B-per-lane stores and float atomics.  Infinity-Cache hits are counted as
traffic (the counters sit on the L2's memory side)."""
import collections
import csv
import glob
import json
import sys


def collect(root, counter, pats):
    f = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        pat = next((p for p in pats if p in name), None)
        if pat is not None and r["Counter_Name"] == counter:
            # "void (anonymous namespace)::conv_igemm_kernel<128, 128, 2, 2>((anonymous namespace)::ConvP)"
            acc[name[name.index(pat):].split("(")[0]].append(float(r["Counter_Value"]) * 1024.0)
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    pats = sys.argv[4:] or ["conv_", "wino_"]
    rd = collect(fetch_dir, "FETCH_SIZE", pats)
    wr = collect(write_dir, "WRITE_SIZE", pats)
    rows = {}
    for short in sorted(set(rd) | set(wr)):
        name = short
        r, w = rd.get(name, []), wr.get(name, [])
        n = max(len(r), len(w))
        read = 2.0 * sum(r) / max(len(r), 1)
        write = sum(w) / max(len(w), 1)
        rows[short] = dict(launches=n, read_bytes_per_launch=round(read), write_bytes_per_launch=round(write),
                           bytes_per_launch=round(read + write))
    import os

    cmd = os.environ.get("CLX_TRAFFIC_CMD", "bench.py --steps 1 --warmup 1 --no-infer --no-cpu-baseline")
    doc = dict(source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of "
                      f"{cmd}; mean over all launches of the run; FETCH_SIZE doubled (gfx950 correction)", kernels=rows)
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in rows.items():
        print(f"{k[:60]:60s} launches {v['launches']:4d}  read {v['read_bytes_per_launch'] / 1e9:7.3f} GB  "
              f"write {v['write_bytes_per_launch'] / 1e9:7.3f} GB  per launch")


main()
