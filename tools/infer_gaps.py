"""infer_on_device() calls (512^2 tile, 32 noisy forwards; Kaiming weights as bench_infer.py) for a kernel trace:
rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python3 tools/infer_gaps.py ; then tools/infer_gaps.py --digest OUT"""
import sys
if len(sys.argv) > 2 and sys.argv[1] == "--digest":
    import csv, glob, collections, re
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "noise_stats" in r["Kernel_Name"]]
    a, b = marks[-2], marks[-1]
    seg = rows[a + 1:b + 1]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
    span = (int(seg[-1]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3
    prev = int(rows[a]["End_Timestamp"]); gaps = []
    for r in seg:
        gaps.append(((int(r["Start_Timestamp"]) - prev) / 1e3, r["Kernel_Name"][:60])); prev = max(prev, int(r["End_Timestamp"]))
    print(f"ONE tile (the last infer_on_device call of the trace: kernels between two noise_stats launches, the second included)")
    print(f"kernels {len(seg)}, busy {busy:.0f} us, span {span:.0f} us, idle {span - busy:.0f} us")
    print("largest gaps before a kernel (us):", sorted(gaps, reverse=True)[:5])
    per = collections.OrderedDict()
    for r in seg:
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*$", "", name)[:110] or r["Kernel_Name"][:110]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        n, t = per.get(name, (0, 0.0))
        per[name] = (n + 1, t + d)
    print(f"{'kernel':112s} {'launches':>8s} {'us/tile':>10s} {'avg us':>9s} {'share':>7s}")
    for name, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:112s} {n:8d} {t:10.1f} {t / n:9.1f} {t / span:7.3f}")
    print(f"{'(idle)':112s} {'':8s} {span - busy:10.1f} {'':9s} {(span - busy) / span:7.3f}")
    sys.exit(0)
sys.path.insert(0, ".")
import os
os.environ.setdefault("CLX_INFER_STREAMS", "1")      # every kernel alone on the device: the pass bench_infer.py takes its rooflines from
import numpy as np
import torch
from cellulus_amd.models import get_model
dev = torch.device("cuda:0")
cfg = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64,
           downsampling_factors=[[2, 2]], num_spatial_dims=2)
torch.manual_seed(0)
model = get_model(**cfg).to(dev)
for _n, layer in model.named_modules():
    if isinstance(layer, torch.nn.modules.conv._ConvNd):
        torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
model.eval()
model.set_infer(p_salt_pepper=0.01, num_infer_iterations=16, device=dev)
raw = torch.rand(1, 1, 528, 528, device=dev)
noise = torch.rand(1, 32, 1, 528, 528, device=dev)
for _ in range(4):
    model.infer_on_device(raw, noise=noise)
torch.cuda.synchronize()
