import sys, os; sys.path.insert(0, ".")
import torch
from bench_infer import e2e_infer
dev = torch.device("cuda:0")
for rep in range(2):
    for s in ("2", "4"):
        os.environ["CLX_INFER_STREAMS"] = s
        r = e2e_infer(dev, samples=16, size=512)
        r2 = e2e_infer(dev, samples=16, size=256)
        print("streams", s, "512:", r["mpixels_s"], r["ms_per_sample"], " 256:", r2["mpixels_s"], r2["ms_per_sample"], flush=True)
