"""infer() end to end with the chunks on 2 or 4 streams, same box, alternating (CLX_INFER_STREAMS is read per call)."""
import sys, os; sys.path.insert(0, ".")
import torch
from bench_infer import e2e_infer
dev = torch.device("cuda:0")
n = int(os.environ.get("E2E_SAMPLES", "48"))
for rep in range(3):
    for s in ("2", "4"):
        os.environ["CLX_INFER_STREAMS"] = s
        r = e2e_infer(dev, samples=n, size=512)
        print("streams", s, f"512^2 x {n}:", r["mpixels_s"], "Mpixels/s", r["ms_per_sample"], "ms per sample", flush=True)
