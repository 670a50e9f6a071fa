"""embed stage of the 512^2 inference benchmark at different max_infer_batch settings."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellulus_amd.models import get_model
dev = torch.device("cuda:0")
cfg = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64,
           downsampling_factors=[[2, 2]], num_spatial_dims=2)
torch.manual_seed(0)
model = get_model(**cfg).to(dev); model.eval()
model.set_infer(p_salt_pepper=0.01, num_infer_iterations=16, device=dev)
raw = torch.rand(1, 1, 528, 528, device=dev); noise = torch.rand(1, 32, 1, 528, 528, device=dev)
for mb in (2, 4, 8, 16):
    model.max_infer_batch = mb
    model._plans = {}; model._infer_pair = None
    model.infer_on_device(raw, noise=noise); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): model.infer_on_device(raw, noise=noise)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(f"CLX_INFER_STREAMS={os.environ.get('CLX_INFER_STREAMS', '1')} max_infer_batch={mb}: embed {dt*1e3:.1f} ms  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
