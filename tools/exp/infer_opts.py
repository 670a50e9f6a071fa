"""Embedding stage of the benchmark inference tile (512^2, 32 noisy forwards) under a few plan options."""
import os, sys, time
sys.path.insert(0, ".")
import torch
from bench_infer import embed_stage
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
ref = None
for rep in range(2):
  for streams, off in (("2", "1"),):
    for mb in (8, 16, 4):
        os.environ["CLX_INFER_OFFSET_OP"] = off
        os.environ["CLX_INFER_STREAMS"] = streams
        model.max_infer_batch = mb
        model._plans = {}
        model._infer_pair = None
        model._clean_plan = None
        torch.cuda.empty_cache()
        t, emb, prof = embed_stage(model, dev, 512, 16, 3)
        if ref is None:
            ref = emb.clone()
        print(f"streams {streams} offset-op {off:>2s} chunk {mb:2d}: {t * 1e3:8.2f} ms/tile  identical to the first: {torch.equal(ref, emb)}  "
              f"mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GB", flush=True)
