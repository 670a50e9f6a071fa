import sys, os; sys.path.insert(0, ".")
import torch
from bench_infer import embed_stage
from cellulus_amd.models import get_model
dev = torch.device("cuda:0")
cfg = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64, downsampling_factors=[[2, 2]], num_spatial_dims=2)
torch.manual_seed(0)
model = get_model(**cfg).to(dev)
for _n, layer in model.named_modules():
    if isinstance(layer, torch.nn.modules.conv._ConvNd):
        torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
model.eval(); model.set_infer(p_salt_pepper=0.01, num_infer_iterations=16, device=dev)
for size in (512, 256):
    for sp in ("1", "0", "1", "0"):
        os.environ["CLX_SPARSE_NOISE"] = sp
        t, emb, prof = embed_stage(model, dev, size, 16, 4)
        print(size, "sparse", sp, f"{t*1e3:.2f} ms per tile")
