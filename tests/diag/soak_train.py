"""Soak: N fused train steps at the 2-D benchmark configuration on a few fixed batches of uniform noise;
prints the loss trajectory (must stay finite).  On this structure-free input the loss dips for ~30 steps
and then drifts up again; the direct-convolution path (CLX_WINOGRAD=0) shows the same curve, i.e. it is the
optimisation problem, not the Winograd arithmetic — compare the two runs when touching the kernels."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellulus_amd.criterions import get_loss
from cellulus_amd.models import get_model
from cellulus_amd.optim import Adam
from cellulus_amd.train import train_iteration

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = get_model(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64,
              downsampling_factors=[[2, 2]], num_spatial_dims=2)
for _n, layer in m.named_modules():
    if isinstance(layer, torch.nn.modules.conv._ConvNd):
        torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
m = m.to(dev)
crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=2, device=dev)
opt = Adam(m.parameters(), lr=4e-5, weight_decay=0.01)
rng = np.random.default_rng(0)
batches = []
for _ in range(4):
    anchors = np.repeat(rng.integers(10, 231, size=(8, 4840, 2)), 31, axis=1)
    offs = rng.integers(-9, 10, size=anchors.shape); offs[np.abs(offs).sum(-1) == 0] = 1
    # blobs so that there is structure to learn
    img = torch.rand(8, 1, 256, 256)
    batches.append((img, torch.from_numpy(anchors.astype(np.int64)), torch.from_numpy((anchors + offs).astype(np.int64))))
losses = []
for it in range(steps):
    loss, oce, _ = train_iteration(batches[it % 4], m, crit, opt, dev)
    losses.append(loss)
    if it % 10 == 0 or it == steps - 1:
        print(it, f"{loss:.1f}", flush=True)
assert all(np.isfinite(losses)), "non-finite loss"
print("first 4 mean", np.mean(losses[:4]), "min", np.min(losses), "last 4 mean", np.mean(losses[-4:]))
assert np.min(losses) < losses[0]
print("soak ok")
