"""Training with the opt-in precision against the default: the same initial weights and the same batches (blobs,
pairs drawn as the reference draws them) through N fused train steps at the 2-D benchmark configuration, once per
precision in this process (the default twice: its own run-to-run distance is the yardstick); prints the loss
trajectories and their relative distances.
Usage: python tests/diag/precision_tracking.py [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import sample_pairs, synthetic_raw  # noqa: E402
from cellulus_amd.criterions import get_loss  # noqa: E402
from cellulus_amd.models import get_model  # noqa: E402
from cellulus_amd.optim import Adam  # noqa: E402
from cellulus_amd.train import train_iteration  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
cfg = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64,
           downsampling_factors=[[2, 2]], num_spatial_dims=2)
torch.manual_seed(0)
init = get_model(**cfg)
for _n, layer in init.named_modules():
    if isinstance(layer, torch.nn.modules.conv._ConvNd):
        torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
state = {k: v.clone() for k, v in init.state_dict().items()}
batches = []
for s in range(8):
    a, r = sample_pairs(8, (256, 256), 10.0, 0.1, seed=s)
    batches.append((synthetic_raw(8, (256, 256), s), a, r))
curves = {}
for name in ("f32", "f32 again", "f32x3bf16"):
    prec = name.split()[0]
    os.environ["CLX_PRECISION"] = prec
    m = get_model(**cfg)
    m.load_state_dict(state)
    m = m.to(dev)
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=2, device=dev)
    opt = Adam(m.parameters(), lr=4e-5, weight_decay=0.01)
    losses = []
    for it in range(steps):
        loss, _oce, _ = train_iteration(batches[it % len(batches)], m, crit, opt, dev)
        losses.append(float(loss))
    curves[name] = np.asarray(losses)
    del m, opt, crit
    torch.cuda.empty_cache()
a, b, a2 = curves["f32"], curves["f32x3bf16"], curves["f32 again"]
rel = np.abs(a - b) / np.abs(a)
rel2 = np.abs(a - a2) / np.abs(a)      # the default against itself: float atomics in the weight gradient reorder sums
for it in list(range(0, steps, max(1, steps // 20))) + [steps - 1]:
    print(f"step {it:4d}  f32 {a[it]:14.2f}  f32x3bf16 {b[it]:14.2f}  rel {rel[it]:.2e}")
print(f"relative distance of the trajectories: first 10 steps max {rel[:10].max():.2e}, all {steps} steps max {rel.max():.2e}, "
      f"median {np.median(rel):.2e}; loss {a[0]:.1f} -> {a[-8:].mean():.1f} (f32), {b[-8:].mean():.1f} (f32x3bf16)")
print(f"the default precision against a second run of itself: first 10 steps max {rel2[:10].max():.2e}, all steps max {rel2.max():.2e}, "
      f"median {np.median(rel2):.2e}; loss -> {a2[-8:].mean():.1f}")
assert np.isfinite(a).all() and np.isfinite(b).all()
