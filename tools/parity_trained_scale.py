"""Embedding error at a TRAINED network's output scale: cfg-2 (256 fmaps, 256^2 crop) and
cfg-4 (3-D, 64^3) with the head scaled so that the offsets reach 15 px (object_size 30), per convolution algorithm:
    |HIP - float64 oracle|, |HIP - float32 CPU oracle|, and the float32 CPU oracle's own distance from float64,
in train mode (the plan that keeps activations) and in the inference plan (fused Winograd forms, round 5).

    python tools/parity_trained_scale.py [2d|3d] > gpurun_out/parity_trained_scale.txt
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as O  # noqa: E402  (a measurement tool, like tests/: the oracle is the checker)
from cellulus_amd.models import get_model  # noqa: E402

three_d = len(sys.argv) > 1 and sys.argv[1] == "3d"
if three_d:
    cfg = dict(in_channels=1, out_channels=3, num_fmaps=64, fmap_inc_factor=3, features_in_last_layer=64,
               downsampling_factors=[[2, 2, 2]], num_spatial_dims=3)
    crop = (64, 64, 64)
else:
    cfg = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64,
               downsampling_factors=[[2, 2]], num_spatial_dims=2)
    crop = (256, 256)


def blobs(crop, seed):
    rs = np.random.RandomState(seed)
    nd = len(crop)
    grids = np.meshgrid(*[np.arange(c, dtype=np.float32) for c in crop], indexing="ij")
    img = np.zeros(crop, dtype=np.float32)
    for c in np.stack(np.meshgrid(*[np.arange(24, c, 48) for c in crop], indexing="ij"), -1).reshape(-1, nd):
        c = c + rs.randint(-6, 7, size=nd)
        img += np.exp(-sum((g - ci) ** 2 for g, ci in zip(grids, c)) / (2 * 6.0 ** 2))
    img += rs.normal(0, 0.02, size=crop).astype(np.float32)
    return torch.from_numpy(np.clip(img, 0, 1)[None, None])


dev = torch.device("cuda:0")
torch.manual_seed(0)
oracle = O.OracleUNetModel(**cfg)
for _n, layer in oracle.named_modules():
    if isinstance(layer, torch.nn.modules.conv._ConvNd):
        torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
raw = blobs(crop, 3)
with torch.no_grad():
    last = oracle.head[2]
    s = 15.0 / oracle(raw).abs().max().item()
    last.weight.mul_(s)
    last.bias.mul_(s)
    t0 = time.time()
    ref32 = oracle(raw)
    o64 = O.OracleUNetModel(**cfg).double()
    o64.load_state_dict({k: v.double() for k, v in oracle.state_dict().items()})
    with O.gemm_convolutions(o64):
        ref64 = o64(raw.double())
print(f"{'3-D cfg-4' if three_d else '2-D cfg-2'}: output range {ref64.abs().max().item():.2f} (head x {s:.1f}); "
      f"|f32 CPU oracle - f64| = {(ref32.double() - ref64).abs().max().item():.3e}   [{time.time() - t0:.0f} s of CPU oracles]")
cpu_err = (ref32.double() - ref64).abs().max().item()

variants = [
    ("default", {}),
    ("all direct (CLX_WINOGRAD=0)", dict(CLX_WINOGRAD="0")),
    ("F(4x4) from 128 channels (r0.6 64->64 and the skip half direct)", dict(CLX_WINOGRAD_MIN_CHANNELS="128")),
    ("F(4x4) from 512 channels", dict(CLX_WINOGRAD_MIN_CHANNELS="512")),
]
if not three_d:
    variants.insert(2, ("F(2x2) (CLX_WINOGRAD_TILE=2)", dict(CLX_WINOGRAD_TILE="2")))
    variants.append(("no sub-pixel form (CLX_SUBPIXEL=0)", dict(CLX_SUBPIXEL="0")))
    variants.append(("three-launch Winograd in the inference plan (CLX_WINO_FUSED=0)", dict(CLX_WINO_FUSED="0")))
print(f"{'variant':74s} {'train-mode plan':>34s}   {'inference plan':>34s}")
print(f"{'':74s} {'|hip-f64|':>11s} {'/cpu':>5s} {'|hip-f32|':>11s}   {'|hip-f64|':>11s} {'/cpu':>5s} {'|hip-f32|':>11s}")
keys = ("CLX_WINOGRAD", "CLX_WINOGRAD_TILE", "CLX_WINOGRAD_MIN_CHANNELS", "CLX_WINOGRAD_MIN_CHANNELS_3D", "CLX_SUBPIXEL", "CLX_WINO_FUSED")
for name, env in variants:
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(env)
    if three_d and "CLX_WINOGRAD_MIN_CHANNELS" in env:
        os.environ["CLX_WINOGRAD_MIN_CHANNELS_3D"] = env["CLX_WINOGRAD_MIN_CHANNELS"]
    import importlib
    import cellulus_amd.models.plan as plan_mod
    importlib.reload(plan_mod)                       # the channel thresholds are read at import
    import cellulus_amd.models.unet as unet_mod
    importlib.reload(unet_mod)
    import cellulus_amd.models as models_mod
    importlib.reload(models_mod)
    model = models_mod.get_model(**cfg)
    model.load_state_dict(oracle.state_dict(), strict=True)
    model = model.to(dev)
    cols = []
    for mode in ("train", "eval"):
        if mode == "train":
            out = model(raw.to(dev)).detach().cpu()
        else:
            model.eval()
            with torch.no_grad():
                out = model(raw.to(dev)).detach().cpu()
        e64 = (out.double() - ref64).abs().max().item()
        e32 = (out - ref32).abs().max().item()
        cols.append(f"{e64:11.3e} {e64 / cpu_err:5.2f} {e32:11.3e}")
    print(f"{name:74s} {cols[0]}   {cols[1]}")
    del model
