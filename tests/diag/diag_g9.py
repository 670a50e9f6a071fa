"""Diagnose g9 mismatch: offsets, gather, loss of the HIP path vs torch ops on the same device tensors."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellulus_amd.models import get_model
from cellulus_amd.criterions import get_loss
from cellulus_amd.models.unet import UNetModel
from oracle import unet_oracle as UO

g = np.load("tests/golden/g9_train_iteration.npz")
dev = torch.device("cuda:0")
for nd in (2, 3):
    model = get_model(in_channels=1, out_channels=nd, num_fmaps=4, fmap_inc_factor=2, features_in_last_layer=8,
                      downsampling_factors=[[2] * nd], num_spatial_dims=nd)
    pre = f"w{nd}/init/"
    model.load_state_dict({k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}, strict=True)
    model = model.to(dev)
    raw, anchor, reference = (torch.from_numpy(g[f"b{nd}/0/{k}"]) for k in ("raw", "anchor", "reference"))
    with torch.no_grad():
        off = model(raw.to(dev))
    ref_off = torch.from_numpy(g[f"b{nd}/0/offsets"])
    print(nd, "offsets max diff", (off.cpu() - ref_off).abs().max().item())
    ea = UNetModel.select_and_add_coordinates(off, anchor.to(dev)).cpu()
    er = UNetModel.select_and_add_coordinates(off, reference.to(dev)).cpu()
    oa = UO.select_and_add_coordinates(ref_off, anchor)
    orr = UO.select_and_add_coordinates(ref_off, reference)
    print(nd, "gather diff", (ea - oa).abs().max().item(), (er - orr).abs().max().item())
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=nd, device=dev)
    l, o, r = crit(ea.to(dev), er.to(dev))
    lo, oo, ro = UO.oce_loss(oa, orr, 10.0, 1e-5)
    print(nd, "loss hip", l.item(), o.item(), r.item(), "oracle", lo.item(), oo.item(), ro.item(), "golden", g[f"losses{nd}"][0])
    # per-pair terms
    d2 = ((oa - orr) ** 2).sum(-1)
    print(nd, "pairs", d2.numel(), "min d2", d2.min().item(), "n close", int((d2 < 50).sum()))
