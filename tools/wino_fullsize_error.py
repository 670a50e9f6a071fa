"""Full benchmark configuration (256 fmaps, 256^2 crops): Winograd F(2x2) / F(4x4) network output and
parameter gradients against the direct implicit-GEMM path on the same device (which itself is ~1e-6
from the f64 oracle on the configurations where that can be checked)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellulus_amd.models import get_model

dev = torch.device("cuda:0")
three_d = len(sys.argv) > 1 and sys.argv[1] == "3d"
if three_d:      # BASELINE cfg-4: 64^3 crops, 64 feature maps
    cfg = dict(in_channels=1, out_channels=3, num_fmaps=64, fmap_inc_factor=3, features_in_last_layer=64,
               downsampling_factors=[[2, 2, 2]], num_spatial_dims=3)
else:
    cfg = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64,
               downsampling_factors=[[2, 2]], num_spatial_dims=2)
torch.manual_seed(0)
model = get_model(**cfg)
for _n, layer in model.named_modules():
    if isinstance(layer, torch.nn.modules.conv._ConvNd):
        torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
model = model.to(dev)
raw = torch.rand(2, 1, 64, 64, 64, device=dev) if three_d else torch.rand(2, 1, 256, 256, device=dev)
g = None
res = {}
variants = [("direct", dict(CLX_WINOGRAD="0")), ("F(2x2)", dict(CLX_WINOGRAD="1", CLX_WINOGRAD_TILE="2")),
            ("F(4x4)", dict(CLX_WINOGRAD="1", CLX_WINOGRAD_TILE="4"))]
if three_d:
    variants.pop(1)          # 3-D layers have the F(4x4) form only
for name, env in variants:
    os.environ.update(env)
    model._plans = {}
    model.zero_grad()
    out = model(raw)
    if g is None:
        g = torch.randn_like(out)
    out.backward(g)
    res[name] = (out.detach().clone(), [p.grad.detach().clone() for p in model.parameters()])
ref_out, ref_g = res["direct"]
print(f"output range {ref_out.abs().max().item():.3f}")
for name in [v[0] for v in variants[1:]]:
    out, gr = res[name]
    e = (out - ref_out).abs().max().item()
    rel = max(((a - b).norm() / b.norm()).item() for a, b in zip(gr, ref_g))
    print(f"{name}: max |out - direct| = {e:.3e}   worst parameter-gradient rel L2 vs direct = {rel:.3e}")
