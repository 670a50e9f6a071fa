"""Where along the network does the HIP path's distance from the float64 oracle grow faster than the float32 CPU
oracle's?  cfg-2, trained-scale head; per ReLU layer (call order): max |act - f64| / max |f64| for HIP and CPU f32."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import unet_oracle as O
from cellulus_amd.models import get_model
sys.argv = [sys.argv[0]]
exec(open(os.path.join(os.path.dirname(__file__), "..", "parity_trained_scale.py")).read().split("dev = torch.device")[0])
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
    last.weight.mul_(s); last.bias.mul_(s)
def relu_acts(model, x, gemm=False):
    acts = []
    hooks = [m.register_forward_hook(lambda _m, _i, o: acts.append(o.detach())) for m in model.modules() if isinstance(m, torch.nn.ReLU)]
    with torch.no_grad():
        if gemm:
            with O.gemm_convolutions(model):
                out = model(x)
        else:
            out = model(x)
    for h in hooks: h.remove()
    return acts, out
a32, o32 = relu_acts(oracle, raw)
o64m = O.OracleUNetModel(**cfg).double(); o64m.load_state_dict({k: v.double() for k, v in oracle.state_dict().items()})
a64, o64 = relu_acts(o64m, raw.double(), gemm=True)
for label, env in (("default", {}), ("all direct", {"CLX_WINOGRAD": "0"})):
    os.environ.pop("CLX_WINOGRAD", None); os.environ.update(env)
    model = get_model(**cfg); model.load_state_dict(oracle.state_dict(), strict=True); model = model.to(dev)
    out = model(raw.to(dev)).detach().cpu()
    plan = next(iter(model._plans.values()))
    relu_layers = [l for l in plan.topo.convs if l.relu]
    print(f"== {label}: output |hip-f64| {(out.double()-o64).abs().max():.3e}  |cpu-f64| {(o32.double()-o64).abs().max():.3e}")
    for l, r64, r32 in zip(relu_layers, a64, a32):
        shape, c = plan.topo.shapes[l.out]
        h = plan.buf[l.out].view((plan.B,) + tuple(shape) + (-1,))[..., :c].permute(0, 4, 1, 2, 3).contiguous().cpu()[:, :, 0]
        sc = r64.abs().max().item()
        eh = (h.double() - r64).abs().max().item() / sc; ec = (r32.double() - r64).abs().max().item() / sc
        rh = ((h.double() - r64) ** 2).mean().sqrt().item() / sc; rc = ((r32.double() - r64) ** 2).mean().sqrt().item() / sc
        print(f"  {l.name:36s} algo {plan.algo[l.name]['fwd']} K={l.cin*l.taps:5d}  hip max {eh:.2e} rms {rh:.2e}   cpu max {ec:.2e} rms {rc:.2e}   ratio rms {rh/rc:.2f}")
