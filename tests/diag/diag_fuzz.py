"""Per-parameter gradient error of one fuzz case (tests/test_gpu_fuzz.py) vs the f64 oracle."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as F
from cellulus_amd.models import get_model
from oracle.unet_oracle import OracleUNetModel

seed = int(sys.argv[1])
dev = torch.device("cuda:0")
cfg, spatial, batch = F._random_case(seed, wide=seed >= 100)
if len(sys.argv) > 2:
    batch = int(sys.argv[2])
print(cfg, spatial, batch)
torch.manual_seed(seed)
oracle = OracleUNetModel(**cfg).double()
for _n, layer in oracle.named_modules():
    if isinstance(layer, torch.nn.modules.conv._ConvNd):
        torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
        torch.nn.init.uniform_(layer.bias, -0.1, 0.1)
model = get_model(**cfg)
model.load_state_dict({k: v.float() for k, v in oracle.state_dict().items()}, strict=True)
model = model.to(dev)
raw = torch.rand(batch, cfg["in_channels"], *spatial)
ref = oracle(raw.double()); got = model(raw.to(dev))
print("fwd err", (got.detach().cpu().double() - ref.detach()).abs().max().item())
w = torch.randn_like(ref)
(ref * w).sum().backward(); (got * w.float().to(dev)).sum().backward()
for (n, po), (_, pm) in zip(oracle.named_parameters(), model.named_parameters()):
    g_ref, g = po.grad, pm.grad.detach().cpu().double()
    print(f"{n:45s} rel {(g - g_ref).norm().item() / max(g_ref.norm().item(), 1e-12):.3e} max {(g - g_ref).abs().max().item():.3e} |g| {g_ref.norm().item():.3e}")

# ---- intermediate gradients: oracle hooks vs plan.gbuf (post-activation in the oracle -> the plan
# stores PRE-activation gradients, so compare gbuf with grad * (y > 0))
if os.environ.get("DIAG_INTERMEDIATE"):
    oracle.zero_grad()
    acts = {}
    def keep(name):
        def hook(_m, _inp, out):
            out.retain_grad(); acts[name] = out
        return hook
    bb = oracle.backbone
    hs = [bb.l_conv[1].register_forward_hook(keep("l1.3")), bb.l_conv[0].register_forward_hook(keep("l0.3")),
          bb.r_up[0][0].register_forward_hook(keep("cat")), bb.r_conv[0][0].conv_pass[0].register_forward_hook(keep("r0.0pre"))]
    ref = oracle(raw.double()); (ref * w).sum().backward()
    plan = next(iter(model._plans.values()))
    t = plan.topo
    for name in ("l1.3", "l0.3"):
        shape, c = t.shapes[name]
        g = plan.gbuf[name].view(batch, *shape, -1)[..., :c].permute(0, 4, 1, 2, 3).cpu().double()
        a = acts[name]
        gr = a.grad * (a > 0)
        d = (g - gr).abs()
        print(name, "gbuf vs oracle: max", d.max().item(), "rel", (g - gr).norm().item() / gr.norm().item())
        # where are the errors?
        idx = (d > 1e-3 * gr.abs().max()).nonzero()
        print("  bad count", len(idx), "of", d.numel())
        if len(idx):
            for dim, nm in zip(range(5), "bczyx"):
                vals, cnt = idx[:, dim].unique(return_counts=True)
                print("   ", nm, dict(zip(vals.tolist(), cnt.tolist())))
    if "cat0" in plan.gbuf:
        cat = acts["cat"]
        g = plan.gbuf["cat0"].view(batch, *cat.shape[2:], -1).permute(0, 4, 1, 2, 3).cpu().double()
        print("cat ch", cat.shape[1], "gbuf ch", g.shape[1])
        gr = cat.grad
        for lo, hi, nm in ((0, 8, "skip"), (8, 24, "up")):
            d = (g[:, lo:hi] - gr[:, lo:hi]).abs()
            print(" cat", nm, "max", d.max().item(), "rel", d.norm().item() / gr[:, lo:hi].norm().item())
