"""GPU tier: seeded random U-Net configurations (channel counts, level counts, anisotropic
factors, batch, extents) through the HIP forward + backward vs the oracle — guards the
kernel-selection logic (small-channel / implicit-GEMM / sub-pixel / two-source) on shapes
the hand-picked configurations do not reach."""

import numpy as np
import pytest
import torch

from cellulus_amd.models import get_model
from cellulus_amd.models.plan import build_topology
from oracle.unet_oracle import OracleUNetModel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


def _random_case(seed, wide=False):
    """wide=True: >= 128 feature maps, so the Winograd kernels are selected as well."""
    rng = np.random.default_rng(seed)
    nd = 2 if rng.random() < (0.8 if wide else 0.65) else 3
    levels = 1 if wide else int(rng.integers(1, 3 if nd == 2 else 2))
    cfg = dict(
        in_channels=int(rng.integers(1, 6)),
        out_channels=nd,
        num_fmaps=int(rng.choice([128, 132, 160, 192] if wide else [1, 2, 3, 4, 5, 7, 8, 12, 16, 20])),
        fmap_inc_factor=int(rng.integers(1, 3 if wide else 4)),
        features_in_last_layer=int(rng.choice([16, 64, 130] if wide else [1, 3, 4, 8, 9, 16, 33])),
        downsampling_factors=[[int(rng.integers(1, 4)) for _ in range(nd)] for _ in range(levels)],
        num_spatial_dims=nd,
    )
    if all(f == 1 for fs in cfg["downsampling_factors"] for f in fs):
        cfg["downsampling_factors"][0][-1] = 2
    lo, hi = (30, 90) if nd == 2 else (18, 40)
    if wide:
        lo, hi = (30, 60) if nd == 2 else (18, 26)
    for _ in range(2000):
        spatial = tuple(int(rng.integers(lo, hi)) for _ in range(nd))
        try:
            build_topology(spatial=spatial, **cfg)
        except (ValueError, RuntimeError, AssertionError):
            continue
        return cfg, spatial, int(rng.integers(1, 3 if wide else 4))
    pytest.skip(f"no valid extent found for {cfg}")


@pytest.mark.parametrize("seed", list(range(24)) + [100 + k for k in range(8)])
def test_random_config_forward_backward_match_oracle(seed, device):
    cfg, spatial, batch = _random_case(seed, wide=seed >= 100)
    torch.manual_seed(seed)
    oracle = OracleUNetModel(**cfg).double()
    for _n, layer in oracle.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
            torch.nn.init.uniform_(layer.bias, -0.1, 0.1)
    model = get_model(**cfg)
    model.load_state_dict({k: v.float() for k, v in oracle.state_dict().items()}, strict=True)
    model = model.to(device)
    raw = torch.rand(batch, cfg["in_channels"], *spatial)
    ref = oracle(raw.double())
    got = model(raw.to(device))
    assert got.shape == ref.shape, (cfg, spatial)
    scale = max(1.0, ref.abs().max().item())
    err = (got.detach().cpu().double() - ref.detach()).abs().max().item()
    assert err < 1e-4 * scale, f"{cfg} {spatial} B={batch}: forward err {err} (scale {scale})"

    # Gradients.  A pre-activation that is ~0 can land on different sides of the ReLU in f32 (HIP)
    # and f64 (oracle); such a gate flip changes upstream gradients by a whole term (1e-4 .. 1e-2
    # relative, see tests/diag/diag_fuzz.py) and says nothing about the kernels.  So the oracle's backward
    # runs through the SAME gates as the device: every ReLU of the oracle passes its input where the
    # device's stored activation is positive.  With equal gates the two backward passes are the same
    # linear map and must agree to rounding.
    plan = [p for k, p in model._plans.items() if k[2]][0]
    relu_layers = [layer for layer in plan.topo.convs if layer.relu]
    modules = dict(oracle.named_modules())
    relus = []
    for layer in relu_layers:         # the ReLU that follows conv "<prefix>.<i>" is module "<prefix>.<i+1>"
        prefix, idx = layer.name.rsplit(".", 1)
        relus.append(modules[f"{prefix}.{int(idx) + 1}"])
        assert isinstance(relus[-1], torch.nn.ReLU), layer.name
    assert len(relus) == sum(isinstance(m, torch.nn.ReLU) for m in oracle.modules())
    hooks = []
    for m, layer in zip(relus, relu_layers):
        shape, c = plan.topo.shapes[layer.out]
        act = plan.buf[layer.out].view(batch, *shape, -1)[..., :c]
        gate = (act > 0).permute(0, 4, 1, 2, 3).cpu()
        if cfg["num_spatial_dims"] == 2:
            gate = gate[:, :, 0]
        hooks.append(m.register_forward_hook(lambda _m, inp, _out, gate=gate: inp[0] * gate))
    # ... and every max-pool routes its gradient to the element the device selected (two nearly equal
    # candidates are the same kind of discontinuity)
    import torch.nn.functional as F

    nd = cfg["num_spatial_dims"]
    for i, pool in enumerate(plan.topo.pools):
        shape, c = plan.topo.shapes[pool.src]
        act = plan.buf[pool.src].view(batch, *shape, -1)[..., :c].permute(0, 4, 1, 2, 3).cpu().double()
        fac = tuple(pool.factor)
        if nd == 2:
            _, idx = F.max_pool2d(act[:, :, 0], fac[1:], fac[1:], return_indices=True)
        else:
            _, idx = F.max_pool3d(act, fac, fac, return_indices=True)

        def route(_m, inp, out, idx=idx):
            return inp[0].flatten(2).gather(2, idx.flatten(2)).view_as(out)

        hooks.append(oracle.backbone.l_down[i].register_forward_hook(route))
    oracle.zero_grad()
    ref = oracle(raw.double())
    for h in hooks:
        h.remove()
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    (got * w.float().to(device)).sum().backward()
    for (n, po), (_, pm) in zip(oracle.named_parameters(), model.named_parameters()):
        g_ref, g = po.grad, pm.grad.detach().cpu().double()
        rel = (g - g_ref).norm().item() / max(g_ref.norm().item(), 1e-12)
        assert rel < 1e-4 or (g - g_ref).abs().max().item() < 1e-6, f"{cfg} {spatial} B={batch}: grad {n} rel {rel}"
