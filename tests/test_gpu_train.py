"""GPU parity of the train-step tail: gather, OCE loss (+ gradient), Adam and the
fused train_iteration vs the CPU oracle (cellulus/train.py:160-180 restated)."""

import numpy as np
import pytest
import torch

from cellulus_amd.criterions import get_loss
from cellulus_amd.models import get_model
from cellulus_amd.models.unet import UNetModel
from cellulus_amd.optim import Adam
from cellulus_amd.train import train_iteration
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu


def _coords(B, P, shape, rng):
    """(B, P, ND) int64 with column 0 = x (last axis)."""
    cols = [rng.integers(0, s, size=(B, P)) for s in shape[::-1]]
    return torch.from_numpy(np.stack(cols, axis=2).astype(np.int64))


@pytest.mark.parametrize("shape", [(9, 11), (5, 6, 7)])
def test_gather_add_fwd_bwd(shape, device):
    rng = np.random.default_rng(0)
    B, P, ND = 3, 257, len(shape)
    out = torch.randn(B, ND, *shape, requires_grad=True)
    coords = _coords(B, P, shape, rng)
    coords[0, :40] = coords[0, 0]           # heavy duplication, as np.repeat produces
    ref = O.select_and_add_coordinates(out, coords)
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    out_d = out.detach().to(device).requires_grad_(True)
    got = UNetModel.select_and_add_coordinates(out_d, coords.to(device))
    assert torch.allclose(got.cpu(), ref.detach(), atol=1e-6)
    (got * w.to(device)).sum().backward()
    assert torch.allclose(out_d.grad.cpu(), out.grad, atol=1e-5)


def test_gather_known_answer(device):
    # SURVEY.md §8c probe: out[0,0] = arange(30).view(5,6), coord (x=1, y=2) -> 13 + 1 = 14
    out = torch.zeros(1, 2, 5, 6)
    out[0, 0] = torch.arange(30.0).view(5, 6)
    got = UNetModel.select_and_add_coordinates(out.to(device), torch.tensor([[[1, 2]]], device=device))
    assert got.cpu().tolist() == [[[14.0, 2.0]]]


@pytest.mark.parametrize("nd", [2, 3])
def test_oce_loss_matches_oracle(nd, device):
    torch.manual_seed(0)
    a = (torch.randn(4, 1000, nd) * 4).requires_grad_(True)
    r = a.detach() + torch.randn(4, 1000, nd) * 3
    with torch.no_grad():
        a[0, 0] = 0.0                # zero-norm row
        r[0, 1] = a[0, 1]            # zero-distance row
    loss, oce, reg = O.oce_loss(a, r, 10.0, 1e-5)
    loss.backward()
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=nd, device=device)
    a_d = a.detach().to(device).requires_grad_(True)
    l2, o2, r2 = crit(a_d, r.to(device))
    l2.backward()
    assert abs(l2.item() - loss.item()) < 1e-5 * max(1.0, abs(loss.item()))
    assert abs(o2.item() - oce.item()) < 1e-5 * max(1.0, abs(oce.item()))
    assert abs(r2.item() - reg.item()) < 1e-5 * max(1e-3, abs(reg.item())) + 1e-9
    assert torch.isfinite(a_d.grad).all()
    assert torch.allclose(a_d.grad.cpu(), a.grad, atol=1e-6, rtol=1e-4)


def test_oce_known_answer(device):
    # T=10, w=1e-5
    a = torch.tensor([[[3.0, 4.0], [1.0, 1.0], [0.0, 0.0]]], device=device, requires_grad=True)
    r = torch.tensor([[[0.0, 0.0], [1.0, 1.0], [2.0, 0.0]]], device=device)
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=2, device=device)
    loss, oce, reg = crit(a, r)
    loss.backward()
    assert abs(loss.item() - 1.2476590872) < 1e-6
    assert abs(oce.item() - 1.2475949526) < 1e-6
    assert abs(reg.item() - 6.41421e-5) < 1e-9
    g = a.grad.cpu()[0]
    assert torch.allclose(g[0], torch.tensor([0.049257, 0.065676]), atol=1e-5)
    assert torch.allclose(g[2], torch.tensor([-0.26813, 0.0]), atol=1e-5)


def test_adam_matches_torch(device):
    torch.manual_seed(0)
    shapes = [(7, 3, 3, 3), (7,), (5, 7, 1, 1), (5,)]
    ps = [torch.randn(s) for s in shapes]
    ref_p = [p.clone().requires_grad_(True) for p in ps]
    ref_opt = torch.optim.Adam(ref_p, lr=4e-5, weight_decay=0.01)
    flat = torch.cat([p.reshape(-1) for p in ps]).to(device)
    flat_g = torch.zeros_like(flat)
    got_p, off = [], 0
    for p in ps:
        q = torch.nn.Parameter(flat[off:off + p.numel()].view(p.shape))
        q.grad = flat_g[off:off + p.numel()].view(p.shape)
        got_p.append(q)
        off += p.numel()
    opt = Adam(got_p, lr=4e-5, weight_decay=0.01)
    for step in range(5):
        gs = [torch.randn(s) * 10 ** (step - 2) for s in shapes]
        for p, q, g in zip(ref_p, got_p, gs):
            p.grad = g.clone()
            q.grad.copy_(g)
        ref_opt.step()
        opt.step()
    for p, q in zip(ref_p, got_p):
        assert torch.allclose(q.detach().cpu(), p.detach(), rtol=1e-6, atol=1e-7)
    sd = opt.state_dict()
    ref_sd = ref_opt.state_dict()
    assert set(sd["state"][0].keys()) == set(ref_sd["state"][0].keys())
    assert torch.allclose(sd["state"][0]["exp_avg"].cpu(), ref_sd["state"][0]["exp_avg"], rtol=1e-5, atol=1e-8)
    # a torch.optim.Adam state dict loads into ours
    opt.load_state_dict(ref_sd)
    # ... also one written by an older torch, whose `step` is a Python int
    for st in ref_sd["state"].values():
        st["step"] = int(st["step"].item()) if torch.is_tensor(st["step"]) else int(st["step"])
    opt.load_state_dict(ref_sd)
    for q in got_p:
        q.grad.zero_()
    opt.step()
    assert all(torch.is_tensor(st["step"]) and st["step"].item() == 6.0 for st in opt.state.values())


def test_adam_undo_takes_back_only_what_the_guarded_step_advanced(device):
    """step(guard=...) with a positive guard leaves parameters and moments alone; undo_step() then
    takes back the counters of exactly the parameters that step advanced — one without a gradient
    (frozen / unused) keeps its count, so the counters never drift apart (round-2 advisor finding)."""
    torch.manual_seed(0)
    a = torch.nn.Parameter(torch.randn(6, device=device))
    b = torch.nn.Parameter(torch.randn(4, device=device))
    opt = Adam([a, b], lr=1e-2, weight_decay=0.01)
    a.grad, b.grad = torch.randn(6, device=device), torch.randn(4, device=device)
    opt.step()
    opt.step()
    step_tensor = opt.state[a]["step"]
    assert opt.state[a]["step"].item() == opt.state[b]["step"].item() == 2.0
    b.grad = None                                     # frozen from here on
    before = a.detach().clone()
    guard = torch.ones(1, dtype=torch.float64, device=device)
    opt.step(guard=guard)                             # guarded off on the device
    torch.cuda.synchronize()
    assert torch.equal(a.detach(), before)
    assert opt.state[a]["step"].item() == 3.0 and opt.state[b]["step"].item() == 2.0
    opt.undo_step()
    assert opt.state[a]["step"].item() == 2.0 and opt.state[b]["step"].item() == 2.0
    assert opt.state[a]["step"] is step_tensor         # updated in place, as torch.optim.Adam does
    opt.undo_step()                                   # a second undo has nothing to take back
    assert opt.state[a]["step"].item() == 2.0
    guard.zero_()
    opt.step(guard=guard)
    torch.cuda.synchronize()
    assert not torch.equal(a.detach(), before) and opt.state[a]["step"].item() == 3.0


def test_out_of_range_coordinates_raise_like_the_reference(device):
    """unet.py:113-118 indexes outputs[b, :, y, x]: torch wraps -n..-1 and raises IndexError beyond;
    the kernels must never dereference such a row (they used to)."""
    out = torch.randn(2, 2, 5, 6)
    good = torch.tensor([[[1, 2], [-1, -5], [5, 4]]] * 2)          # x in [-6, 6), y in [-5, 5)
    ref = O.select_and_add_coordinates(out, good)
    got = UNetModel.select_and_add_coordinates(out.to(device), good.to(device))
    assert torch.allclose(got.cpu(), ref, atol=1e-6)
    for bad_row in ([6, 0], [0, 5], [-7, 0], [0, -6], [2 ** 40, 1]):
        bad = good.clone()
        bad[1, 2] = torch.tensor(bad_row)
        with pytest.raises(IndexError):
            O.select_and_add_coordinates(out, bad)
        with pytest.raises(IndexError):
            UNetModel.select_and_add_coordinates(out.to(device), bad.to(device))
    # the fused train step: IndexError BEFORE the parameters move
    cfg = dict(in_channels=1, out_channels=2, num_fmaps=4, fmap_inc_factor=2, features_in_last_layer=8,
               downsampling_factors=[[2, 2]], num_spatial_dims=2)
    torch.manual_seed(0)
    model = get_model(**cfg).to(device)
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=2, device=device)
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=0.01)
    raw = torch.rand(1, 1, 40, 40)
    a = torch.tensor([[[3, 4], [5, 6]]])
    r = torch.tensor([[[4, 4], [24, 6]]])                          # output is 24 x 24: x = 24 is outside
    before = [p.detach().clone() for p in model.parameters()]
    with pytest.raises(IndexError):
        train_iteration((raw, a, r), model, crit, opt, device)
    for p, q in zip(model.parameters(), before):
        assert torch.equal(p.detach(), q)
    # (the update was already in the stream, guarded on the device by the bad-coordinate count: it did nothing,
    # moments included, and the step counters were taken back)
    for st in opt.state.values():
        assert st["step"].item() == 0.0 and not st["exp_avg"].any() and not st["exp_avg_sq"].any()
    r[0, 1, 0] = 23
    loss, _, _ = train_iteration((raw, a, r), model, crit, opt, device)
    assert np.isfinite(loss)
    assert all(st["step"].item() == 1.0 for st in opt.state.values())
    assert any(not torch.equal(p.detach(), q) for p, q in zip(model.parameters(), before))
    # after good steps a bad batch leaves the trained state where it was
    train_iteration((raw, a, r), model, crit, opt, device)
    trained = [p.detach().clone() for p in model.parameters()]
    moments = [st["exp_avg"].clone() for st in opt.state.values()]
    r[0, 1, 0] = 24
    with pytest.raises(IndexError):
        train_iteration((raw, a, r), model, crit, opt, device)
    assert all(torch.equal(p.detach(), q) for p, q in zip(model.parameters(), trained))
    assert all(torch.equal(st["exp_avg"], m) and st["step"].item() == 2.0 for st, m in zip(opt.state.values(), moments))


def _pairs(rng, B, out_shape, kappa, n_anchor, n_ref):
    nd = len(out_shape)
    anchors, refs = [], []
    for _ in range(B):
        a = np.stack([rng.integers(kappa, out_shape[::-1][d] - kappa + 1, size=n_anchor) for d in range(nd)], 1)
        a = np.repeat(a, n_ref, axis=0)
        off = rng.integers(-kappa + 1, kappa, size=a.shape)
        off[np.abs(off).sum(1) == 0] = 1
        anchors.append(a)
        refs.append(a + off)
    return (torch.from_numpy(np.stack(anchors).astype(np.int64)),
            torch.from_numpy(np.stack(refs).astype(np.int64)))


@pytest.mark.parametrize("nd", [2, 3])
def test_train_iteration_matches_oracle(nd, device):
    cfg = dict(in_channels=1, out_channels=nd, num_fmaps=8, fmap_inc_factor=3, features_in_last_layer=16,
               downsampling_factors=[[2] * nd], num_spatial_dims=nd)
    spatial = (44, 52) if nd == 2 else (28, 24, 32)
    B = 2
    torch.manual_seed(0)
    oracle = O.OracleUNetModel(**cfg)
    for _n, layer in oracle.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    model = get_model(**cfg)
    model.load_state_dict(oracle.state_dict())
    model = model.to(device)
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=nd, device=device)
    opt = Adam(model.parameters(), lr=4e-5, weight_decay=0.01)
    ref_opt = torch.optim.Adam(oracle.parameters(), lr=4e-5, weight_decay=0.01)
    rng = np.random.default_rng(1)
    out_shape = tuple(s - 16 for s in spatial)
    for step in range(3):
        raw = torch.rand(B, 1, *spatial)
        anchor, reference = _pairs(rng, B, out_shape, 3, 50, 7)
        l_ref, o_ref, off_ref = O.train_step(oracle, ref_opt, raw, anchor, reference, 10.0, 1e-5)
        l, o, off = train_iteration((raw, anchor, reference), model, crit, opt, device)
        assert abs(l - l_ref) < 1e-4 * max(1.0, abs(l_ref)), (step, l, l_ref)
        assert abs(o - o_ref) < 1e-4 * max(1.0, abs(o_ref))
        assert (off.cpu() - off_ref.detach()).abs().max().item() < 1e-4, step
    for (n, po), (_, pm) in zip(oracle.named_parameters(), model.named_parameters()):
        assert torch.allclose(pm.detach().cpu(), po.detach(), atol=5e-5), n


def test_autograd_path_equals_fused_path(device):
    cfg = dict(in_channels=1, out_channels=2, num_fmaps=8, fmap_inc_factor=3, features_in_last_layer=16,
               downsampling_factors=[[2, 2]], num_spatial_dims=2)
    torch.manual_seed(0)
    m1 = get_model(**cfg).to(device)
    m2 = get_model(**cfg).to(device)
    m2.load_state_dict(m1.state_dict())
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=2, device=device)
    rng = np.random.default_rng(3)
    raw = torch.rand(2, 1, 44, 52)
    anchor, reference = _pairs(rng, 2, (28, 36), 3, 40, 5)
    o1 = Adam(m1.parameters(), lr=1e-3, weight_decay=0.01)
    o2 = Adam(m2.parameters(), lr=1e-3, weight_decay=0.01)
    l1, _, _ = train_iteration((raw, anchor, reference), m1, crit, o1, device)
    # generic composition through autograd
    off = m2(raw.to(device))
    ea = m2.select_and_add_coordinates(off, anchor.to(device))
    er = m2.select_and_add_coordinates(off, reference.to(device))
    loss, _, _ = crit(ea, er)
    o2.zero_grad()
    loss.backward()
    o2.step()
    assert abs(loss.item() - l1) < 1e-4 * max(1.0, abs(l1))
    for p1, p2 in zip(m1.parameters(), m2.parameters()):
        assert torch.allclose(p1, p2, atol=1e-5)


@pytest.mark.parametrize("nd", [2, 3])
def test_train_iteration_matches_real_reference_golden(nd, device):
    """g9: four iterations of the REAL cellulus.train.train_iteration (tests/golden/make_golden.py)
    vs the fused HIP step: losses, offsets and the parameters after four Adam updates."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_train_iteration.npz"))
    model = get_model(in_channels=1, out_channels=nd, num_fmaps=4, fmap_inc_factor=2, features_in_last_layer=8,
                      downsampling_factors=[[2] * nd], num_spatial_dims=nd)
    pre = f"w{nd}/init/"
    model.load_state_dict({k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}, strict=True)
    model = model.to(device)
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=nd, device=device)
    opt = Adam(model.parameters(), lr=float(g["lr"]), weight_decay=0.01)
    for it in range(4):
        b = tuple(torch.from_numpy(g[f"b{nd}/{it}/{k}"]) for k in ("raw", "anchor", "reference"))
        loss, oce, offsets = train_iteration(b, model, crit, opt, device)
        ref_loss, ref_oce = g[f"losses{nd}"][it]
        assert abs(loss - ref_loss) < 1e-4 * max(1.0, abs(ref_loss)), (it, loss, ref_loss)
        assert abs(oce - ref_oce) < 1e-4 * max(1.0, abs(ref_oce))
        assert np.abs(offsets.cpu().numpy() - g[f"b{nd}/{it}/offsets"]).max() < 1e-4     # north_star tolerance
    post = f"w{nd}/final/"
    moved = 0.0
    for k, v in model.state_dict().items():
        ref = g[post + k]
        moved = max(moved, float(np.abs(ref - g[pre + k]).max()))
        # parameters travel ~4 * lr = 4e-3; agree to 1 % of that
        assert np.abs(v.cpu().numpy() - ref).max() < 4e-5, k
    assert moved > 1e-3


@pytest.mark.parametrize("nd", [2, 3])
def test_device_pair_sampler_has_the_reference_distribution(nd, device):
    """clx_sample_pairs (opt-in, CLX_DEVICE_PAIRS=1): the support and the structure of
    ZarrDataset.sample_coordinates — anchors in [kappa, out - kappa] per column, each repeated
    num_references times, reference - anchor in the open kappa-ball without the origin — uniform use
    of the offsets, reproducible per (seed, step), and usable by the fused train step."""
    from cellulus_amd.datasets.zarr_dataset import DevicePairSampler, ZarrDataset

    ds = ZarrDataset.__new__(ZarrDataset)
    ds.num_spatial_dims, ds.kappa, ds.density = nd, 6.0, 0.1
    ds.output_shape = (40, 48) if nd == 2 else (24, 20, 28)
    ds.unbiased_shape = tuple(int(s - 2 * ds.kappa) for s in ds.output_shape)
    sampler = DevicePairSampler(ds, device, seed=123)
    B = 4
    a, r = sampler.sample(B, step=5)
    na, nr = ds.get_num_anchors(), ds.get_num_references()
    assert a.shape == r.shape == (B, na * nr, nd) and a.dtype == torch.int64
    a_c, r_c = a.cpu().numpy(), r.cpu().numpy()
    np.random.seed(0)
    ref_a, ref_r = ds.sample_coordinates()                      # the reference's sampler: same shapes / support
    assert ref_a.shape == a_c.shape[1:]
    for d in range(nd):
        assert a_c[..., d].min() >= 6 and a_c[..., d].max() <= ds.output_shape[d] - 6
        assert a_c[..., d].min() == 6 and a_c[..., d].max() == ds.output_shape[d] - 6     # both ends are reached
    blocks = a_c.reshape(B, na, nr, nd)
    assert (blocks == blocks[:, :, :1]).all()                   # np.repeat structure
    off = r_c - a_c
    d2 = (off ** 2).sum(-1)
    assert d2.max() < 36 and d2.min() >= 1
    table = sampler.offsets.cpu().numpy()
    ref_off = ref_r - ref_a
    assert {tuple(o) for o in ref_off} <= {tuple(o) for o in table}
    # uniform over the table: every offset used, counts within 6 sigma of the mean
    keys = (off.reshape(-1, nd) + 6) @ (13 ** np.arange(nd))
    counts = np.bincount(keys, minlength=13 ** nd)
    used = counts[(table + 6) @ (13 ** np.arange(nd))]
    mean = off.reshape(-1, nd).shape[0] / len(table)
    assert counts.sum() == used.sum()                            # nothing outside the table
    if mean >= 20:                                               # enough draws per offset to say "uniform"
        assert used.min() > 0 and np.abs(used - mean).max() < 6 * np.sqrt(mean) + 1
    # anchors of different batch rows / steps differ, the same (seed, step) repeats
    a2, r2 = sampler.sample(B, step=5)
    assert torch.equal(a, a2) and torch.equal(r, r2)
    a3, _ = sampler.sample(B, step=6)
    assert not torch.equal(a, a3) and not np.array_equal(a_c[0], a_c[1])
    if nd == 2:
        cfg = dict(in_channels=1, out_channels=2, num_fmaps=4, fmap_inc_factor=2, features_in_last_layer=8,
                   downsampling_factors=[[2, 2]], num_spatial_dims=2)
        model = get_model(**cfg).to(device)
        crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=2, device=device)
        opt = Adam(model.parameters(), lr=1e-3, weight_decay=0.01)
        # coordinate column 0 ranges over output_shape[0] but indexes the LAST axis (the reference's
        # convention): a square output keeps every pair inside
        ds.output_shape = (40, 40)
        ds.unbiased_shape = (28, 28)
        sq = DevicePairSampler(ds, device, seed=1)
        a4, r4 = sq.sample(2, step=0)
        loss, _, _ = train_iteration((torch.rand(2, 1, 56, 56), a4, r4), model, crit, opt, device)
        assert np.isfinite(loss)


@pytest.mark.parametrize("nd", [2, 3])
def test_deterministic_mode_is_bit_reproducible(nd, device, monkeypatch):
    """CLX_DETERMINISTIC=1 (opt-in): two runs of ten fused train steps from the same weights on the same
    batches end with bit-identical losses, gradients and parameters — as the reference's CPU path does
    (cellulus/train.py:177-179).  Weight-gradient slices add in slice order, bias gradients are ordered
    column sums, the pair-gradient scatter is fixed-point.  The result equals the default mode's to
    rounding (same arithmetic, another summation order).  Wide enough for Winograd layers, the sub-pixel
    upsample convolution and split-K slices."""
    cfg = dict(in_channels=1, out_channels=nd, num_fmaps=64 if nd == 2 else 32, fmap_inc_factor=2,
               features_in_last_layer=64, downsampling_factors=[[2] * nd], num_spatial_dims=nd)
    spatial = (76, 84) if nd == 2 else (28, 28, 32)
    out_shape = tuple(s - 16 for s in spatial)
    rng = np.random.default_rng(5)
    B = 3
    raw = torch.rand(B, 1, *spatial)
    anchor, reference = _pairs(rng, B, out_shape, 4, 150, 20)

    def run(det):
        monkeypatch.setenv("CLX_DETERMINISTIC", "1" if det else "0")
        torch.manual_seed(11)
        model = get_model(**cfg).to(device)
        for layer in model.modules():
            if isinstance(layer, torch.nn.modules.conv._ConvNd):
                torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
        crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=nd, device=device)
        opt = Adam(model.parameters(), lr=4e-5, weight_decay=0.01)       # train_config.py's default rate
        losses = [train_iteration((raw, anchor, reference), model, crit, opt, device)[0] for _ in range(10)]
        plan = next(iter(model._plans.values()))
        assert plan.deterministic == det and (not plan.chains if det else True)
        torch.cuda.synchronize()
        return losses, model._flat.detach().clone(), model._flat_grad.detach().clone()

    l1, p1, g1 = run(True)
    l2, p2, g2 = run(True)
    assert l1 == l2
    assert torch.equal(g1, g2) and torch.equal(p1, p2)
    l0, p0, _g0 = run(False)
    np.testing.assert_allclose(l1, l0, rtol=2e-4)
    # (an element whose gradient is rounding noise moves by +-lr per Adam step in either mode: 2 x 10 x 4e-5 apart at most)
    assert (p1 - p0).abs().max().item() < 1e-3
    assert (p1 - p0).abs().mean().item() < 2e-5


def test_ordered_column_sums_and_fixed_point_scatter_through_the_c_abi(device):
    """clx_colsum_ordered / clx_oce_pairs_fused_det: same values as the float64 reference, and identical bits
    from call to call."""
    from cellulus_amd import _clx

    st = _clx.stream_ptr(device)
    lib = _clx.load()
    torch.manual_seed(0)
    for M, N, ld in ((1000, 64, 64), (70001, 3, 4), (513, 768, 768), (40000, 1100, 1100)):
        x = torch.randn(M, ld, device=device)
        scratch = torch.empty(int(lib.clx_colsum_scratch_bytes(N)), dtype=torch.uint8, device=device)
        outs = []
        for _ in range(2):
            out = torch.zeros(N, device=device)             # the call ADDS (as the atomic bias path does)
            _clx.call("clx_colsum_ordered", _clx.ptr(x), ld, M, N, _clx.ptr(out), _clx.ptr(scratch), st)
            outs.append(out)
        assert torch.equal(outs[0], outs[1])
        twice = outs[1].clone()
        _clx.call("clx_colsum_ordered", _clx.ptr(x), ld, M, N, _clx.ptr(twice), _clx.ptr(scratch), st)
        assert torch.equal(twice, outs[0] + outs[0])
        ref = x[:, :N].double().sum(0)
        assert ((outs[0].double() - ref).abs() / (ref.abs() + M ** 0.5)).max().item() < 1e-5
    # the pair loss: against the atomic-float kernel and the oracle
    B, ND, Y, X = 2, 2, 40, 44
    rng = np.random.default_rng(1)
    offsets = torch.randn(B, ND, Y, X, device=device)
    anchor, reference = _pairs(rng, B, (Y, X), 4, 60, 31)
    a_d, r_d = anchor.to(device), reference.to(device)
    scratch = torch.empty(int(lib.clx_oce_pairs_det_scratch_bytes(B, ND, Y * X)), dtype=torch.uint8, device=device)
    res = []
    for _ in range(2):
        d = torch.full_like(offsets, float("nan"))
        sums = torch.zeros(4, dtype=torch.float64, device=device)
        _clx.call("clx_oce_pairs_fused_det", _clx.ptr(offsets), _clx.ptr(a_d), _clx.ptr(r_d), _clx.ptr(d),
                  _clx.ptr(sums), B, anchor.shape[1], ND, 1, Y, X, 10.0, 1e-5, _clx.ptr(scratch), st)
        res.append((d, sums.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    # ... and the loss sums ADD up over calls (half batches of a step), as clx_oce_pairs_fused's do
    d = torch.empty_like(offsets)
    sums = res[1][1].clone()
    _clx.call("clx_oce_pairs_fused_det", _clx.ptr(offsets), _clx.ptr(a_d), _clx.ptr(r_d), _clx.ptr(d),
              _clx.ptr(sums), B, anchor.shape[1], ND, 1, Y, X, 10.0, 1e-5, _clx.ptr(scratch), st)
    assert torch.equal(sums, 2 * res[0][1]) and torch.equal(d, res[0][0])
    d0 = torch.zeros_like(offsets)
    s0 = torch.zeros(4, dtype=torch.float64, device=device)
    _clx.call("clx_oce_pairs_fused", _clx.ptr(offsets), _clx.ptr(a_d), _clx.ptr(r_d), _clx.ptr(d0), _clx.ptr(s0),
              B, anchor.shape[1], ND, 1, Y, X, 10.0, 1e-5, st)
    assert torch.allclose(res[0][0], d0, atol=1e-5) and torch.allclose(res[0][1], s0, rtol=1e-9)
