"""GPU tier at BASELINE.json's FULL sizes (2-D 256^2 crops, num_fmaps=256, batch 8; 512^2 inference),
where the CPU oracle cannot run in test time: size-independent properties of the path instead —
sample independence, translation equivariance, determinism, linearity of the backward pass,
a decreasing loss, idempotence / ordering of the label maps — plus the full-size inference
post-processing against the (fast) C oracle."""

import numpy as np
import pytest
import torch

from cellulus_amd.criterions import get_loss
from cellulus_amd.models import get_model
from cellulus_amd.optim import Adam
from cellulus_amd.train import train_iteration
from oracle import infer_oracle as IO

pytestmark = pytest.mark.gpu

CFG2 = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64,
            downsampling_factors=[[2, 2]], num_spatial_dims=2)


@pytest.fixture(scope="module")
def model(device):
    torch.manual_seed(0)
    m = get_model(**CFG2)
    for _n, layer in m.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    return m.to(device)


def test_forward_is_deterministic_and_samples_are_independent(model, device):
    """No cross-sample operation anywhere: row i of a batch-8 forward equals the
    batch-1 forward of sample i BIT FOR BIT (every output element's contraction order is fixed
    by the K loop, not by the M tiling), and two launches agree bit for bit (no atomics)."""
    torch.manual_seed(1)
    raw = torch.rand(8, 1, 256, 256, device=device)
    with torch.no_grad():
        full = model(raw).clone()
        again = model(raw).clone()
        assert full.shape == (8, 2, 240, 240)
        assert torch.equal(full, again)
        for i in (0, 3, 7):
            single = model(raw[i:i + 1].contiguous())
            assert torch.equal(single[0], full[i]), f"sample {i}"
    assert torch.isfinite(full).all()


def test_inference_chunk_of_eight_528_tiles_equals_its_halves_bit_for_bit(model, device):
    """The benchmark's inference chunk: eight 528^2 copies per forward.  The Winograd operand of its first wide layer
    (36 x tiles x 256 channels) is 1.3 G floats, above what a 32-bit byte offset reaches — the batched GEMM launches
    address it as a 64-bit per-product base + 32-bit offsets INSIDE one product (csrc/conv_igemm.hip::clx_igemm_launch),
    on the same staging path as the four-copy forward, whose operands are small: the same bits, sample by sample."""
    torch.manual_seed(4)
    raw = torch.rand(8, 1, 528, 528, device=device)
    with torch.no_grad():
        full = model(raw).clone()
        assert full.shape == (8, 2, 512, 512) and torch.isfinite(full).all()
        for lo in (0, 4):
            half = model(raw[lo:lo + 4].contiguous())
            assert torch.equal(half, full[lo:lo + 4]), f"copies {lo}..{lo + 3}"


def test_benchmark_tile_through_changed_rows_equals_the_dense_forward_bit_for_bit(model, device, monkeypatch):
    """The benchmark's inference tile (528^2, 256 feature maps, 32 noisy copies at 1 % salt / pepper, chunks of eight on
    two streams): the 1x1 layers of the first level on the clean tile + the changed rows of each copy
    against every layer densely on every copy (CLX_SPARSE_NOISE=0) — the kernels and tile shapes the bench line is
    measured on — torch.equal on mean and std; 8-9 % of the rows are recomputed."""
    model.eval()
    model.set_infer(p_salt_pepper=0.01, num_infer_iterations=16, device=device)
    torch.manual_seed(7)
    raw = torch.rand(1, 1, 528, 528, device=device)
    noise = torch.rand(1, 32, 1, 528, 528, device=device)
    try:
        monkeypatch.setenv("CLX_SPARSE_NOISE", "0")
        dense = model.infer_on_device(raw, noise=noise).clone()
        monkeypatch.delenv("CLX_SPARSE_NOISE", raising=False)
        model._last_changed_rows = None
        sparse = model.infer_on_device(raw, noise=noise).clone()
        info = model._last_changed_rows
        assert info is not None and info["used"] and 0.07 < info["fraction"] < 0.10, info
        assert torch.equal(dense, sparse)
        monkeypatch.setenv("CLX_INFER_STREAMS", "1")
        assert torch.equal(dense, model.infer_on_device(raw, noise=noise))
    finally:
        model.train()
        model.mode = "train"


def test_translation_equivariance_at_multiples_of_the_downsampling(model, device):
    """crop_to_factor makes the valid U-Net equivariant to shifts that are multiples of the
    cumulative downsampling factor: shifting the input window by 4 px shifts the output by 4 px."""
    torch.manual_seed(2)
    big = torch.rand(1, 1, 260, 260, device=device)
    with torch.no_grad():
        a = model(big[:, :, :256, :256].contiguous()).clone()
        b = model(big[:, :, 4:, 4:].contiguous()).clone()
    diff = (a[:, :, 4:, 4:] - b[:, :, :-4, :-4]).abs().max().item()
    scale = a.abs().max().item()
    assert diff <= 1e-5 * max(1.0, scale), (diff, scale)


def test_backward_is_linear_in_the_output_gradient(model, device):
    """With the forward (and so every ReLU / max-pool gate) fixed, parameter gradients are linear
    in dL/d(output): grad(g1 + 2 g2) = grad(g1) + 2 grad(g2).  Split-K atomics make the sums
    order-dependent, so the comparison is relative (1e-4 of the gradient norm), not bitwise."""
    torch.manual_seed(3)
    raw = torch.rand(8, 1, 256, 256, device=device)

    def grads(g):
        model.zero_grad()
        out = model(raw)
        out.backward(g)
        return [p.grad.detach().clone() for p in model.parameters()]

    with torch.no_grad():
        shape = model(raw).shape
    g1 = torch.randn(shape, device=device)
    g2 = torch.randn(shape, device=device)
    ga, gb, gc = grads(g1), grads(g2), grads(g1 + 2 * g2)
    for a, b, c in zip(ga, gb, gc):
        ref = a + 2 * b
        assert torch.isfinite(c).all()
        assert (c - ref).norm().item() <= 1e-4 * max(ref.norm().item(), 1e-6)
    model.zero_grad()


def test_full_size_train_steps_reduce_the_loss(device):
    """Six fused steps (U-Net forward/backward, OCE loss on 150 040 pairs per crop, Adam) on one
    fixed full-size batch: finite, and the loss goes down."""
    torch.manual_seed(4)
    m = get_model(**CFG2)
    for _n, layer in m.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    m = m.to(device)
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=2, device=device)
    opt = Adam(m.parameters(), lr=4e-5, weight_decay=0.01)
    rng = np.random.default_rng(0)
    B, out, kappa, n_anchor, n_ref = 8, 240, 10, 4840, 31
    anchors = np.repeat(rng.integers(kappa, out - kappa + 1, size=(B, n_anchor, 2)), n_ref, axis=1)
    offs = rng.integers(-kappa + 1, kappa, size=anchors.shape)
    offs[np.abs(offs).sum(-1) == 0] = 1
    batch = (torch.rand(B, 1, 256, 256), torch.from_numpy(anchors.astype(np.int64)),
             torch.from_numpy((anchors + offs).astype(np.int64)))
    assert batch[1].shape == (8, 150040, 2)
    losses = [train_iteration(batch, m, crit, opt, device)[0] for _ in range(6)]
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses


def test_noise_statistics_of_identical_predictions(model, device):
    """infer mode with p_salt_pepper = 0: all 32 'noisy' forwards are the same image, so the mean is
    that forward (bit for bit: a mean of equal f32 values is exact) and the std channel is 0."""
    torch.manual_seed(5)
    raw = torch.rand(1, 1, 256, 256, device=device)
    with torch.no_grad():
        model.train()
        ref = model(raw).clone()
        model.eval()
        model.set_infer(p_salt_pepper=0.0, num_infer_iterations=16, device=device)
        noise = torch.rand(1, 32, 1, 256, 256, device=device) + 1.0      # never <= 0: nothing replaced
        emb = model.infer_on_device(raw, noise=noise)
        model.mode = "train"
        model.train()
    assert emb.shape == (1, 3, 240, 240)
    assert torch.allclose(emb[0, :2], ref[0], rtol=0, atol=1e-6)
    assert emb[0, 2].abs().max().item() <= 1e-6


def test_full_size_inference_postprocessing_matches_oracle(device):
    """BASELINE cfg-5 (512^2, ~100 objects, bandwidth 15, reduction_probability 0.1): Otsu +
    mean-shift detection + grow/shrink + size filter on the device == the C/numpy oracle, and
    the result is canonical: ids 1..n in raster order of first appearance, size filter idempotent."""
    from cellulus_amd.segment import grow_shrink_on_device
    from cellulus_amd.utils.mean_shift import mean_shift_on_device
    from cellulus_amd.utils.misc import label_on_device
    from cellulus_amd.utils.otsu import threshold_otsu

    mean, std = IO.synthetic_embeddings((512, 512), spacing=48, radius=12, noise=0.3, seed=1)
    mean_d, std_d = torch.from_numpy(mean[0]).to(device), torch.from_numpy(std).to(device)
    np.random.seed(1)
    thr = threshold_otsu(std_d)
    labels, _centers = mean_shift_on_device(mean_d.clone(), std_d, 15.0, 0.1, thr, None)
    seg = labels.clone()
    grow_shrink_on_device(seg, 3, 6)
    out, n = label_on_device(seg, 70)
    np.random.seed(1)
    ref_thr = IO.threshold_otsu(std)
    assert thr == ref_thr
    ref = IO.mean_shift_segmentation(mean.copy(), std, 15.0, 70, 0.1, ref_thr, None)
    ref_seg = IO.size_filter(IO.grow_shrink(ref, 3, 6), 70)
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got, ref_seg)
    assert int(n.item()) == got.max() >= 90
    first = [np.flatnonzero(got.ravel() == k)[0] for k in range(1, got.max() + 1)]
    assert first == sorted(first)                        # raster order of first appearance
    again, n2 = label_on_device(out.clone(), 70)         # idempotent
    assert torch.equal(again, out) and int(n2.item()) == int(n.item())


def test_full_size_3d_configuration(device):
    """BASELINE cfg-4 (3-D 64^3 crops, num_fmaps=64, batch 8): sample independence and determinism of
    the forward pass bit for bit (Winograd in (y, x) with the z taps inside the GEMMs, the one-channel
    first-layer kernels), then four fused train steps with 2 418 pairs per crop: finite, loss down."""
    cfg3 = dict(in_channels=1, out_channels=3, num_fmaps=64, fmap_inc_factor=3, features_in_last_layer=64,
                downsampling_factors=[[2, 2, 2]], num_spatial_dims=3)
    torch.manual_seed(6)
    m = get_model(**cfg3)
    for _n, layer in m.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    m = m.to(device)
    raw = torch.rand(8, 1, 64, 64, 64, device=device)
    with torch.no_grad():
        full = m(raw).clone()
        assert full.shape == (8, 3, 48, 48, 48) and torch.isfinite(full).all()
        assert torch.equal(full, m(raw))
        for i in (0, 5):
            assert torch.equal(m(raw[i:i + 1].contiguous())[0], full[i]), f"sample {i}"
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=3, device=device)
    opt = Adam(m.parameters(), lr=4e-5, weight_decay=0.01)
    rng = np.random.default_rng(1)
    B, out, kappa, n_anchor, n_ref = 8, 48, 10, 78, 31
    anchors = np.repeat(rng.integers(kappa, out - kappa + 1, size=(B, n_anchor, 3)), n_ref, axis=1)
    offs = rng.integers(-kappa + 1, kappa, size=anchors.shape)
    offs[np.abs(offs).sum(-1) == 0] = 1
    batch = (raw.cpu(), torch.from_numpy(anchors.astype(np.int64)), torch.from_numpy((anchors + offs).astype(np.int64)))
    assert batch[1].shape == (8, 2418, 3)
    losses = [train_iteration(batch, m, crit, opt, device)[0] for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
