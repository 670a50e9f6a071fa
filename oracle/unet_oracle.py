"""Oracle (test infrastructure): CPU restatement of the U-Net + head, the
embedding gather and the OCE loss in plain PyTorch fp32.

Follows:
  * cellulus/models/unet.py:24-63   (backbone arguments, 1x1 head)
  * cellulus/models/unet.py:69-100  (train / infer forward)
  * cellulus/models/unet.py:108-124 (select_and_add_coordinates)
  * cellulus/predict.py:21-135      (predict_scan: set_infer, the dry-run forward on zeros that
    fixes the output shape — and consumes 2 * num_infer_iterations torch.rand draws —, then the
    tile scan; gunpowder's Scan order itself is restated, gunpowder is absent)
  * cellulus/criterions/oce_loss.py:45-63 (OCE loss)
  * funlib.learn.torch.models.UNet @ f36decaf (pyproject.toml:30) — NOT present
    under /root/reference; restated from its published architecture (ConvPass =
    valid Conv + ReLU per kernel size, MaxPool down, nearest Upsample,
    crop_to_factor, centre-cropped skip concatenated before the upsampled
    tensor).  PARITY UNPINNED for the backbone topology: the reference holds no
    golden vector for it; only the (crop - 16) output-shape contract
    (cellulus/datasets/zarr_dataset.py:94) is pinned.
The head, the infer-mode noise loop, the gather and the loss are pinned against
the real reference classes by tests/golden (see tests/golden/make_golden.py).
"""

import math

import torch
import torch.nn as nn


class ConvPass(nn.Module):
    def __init__(self, cin, cout, kernel_sizes, nd):
        super().__init__()
        conv = {2: nn.Conv2d, 3: nn.Conv3d}[nd]
        layers = []
        for k in kernel_sizes:
            layers.append(conv(cin, cout, k))       # padding="valid"
            layers.append(nn.ReLU())
            cin = cout
        self.conv_pass = nn.Sequential(*layers)

    def forward(self, x):
        return self.conv_pass(x)


class Downsample(nn.Module):
    def __init__(self, factor, nd):
        super().__init__()
        self.factor = tuple(factor)
        self.nd = nd
        pool = {2: nn.MaxPool2d, 3: nn.MaxPool3d}[nd]
        self.down = pool(self.factor, stride=self.factor)

    def forward(self, x):
        for d in range(1, self.nd + 1):
            if x.size()[-d] % self.factor[-d] != 0:
                raise RuntimeError(
                    "Can not downsample shape %s with factor %s, mismatch in spatial dimension %d"
                    % (x.size(), self.factor, self.nd - d))
        return self.down(x)


class Upsample(nn.Module):
    def __init__(self, factor, nd, crop_factor, next_conv_kernel_sizes):
        super().__init__()
        self.nd = nd
        self.crop_factor = crop_factor
        self.next_conv_kernel_sizes = next_conv_kernel_sizes
        self.up = nn.Upsample(scale_factor=tuple(factor), mode="nearest")

    def crop_to_factor(self, x, factor, kernel_sizes):
        shape = x.size()
        spatial = shape[-self.nd:]
        conv_crop = tuple(sum(ks[d] - 1 for ks in kernel_sizes) for d in range(self.nd))
        ns = (int(math.floor(float(s - c) / f)) for s, c, f in zip(spatial, conv_crop, factor))
        target = tuple(n * f + c for n, c, f in zip(ns, conv_crop, factor))
        if target != tuple(spatial):
            assert all(t > c for t, c in zip(target, conv_crop)), "feature map too small"
            return self.crop(x, target)
        return x

    def crop(self, x, shape):
        x_target = x.size()[:-self.nd] + tuple(shape)
        offset = tuple((a - b) // 2 for a, b in zip(x.size(), x_target))
        slices = tuple(slice(o, o + s) for o, s in zip(offset, x_target))
        return x[slices]

    def forward(self, f_left, g_out):
        g_up = self.up(g_out)
        g_cropped = self.crop_to_factor(g_up, self.crop_factor, self.next_conv_kernel_sizes)
        f_cropped = self.crop(f_left, g_cropped.size()[-self.nd:])
        return torch.cat([f_cropped, g_cropped], dim=1)


class OracleUNet(nn.Module):
    """Backbone with the constructor keywords cellulus/models/unet.py:24-51 passes."""

    def __init__(self, in_channels, num_fmaps, fmap_inc_factor, downsample_factors,
                 kernel_size_down=None, kernel_size_up=None, activation="ReLU",
                 num_fmaps_out=None, padding="valid", constant_upsample=True, **_):
        super().__init__()
        assert activation == "ReLU" and padding == "valid" and constant_upsample
        nd = len(downsample_factors[0]) if len(downsample_factors) else len(kernel_size_down[0][0])
        self.nd = nd
        self.num_levels = len(downsample_factors) + 1
        L = self.num_levels - 1
        crop_factors, prod = [], None
        for f in downsample_factors[::-1]:
            prod = list(f) if prod is None else [a * b for a, b in zip(f, prod)]
            crop_factors.append(prod)
        crop_factors = crop_factors[::-1]
        self.l_conv = nn.ModuleList([
            ConvPass(in_channels if i == 0 else num_fmaps * fmap_inc_factor ** (i - 1),
                     num_fmaps * fmap_inc_factor ** i, kernel_size_down[i], nd)
            for i in range(self.num_levels)])
        self.l_down = nn.ModuleList([Downsample(downsample_factors[i], nd) for i in range(L)])
        self.r_up = nn.ModuleList([nn.ModuleList([
            Upsample(downsample_factors[i], nd, crop_factors[i], kernel_size_up[i]) for i in range(L)])])
        self.r_conv = nn.ModuleList([nn.ModuleList([
            ConvPass(num_fmaps * fmap_inc_factor ** i + num_fmaps * fmap_inc_factor ** (i + 1),
                     num_fmaps * fmap_inc_factor ** i if (num_fmaps_out is None or i != 0) else num_fmaps_out,
                     kernel_size_up[i], nd)
            for i in range(L)])])

    def rec_forward(self, level, f_in):
        i = self.num_levels - level - 1
        f_left = self.l_conv[i](f_in)
        if level == 0:
            return f_left
        g_in = self.l_down[i](f_left)
        g_out = self.rec_forward(level - 1, g_in)
        f_right = self.r_up[0][i](f_left, g_out)
        return self.r_conv[0][i](f_right)

    def forward(self, x):
        return self.rec_forward(self.num_levels - 1, x)


class OracleUNetModel(nn.Module):
    """cellulus/models/unet.py:9-124 with the backbone above (CPU, fp32)."""

    def __init__(self, in_channels, out_channels, num_fmaps, fmap_inc_factor,
                 features_in_last_layer, downsampling_factors, num_spatial_dims):
        super().__init__()
        nd = num_spatial_dims
        ks = [(3,) * nd, (1,) * nd, (1,) * nd, (3,) * nd]
        self.backbone = OracleUNet(
            in_channels=in_channels, num_fmaps=num_fmaps, fmap_inc_factor=fmap_inc_factor,
            downsample_factors=[tuple(f) for f in downsampling_factors], activation="ReLU",
            padding="valid", num_fmaps_out=features_in_last_layer,
            kernel_size_down=[ks] * (len(downsampling_factors) + 1),
            kernel_size_up=[ks] * len(downsampling_factors), constant_upsample=True)
        conv = {2: nn.Conv2d, 3: nn.Conv3d}[nd]
        self.head = nn.Sequential(conv(features_in_last_layer, features_in_last_layer, 1), nn.ReLU(),
                                  conv(features_in_last_layer, out_channels, 1))
        self.mode = "train"

    def set_infer(self, p_salt_pepper, num_infer_iterations, device=None):
        self.mode = "infer"
        self.p_salt_pepper = p_salt_pepper
        self.num_infer_iterations = num_infer_iterations

    def forward(self, raw, noise=None):
        if self.mode == "train":
            return self.head(self.backbone(raw))
        embeddings = []
        for sample in range(raw.shape[0]):          # unet.py:75-98
            raw_sample = raw[sample:sample + 1]
            preds, t = [], 0
            for val in [0.5, 1.0]:
                for _ in range(self.num_infer_iterations):
                    noisy = raw_sample.detach().clone()
                    rnd = torch.rand(*noisy.shape) if noise is None else noise[sample, t][None]
                    noisy[rnd <= self.p_salt_pepper] = val
                    preds.append(self.head(self.backbone(noisy))[0].detach())
                    t += 1
            std, mean = torch.std_mean(torch.stack(preds, dim=0), dim=0, keepdim=False, unbiased=False)
            std = std.sum(dim=0, keepdim=True)
            embeddings.append(torch.cat((mean, std), dim=0))
        return torch.stack(embeddings, dim=0)


def predict_scan(model, raw, crop_size, p_salt_pepper, num_infer_iterations, normalization_factor=1.0,
                 literal_dry_run=True):
    """cellulus/predict.py:9-135 for an in-memory raw array (S, C, *spatial) -> (S, D+1, *spatial)
    float64, on the CPU generator of torch exactly as the reference uses it:

    * predict.py:21-25  ``model.set_infer(...)`` FIRST;
    * predict.py:32-39  ``output_shape = model(zeros(1, C, *crop)).shape`` — the model is already in
      infer mode, so this dry run executes unet.py:75-88: 2 * num_infer_iterations forwards, each
      behind one ``torch.rand(1, C, *crop)``.  ``literal_dry_run=False`` makes the same draws without
      the forwards (same generator state, for sizes where 2N extra CPU forwards cost minutes);
    * predict.py:114-135  Normalize, reflect Pad by the context, Scan: tiles of ``crop_size``, stride
      = output tile, last tile of an axis shifted back inside (later tile overwrites), every tile
      one infer-mode forward of a batch of one (unet.py:73-100)."""
    import itertools

    import numpy as np

    model.set_infer(p_salt_pepper, num_infer_iterations)
    crop = tuple(int(c) for c in crop_size)
    S, C = raw.shape[0], raw.shape[1]
    spatial = tuple(raw.shape[2:])
    nd = len(spatial)
    zeros = torch.zeros((1, C) + crop, dtype=torch.float32)
    if literal_dry_run:
        with torch.no_grad():
            out_shape = tuple(model(zeros).shape)
    else:
        for val in [0.5, 1.0]:
            for _ in range(num_infer_iterations):
                torch.rand(*zeros.shape)
        model.mode = "train"
        with torch.no_grad():       # the shape without the noise loop (no draw in train mode)
            out_shape = (1, nd + 1) + tuple(model(torch.zeros((1, C) + crop)).shape[2:])
        model.mode = "infer"
    out_tile = out_shape[2:]
    context = tuple((c - o) // 2 for c, o in zip(crop, out_tile))
    result = np.zeros((S, nd + 1) + spatial, dtype=np.float64)

    def offsets(size, tile):
        offs = list(range(0, size - tile + 1, tile))
        if offs[-1] + tile < size:
            offs.append(size - tile)
        return offs

    for s in range(S):
        img = raw[s].astype(np.float32) * np.float32(normalization_factor)
        img = np.pad(img, [(0, 0)] + [(c, c) for c in context], mode="reflect")
        for off in itertools.product(*[offsets(sz, t) for sz, t in zip(spatial, out_tile)]):
            sl = (slice(None),) + tuple(slice(o, o + c) for o, c in zip(off, crop))
            with torch.no_grad():
                e = model(torch.from_numpy(np.ascontiguousarray(img[sl]))[None])[0].numpy()
            result[(s, slice(None)) + tuple(slice(o, o + t) for o, t in zip(off, out_tile))] = e
    return result


def select_and_add_coordinates(outputs, coordinates):
    """cellulus/models/unet.py:108-124."""
    selections = []
    for output, coordinate in zip(outputs, coordinates):
        if output.ndim == 3:
            selection = output[:, coordinate[:, 1], coordinate[:, 0]]
        else:
            selection = output[:, coordinate[:, 2], coordinate[:, 1], coordinate[:, 0]]
        selection = selection.transpose(1, 0)
        selection = selection + coordinate
        selections.append(selection)
    return torch.stack(selections, dim=0)


def oce_loss(anchor, reference, temperature, regularization_weight):
    """cellulus/criterions/oce_loss.py:45-63 -> (loss, oce, reg)."""
    distance = (anchor - reference.detach()).norm(2, dim=-1)
    oce = (1 - (-distance.pow(2) / temperature).exp()).sum()
    reg = regularization_weight * anchor.norm(2, dim=-1).sum()
    return oce + reg, oce, reg


def train_step(model, optimizer, raw, anchor, reference, temperature, regularization_weight):
    """cellulus/train.py:160-180 on the CPU; returns (loss, oce, offsets)."""
    model.train()
    offsets = model(raw)
    ea = select_and_add_coordinates(offsets, anchor)
    er = select_and_add_coordinates(offsets, reference)
    loss, oce, _ = oce_loss(ea, er, temperature, regularization_weight)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.item(), oce.item(), offsets


# ---------------------------------------------------------------------------------------------
# The same convolutions as one matrix product per filter tap, so that the float64 oracle runs at dgemm
# speed at BASELINE sizes: torch's native float64 convolution (no oneDNN path) manages ~2-6
# GFLOP/s, i.e. minutes per 256^2 crop at 256 feature maps; MKL dgemm is 20-50x faster.  The
# arithmetic is nn.ConvNd's (valid cross-correlation + bias) up to summation order;
# tests/test_cpu_oracle_golden.py holds the two forms together in float64 (values and gradients).
# ---------------------------------------------------------------------------------------------
def conv_as_gemm(x, weight, bias):
    """valid N-d cross-correlation of x (B, C, *spatial) with weight (N, C, *k), + bias, as one
    dgemm per filter tap WITHOUT an im2col copy: with the image flattened to one axis, tap
    (dz, dy, dx) reads the same flat range shifted by dz*H*W + dy*W + dx, so every tap's operand is
    a column-offset view of the (C, D*H*W) matrix; the (few) flat positions that wrap around a row
    end are computed and dropped."""
    import itertools

    B, C = x.shape[0], x.shape[1]
    N, k = weight.shape[0], tuple(weight.shape[2:])
    spatial = tuple(x.shape[2:])
    nd = len(k)
    out_shape = tuple(s - kk + 1 for s, kk in zip(spatial, k))
    strides = [math.prod(spatial[d + 1:]) for d in range(nd)]
    length = sum((o - 1) * st for o, st in zip(out_shape, strides)) + 1      # last valid flat index + 1
    outs = []
    for b in range(B):
        flat = x[b].reshape(C, -1)
        acc = None
        for tap in itertools.product(*[range(kk) for kk in k]):
            off = sum(t * st for t, st in zip(tap, strides))
            part = torch.mm(weight[(slice(None), slice(None)) + tap], flat[:, off:off + length])
            acc = part if acc is None else acc + part
        full = torch.cat([acc, acc.new_zeros(N, out_shape[0] * strides[0] - length)], dim=1)
        full = full.reshape((N, out_shape[0]) + spatial[1:])
        outs.append(full[(slice(None), slice(None)) + tuple(slice(0, o) for o in out_shape[1:])])
    out = torch.stack(outs, dim=0)
    if bias is not None:
        out = out + bias.reshape((1, N) + (1,) * nd)
    return out


class gemm_convolutions:
    """Context manager: every nn.Conv2d / nn.Conv3d of `model` computes through conv_as_gemm."""

    def __init__(self, model):
        self.convs = [m for m in model.modules() if isinstance(m, nn.modules.conv._ConvNd)]

    def __enter__(self):
        for m in self.convs:
            assert m.padding in ((0,) * len(m.kernel_size), "valid") and set(m.stride) == {1} \
                and set(m.dilation) == {1} and m.groups == 1
            m.forward = (lambda x, m=m: conv_as_gemm(x, m.weight, m.bias))
        return self

    def __exit__(self, *exc):
        for m in self.convs:
            del m.forward
        return False


# ---------------------------------------------------------------------------------------------
# Gradients through ReLU / max-pool are discontinuous in the forward values: a pre-activation
# within rounding distance of 0 (or two window entries within rounding distance of each other)
# takes one branch in float32 and the other in float64, and that ONE decision moves the gradient
# by a whole term.  At BASELINE sizes (1.3e8 activations per crop) a few hundred such decisions
# differ between ANY float32 forward pass and the float64 one — the float32 CPU path of the
# reference included — which bounds the agreement of float32 and float64 gradients at ~1e-3
# (relative L2) however exact the kernels are.  To check the ARITHMETIC of a float32 backward
# pass against float64 at those sizes, the float64 oracle therefore takes the discrete decisions
# (ReLU gates, pooling winners) from the forward pass under test and computes everything else
# itself.
# ---------------------------------------------------------------------------------------------
class forced_decisions:
    """Context manager: the nn.ReLU modules of `model` multiply by the given 0/1 masks (in call
    order) and the max-pool modules gather the given flat window-winner indices (the
    `return_indices` convention of F.max_pool{2,3}d), instead of deciding on their own input."""

    def __init__(self, model, relu_masks, pool_indices):
        self.relus = [m for m in model.modules() if isinstance(m, nn.ReLU)]
        self.pools = [m for m in model.modules() if isinstance(m, Downsample)]
        self.relu_masks, self.pool_indices = list(relu_masks), list(pool_indices)
        assert len(self.relus) == len(self.relu_masks) and len(self.pools) == len(self.pool_indices)

    def __enter__(self):
        masks, idxs = iter(self.relu_masks), iter(self.pool_indices)

        def relu_forward(x):
            m = next(masks)
            assert m.shape == x.shape, (m.shape, x.shape)
            return x * m.to(x.dtype)

        def pool_forward(x):
            idx = next(idxs)
            flat = x.reshape(x.shape[0], x.shape[1], -1)
            return torch.gather(flat, 2, idx.reshape(idx.shape[0], idx.shape[1], -1)).reshape(idx.shape)

        for m in self.relus:
            m.forward = relu_forward
        for m in self.pools:
            m.forward = pool_forward
        return self

    def __exit__(self, *exc):
        for m in self.relus + self.pools:
            del m.forward
        return False
