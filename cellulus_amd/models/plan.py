"""Launch plan of the U-Net + head on libclx (forward, backward, inference).

The topology is the one ``cellulus/models/unet.py:24-63`` requests from
``funlib.learn.torch.models.UNet`` (valid convolutions ``[3,1,1,3]`` + ReLU per
level, max-pool down, nearest ``constant_upsample`` up, centre-cropped skip
concatenated *before* the upsampled tensor) followed by the 1x1 head.  funlib
is not vendored in the reference; the restated rules (``crop_to_factor``,
channel counts, module names) are documented in SURVEY.md §3.4.

Everything device-side is channels-last f32 with channel counts padded to a
multiple of 4; the padded channels stay zero for the life of a buffer.
"""

import ctypes
import math
import os
from dataclasses import dataclass, field
from typing import List, Tuple

import torch

from .. import _clx
from .._clx import ClxConvDesc, ClxSrc


def pad4(c: int) -> int:
    return (c + 3) // 4 * 4


# Winograd is used for 3x3 layers whose channel counts (both sides) reach this value: below it the
# extra HBM traffic of the transformed tensors and the short contraction (K = C per batched GEMM)
# outweigh the fewer MFMA FLOPs (measured at the benchmark config: 64 beats 128 by 2 %).
# CLX_WINOGRAD=0 forces the direct implicit-GEMM kernels everywhere; CLX_WINOGRAD_TILE selects
# F(2x2, 3x3) (2.25x fewer multiplications, error ~7e-7 of the output range on a 768-channel
# layer) or F(4x4, 3x3) (4x fewer, ~5e-6; the direct kernel: ~4e-7).
WINO_MIN_CHANNELS = int(os.environ.get("CLX_WINOGRAD_MIN_CHANNELS", "64"))
# per-layer algorithm code = clx_conv_algo: 0 direct, 1 Winograd F(2x2), 2 Winograd F(4x4)
# (3 = F(4x4) with the transforms inside the product kernel, forward only: clx_conv_algo CLX_ALGO_WINOGRAD4_FUSED)
WINO_TAPS = {1: 16, 2: 36, 3: 36}
WINO_PACK_FWD = {1: 2, 2: 4, 3: 7}  # clx_pack_mode
WINO_PACK_DGRAD = {1: 3, 2: 5}
WINO_TILE = {1: 2, 2: 4, 3: 4}


# 3-D layers (F(4x4) in (y, x) per z plane, the z taps inside the batched GEMMs): K = 3 * C per GEMM, so
# it pays from fewer channels than in 2-D
WINO_MIN_CHANNELS_3D = int(os.environ.get("CLX_WINOGRAD_MIN_CHANNELS_3D", "64"))


def winograd_code() -> int:
    return 2 if os.environ.get("CLX_WINOGRAD_TILE", "4") == "4" else 1


def wino_taps(code, kernel):
    """packed-weight / weight-gradient planes of a Winograd layer: a^2 transform points x z taps."""
    return WINO_TAPS[code] * kernel[0]


def wino_min_channels(kernel):
    return WINO_MIN_CHANNELS_3D if kernel[0] > 1 else WINO_MIN_CHANNELS


# The arithmetic of the plain products (1x1 layers, the transform-domain products of the 2-D Winograd layers; forward, data
# gradient and weight gradient), clx_conv_precision:
#   "f32x3bf16" (default since round 6): every float32 operand split EXACTLY into three bfloat16 pieces, six exact products
#       per float32 product accumulated in float32 on the bf16 matrix cores (csrc/gemm_sp.hip).  Results
#       are float32; distance from the float64 oracle at a trained network's output scale 7.2e-5 (float32 MFMA: 7.3e-5).
#   "f32": float32 MFMA everywhere (v_mfma_f32_32x32x2_f32), the only arithmetic of rounds 1-5; CLX_PRECISION=f32.
DEFAULT_PRECISION = "f32x3bf16"


def precision_name() -> str:
    name = os.environ.get("CLX_PRECISION", "") or DEFAULT_PRECISION
    if name not in ("f32", "f32x3bf16"):
        raise ValueError(f"CLX_PRECISION must be 'f32' or 'f32x3bf16', got {name!r}")
    return name


def precision_code() -> int:
    """clx_conv_precision: 0 = float32 MFMA, 1 = the three-way bfloat16 split (CLX_PREC_F32X3BF16).  The run-to-run
    reproducible mode (CLX_DETERMINISTIC=1) exists in float32 only and selects it."""
    if os.environ.get("CLX_DETERMINISTIC", "0") == "1":
        return 0
    return 1 if precision_name() == "f32x3bf16" else 0


def winograd_enabled() -> bool:
    return os.environ.get("CLX_WINOGRAD", "1") != "0"


# The fused forms keep the products' results on chip: 36 x 32 x 64 accumulators per workgroup, i.e. 10.7 FLOP per
# byte of operands from L2 where the 128 x 128 tiles of the batched GEMMs have 32 — they top out near 110 TFLOP/s.  That
# beats the three-launch form where ITS GEMMs are short (K = C <= 256: 78-90 TFLOP/s with the transforms) or narrow
# (N = 64: HBM-bound on V and M), and loses at C = 768 (113 TFLOP/s with the transforms; tools/exp/fused_bench.py,
# DESIGN.md 3.1g).
FUSED_MAX_CHANNELS = int(os.environ.get("CLX_WINO_FUSED_MAX_CHANNELS", "256"))


def fused_pays(cin_pad: int, cout: int) -> bool:
    return cout <= 64 or cin_pad <= FUSED_MAX_CHANNELS


def fused_wanted(keep_activations: bool) -> bool:
    """2-D F(4x4) forward layers as ONE launch each (csrc/wino_fused.hip: the transformed tensors never reach HBM).
    CLX_WINO_FUSED=0 keeps the three-launch form everywhere; the training plans (which keep the transformed input for
    the weight gradient) take it with CLX_WINO_FUSED_TRAIN=1 only."""
    if os.environ.get("CLX_WINO_FUSED", "1") == "0":
        return False
    return (not keep_activations) or os.environ.get("CLX_WINO_FUSED_TRAIN", "0") == "1"


@dataclass
class Source:
    """One input of a convolution: a stored tensor seen through crop/upsample."""

    tensor: str                      # buffer name
    channels: int                    # real channels
    crop: Tuple[int, int, int] = (0, 0, 0)
    factor: Tuple[int, int, int] = (1, 1, 1)


@dataclass
class ConvLayer:
    name: str                        # state_dict prefix, e.g. backbone.l_conv.0.conv_pass.0
    sources: List[Source]
    cout: int
    kernel: Tuple[int, int, int]     # (kd, kh, kw), kd = 1 for 2-D
    in_shape: Tuple[int, int, int]   # logical input extent (D, H, W)
    out: str                         # output buffer name
    relu: bool = True
    param_index: int = -1            # index into the flat (weight, bias) list

    @property
    def cin(self):
        return sum(s.channels for s in self.sources)

    @property
    def cin_pad(self):
        return sum(pad4(s.channels) for s in self.sources)

    @property
    def taps(self):
        return self.kernel[0] * self.kernel[1] * self.kernel[2]

    @property
    def out_shape(self):
        return tuple(i - k + 1 for i, k in zip(self.in_shape, self.kernel))


@dataclass
class PoolOp:
    src: str
    out: str
    channels: int
    in_shape: Tuple[int, int, int]
    factor: Tuple[int, int, int]


@dataclass
class Topology:
    """Static description of the network for one input crop shape."""

    nd: int
    in_channels: int
    out_channels: int
    in_shape: Tuple[int, int, int]
    convs: List[ConvLayer] = field(default_factory=list)
    pools: List[PoolOp] = field(default_factory=list)
    fwd_order: list = field(default_factory=list)      # ConvLayer | PoolOp in execution order
    shapes: dict = field(default_factory=dict)         # buffer -> ((D,H,W), channels)
    levels: int = 0
    # per r-level: (conv0 layer, skip tensor, up tensor)
    r_info: list = field(default_factory=list)
    out_shape: Tuple[int, int, int] = (1, 1, 1)


def build_topology(in_channels, out_channels, num_fmaps, fmap_inc_factor, features_in_last_layer,
                   downsampling_factors, num_spatial_dims, spatial):
    """Derives every layer's geometry for an input of spatial extent `spatial`."""
    nd = num_spatial_dims
    assert nd in (2, 3), "num_spatial_dims must be 2 or 3"
    assert len(spatial) == nd
    L = len(downsampling_factors)
    factors = []
    for f in downsampling_factors:
        f = tuple(int(a) for a in f)
        assert len(f) == nd, "downsampling factor rank must equal num_spatial_dims"
        factors.append((1,) + f if nd == 2 else f)
    shape = ((1,) + tuple(int(s) for s in spatial)) if nd == 2 else tuple(int(s) for s in spatial)
    k3 = (1, 3, 3) if nd == 2 else (3, 3, 3)
    k1 = (1, 1, 1)
    pass_kernels = [k3, k1, k1, k3]
    conv_crop = tuple(sum(k[d] - 1 for k in pass_kernels) for d in range(3))

    topo = Topology(nd=nd, in_channels=in_channels, out_channels=out_channels, in_shape=shape, levels=L)
    topo.shapes["raw"] = (shape, in_channels)

    def add_pass(prefix, first_sources, first_in_shape, cout, out_prefix):
        srcs, ishape = first_sources, first_in_shape
        layers = []
        for j, k in enumerate(pass_kernels):
            name = f"{prefix}.conv_pass.{2 * j}"
            out = f"{out_prefix}.{j}"
            layer = ConvLayer(name=name, sources=srcs, cout=cout, kernel=k, in_shape=ishape, out=out)
            for d in range(3):
                if layer.out_shape[d] <= 0:
                    raise ValueError(
                        f"input extent {spatial} is too small for the U-Net (layer {name} would be empty)")
            topo.convs.append(layer)
            topo.fwd_order.append(layer)
            topo.shapes[out] = (layer.out_shape, cout)
            layers.append(layer)
            srcs, ishape = [Source(out, cout)], layer.out_shape
        return layers

    # ---- left (contracting) path
    left_out = []
    cur, cur_c, cur_shape = "raw", in_channels, shape
    for i in range(L + 1):
        cout = num_fmaps * fmap_inc_factor ** i
        layers = add_pass(f"backbone.l_conv.{i}", [Source(cur, cur_c)], cur_shape, cout, f"l{i}")
        y, yshape = layers[-1].out, layers[-1].out_shape
        left_out.append((y, cout, yshape))
        if i < L:
            f = factors[i]
            for d in range(3):
                if yshape[d] % f[d] != 0:
                    raise RuntimeError(
                        f"Can not downsample shape {yshape[3 - nd:]} with factor {f[3 - nd:]}, "
                        f"mismatch in spatial dimension {d - (3 - nd)}")
            pshape = tuple(s // ff for s, ff in zip(yshape, f))
            pool = PoolOp(src=y, out=f"p{i}", channels=cout, in_shape=yshape, factor=f)
            topo.pools.append(pool)
            topo.fwd_order.append(pool)
            topo.shapes[pool.out] = (pshape, cout)
            cur, cur_c, cur_shape = pool.out, cout, pshape

    # ---- right (expanding) path, bottom-up
    crop_factors = []
    prod = None
    for f in factors[::-1]:
        prod = tuple(f) if prod is None else tuple(a * b for a, b in zip(f, prod))
        crop_factors.append(prod)
    crop_factors = crop_factors[::-1]

    below, below_c, below_shape = left_out[L]
    topo.r_info = [None] * L
    for i in range(L - 1, -1, -1):
        f = factors[i]
        up_shape = tuple(s * ff for s, ff in zip(below_shape, f))
        # crop_to_factor: keep (size - conv_crop) a multiple of the cumulative factor
        cf = crop_factors[i]
        target = tuple(int(math.floor((s - c) / ff)) * ff + c for s, c, ff in zip(up_shape, conv_crop, cf))
        for d in range(3):
            if target[d] <= conv_crop[d] and up_shape[d] != target[d]:
                raise RuntimeError(f"Feature map with shape {up_shape} is too small for cropping to factor")
        up_crop = tuple((s - t) // 2 for s, t in zip(up_shape, target))
        skip, skip_c, skip_shape = left_out[i]
        for d in range(3):
            if skip_shape[d] < target[d]:
                raise RuntimeError("skip connection smaller than the upsampled feature map")
        skip_crop = tuple((s - t) // 2 for s, t in zip(skip_shape, target))
        cout = features_in_last_layer if i == 0 else num_fmaps * fmap_inc_factor ** i
        srcs = [Source(skip, skip_c, crop=skip_crop),
                Source(below, below_c, crop=up_crop, factor=f)]
        layers = add_pass(f"backbone.r_conv.0.{i}", srcs, target, cout, f"r{i}")
        topo.r_info[i] = dict(conv0=layers[0], skip=skip, up=below, level=i)
        below, below_c, below_shape = layers[-1].out, cout, layers[-1].out_shape

    # ---- head: 1x1 conv + ReLU + 1x1 conv (unet.py:52-63)
    top, top_c, top_shape = below, below_c, below_shape
    if L > 0 and top_c != features_in_last_layer:
        raise AssertionError("internal: top level width mismatch")
    h0 = ConvLayer(name="head.0", sources=[Source(top, top_c)], cout=features_in_last_layer,
                   kernel=k1, in_shape=top_shape, out="h0", relu=True)
    h1 = ConvLayer(name="head.2", sources=[Source("h0", features_in_last_layer)], cout=out_channels,
                   kernel=k1, in_shape=top_shape, out="h1", relu=False)
    if L == 0 and top_c != features_in_last_layer:
        # funlib keeps num_fmaps at level 0 when there is no upsampling path; the
        # reference head then expects features_in_last_layer inputs (a config error).
        raise ValueError("with no downsampling, num_fmaps must equal features_in_last_layer")
    for layer in (h0, h1):
        topo.convs.append(layer)
        topo.fwd_order.append(layer)
        topo.shapes[layer.out] = (layer.out_shape, layer.cout)
    topo.out_shape = top_shape
    for idx, layer in enumerate(topo.convs):
        layer.param_index = idx
    return topo


def tensor_consumers(topo):
    """How many operations READ each stored tensor: every source of every convolution (skip connections and
    upsampled tensors are sources of a level's first convolution) and every pooling."""
    n = {}
    for layer in topo.convs:
        for src in layer.sources:
            n[src.tensor] = n.get(src.tensor, 0) + 1
    for pool in topo.pools:
        n[pool.src] = n.get(pool.src, 0) + 1
    return n


def find_chain_pairs(topo, algo_fwd, batch):
    """The (a, b) pairs of consecutive 64-channel 1x1 layers that may run as one launch each way.

    The fused backward pass OVERWRITES the gradient of the pair's input and never writes the gradient of the
    middle tensor (and neither tensor gets ReLU gate bits), so a pair qualifies only if the middle tensor is read
    by `b` alone and the pair's input by `a` alone: a tensor that is also a skip connection, pooled, or read by a
    second convolution keeps the layer-by-layer path, where gradients add up."""
    produced_by_conv = {layer.out: layer for layer in topo.convs}
    readers = tensor_consumers(topo)
    one = (1, 1, 1)
    plain = lambda s: tuple(s.crop) == (0, 0, 0) and tuple(s.factor) == (1, 1, 1)       # noqa: E731
    pairs = []
    i = 0
    while i + 1 < len(topo.convs):
        a, b = topo.convs[i], topo.convs[i + 1]
        ok = (tuple(a.kernel) == one and tuple(b.kernel) == one and len(a.sources) == 1 and len(b.sources) == 1
              and plain(a.sources[0]) and plain(b.sources[0]) and b.sources[0].tensor == a.out
              and a.sources[0].tensor in produced_by_conv and a.sources[0].channels == 64 and a.cout == 64
              and a.relu and (b.cout == 64 or b.cout <= 8)
              and readers.get(a.out, 0) == 1 and readers.get(a.sources[0].tensor, 0) == 1
              and not algo_fwd[a.name] and not algo_fwd[b.name]
              and batch * a.in_shape[0] * a.in_shape[1] * a.in_shape[2] < (1 << 31) - 256)
        if ok:
            pairs.append((a, b))
            i += 2
        else:
            i += 1
    return pairs


class UNetPlan:
    """Executes a Topology for a fixed batch size on one HIP device."""

    def __init__(self, topo: Topology, batch: int, device: torch.device, keep_activations: bool):
        self.topo = topo
        self.B = int(batch)
        self.device = device
        self.keep = keep_activations
        self.precision = precision_code()
        # (the one-launch Winograd kernels multiply in float32: a layer whose products can run in the split precision takes
        #  the three-launch form — _fused_ok —, the narrow layers keep the one-launch form)
        self.fused = fused_wanted(keep_activations)
        # opt-in: run-to-run reproducible training (CLX_DETERMINISTIC=1; the reference's CPU autograd is
        # deterministic, cellulus/train.py:177-179).  Weight-gradient slices add in a fixed order, bias
        # gradients come from ordered column sums, the first layer takes the generic kernel and the fused
        # 1x1 pairs are off (their block sums meet in float atomics); train._fused_step switches the loss.
        self.deterministic = os.environ.get("CLX_DETERMINISTIC", "0") == "1"
        self._wplanes = {}          # data_ptr of a packed-weight tensor -> (tensor, its P3 planes, rows, K)
        self.aplanes = None         # scratch for the planes of a 1x1 layer's input (inference) ...
        self.xplanes = {}           # ... or one buffer per layer (training: the weight gradient reuses them)
        self._xplanes_fresh = set()
        self.dyplanes = None        # planes of the dY a 1x1 layer's weight gradient has split, reused by its data gradient
        self.dyplanes2 = None
        self._pointwise_reader = {}
        self.buf = {}
        self._alloc()

    def _fused_ok(self, cin_pad, cout):
        """the one-launch (float32) Winograd form for a 2-D layer cin_pad -> cout?  Where it pays (fused_pays) — in the
        float32-MFMA precision only: its 36 x 32 x 64 accumulators are ONE chain over the contraction (no room for the
        second set of the two-level summation), and with it on the 64-channel layers the default precision's inference
        plan sat at 8.26e-5 from float64 at a trained network's output scale where the three-launch forms give 7.20e-5
        and float32 MFMA 8.16e-5 (profiles/r06_parity_trained_scale_2d*.txt) — the promotion rule of DESIGN.md 4 asks
        for <=, so the default precision runs every Winograd layer in three launches (-4 % on an inference tile)."""
        return bool(self.fused and not self.precision and fused_pays(cin_pad, cout))

    # ------------------------------------------------------------------ memory
    def _alloc(self):
        t = self.topo
        for name, (shape, c) in t.shapes.items():
            n = self.B * shape[0] * shape[1] * shape[2]
            self.buf[name] = _clx.zeros((n, pad4(c)), torch.float32, self.device)
        # per-layer algorithm (direct implicit GEMM / Winograd) for forward, dgrad and wgrad
        self.algo = {}
        self.workspace = None
        ws_bytes = 0
        for layer in t.convs:
            a = dict(fwd=0, dgrad=0, wgrad=0)
            if (winograd_enabled() and min(layer.cin_pad, layer.cout) >= wino_min_channels(layer.kernel)
                    and (layer.kernel[0] == 1 or winograd_code() == 2)):
                lib = _clx.load()
                code = winograd_code()
                d = self._desc(layer)
                d.algo = code
                d.N = layer.cout
                nf = int(lib.clx_conv_workspace_bytes(ctypes.byref(d), 0))
                if nf:
                    a["fwd"], ws_bytes = code, max(ws_bytes, nf)
                    if (code == 2 and self._fused_ok(layer.cin_pad, layer.cout) and layer.cout == pad4(layer.cout)
                            and int(lib.clx_conv_fused_applicable(ctypes.byref(d)))):
                        a["fwd"] = 3
                        ws_bytes = max(ws_bytes, int(lib.clx_conv_fused_workspace_bytes(ctypes.byref(d))))
                d.N = pad4(layer.cout)
                nw = int(lib.clx_conv_workspace_bytes(ctypes.byref(d), 1))
                if nw and self.keep:
                    a["wgrad"], ws_bytes = code, max(ws_bytes, nw)
                if self.keep and layer.param_index > 0:
                    dd = self._dgrad_desc(layer, None)
                    dd.algo = code
                    nd_ = int(lib.clx_conv_workspace_bytes(ctypes.byref(dd), 0))
                    if nd_:
                        a["dgrad"], ws_bytes = code, max(ws_bytes, nd_)
            self.algo[layer.name] = a
        # sub-pixel form of the convolutions that read a nearest-upsampled tensor
        self.subpixel = {}
        for info in t.r_info:
            sp = self._subpixel_geometry(info["conv0"])
            if sp is not None:
                self.subpixel[info["conv0"].name] = sp
                n = self.B * sp["zshape"][0] * sp["zshape"][1] * sp["zshape"][2]
                self.buf[sp["zname"]] = _clx.zeros((n, sp["P"] * sp["N"]), torch.float32, self.device)
                # the 2x2 convolution over the low-res tensor as Winograd F(4x4, 2x2)
                sp["wino"] = 0
                sp["fused_z"] = sp["fused_skip"] = False    # forward halves in the one-launch form (wino_fused.hip)
                if (winograd_enabled() and winograd_code() == 2 and sp["zk"] in ((1, 2, 2), (2, 2, 2))
                        and min(sp["C1p"], sp["P"] * sp["N"]) >= wino_min_channels(sp["zk"])):
                    lib = _clx.load()
                    dz, _ds = self._sp_descs(info["conv0"], sp)
                    dz.algo = 2
                    need = [int(lib.clx_conv_workspace_bytes(ctypes.byref(dz), 0))]
                    if self.keep:
                        need.append(int(lib.clx_conv_workspace_bytes(ctypes.byref(dz), 1)))
                        dl = self._sp_low_dgrad_desc(info["conv0"], sp, None)
                        dl.algo = 2
                        need.append(int(lib.clx_conv_workspace_bytes(ctypes.byref(dl), 0)))
                    if all(need):
                        sp["wino"] = 2
                        ws_bytes = max([ws_bytes] + need)
                        sp["fused_z"] = bool(self._fused_ok(sp["C1p"], sp["P"] * sp["N"])
                                             and int(lib.clx_conv_fused_applicable(ctypes.byref(dz))))
                        if sp["fused_z"]:
                            ws_bytes = max(ws_bytes, int(lib.clx_conv_fused_workspace_bytes(ctypes.byref(dz))))
                # ... and the 3x3 convolution over the skip tensor as F(4x4, 3x3), forward and weight
                # gradient only: its data gradient has K = N (64 at the benchmark config), too short
                # a contraction for the batched GEMMs to pay
                sp["wino_skip"] = 0
                conv0 = info["conv0"]
                if (winograd_enabled() and winograd_code() == 2 and tuple(conv0.kernel) in ((1, 3, 3), (3, 3, 3))
                        and sp["C0p"] >= wino_min_channels(conv0.kernel)
                        and sp["N"] >= wino_min_channels(conv0.kernel) // 2):
                    lib = _clx.load()
                    _dz, ds = self._sp_descs(conv0, sp)
                    ds.algo = 2
                    ds.N = conv0.cout
                    need = [int(lib.clx_conv_workspace_bytes(ctypes.byref(ds), 0))]
                    if self.keep:
                        ds.N = sp["N"]
                        need.append(int(lib.clx_conv_workspace_bytes(ctypes.byref(ds), 1)))
                    if all(need):
                        sp["wino_skip"] = 2
                        ws_bytes = max([ws_bytes] + need)
                        ds.N = conv0.cout
                        sp["fused_skip"] = bool(self._fused_ok(sp["C0p"], conv0.cout) and conv0.cout == sp["N"]
                                                and int(lib.clx_conv_fused_applicable(ctypes.byref(ds))))
                # its data gradient contracts over z taps x output channels: long enough only in 3-D
                sp["wino_skip_dgrad"] = 0
                if sp["wino_skip"] and self.keep and sp["N"] * conv0.kernel[0] >= WINO_MIN_CHANNELS:
                    dd = self._dgrad_desc(conv0, None)
                    dd.N = sp["C0p"]
                    dd.algo = 2
                    need = int(_clx.load().clx_conv_workspace_bytes(ctypes.byref(dd), 0))
                    if need:
                        sp["wino_skip_dgrad"] = 2
                        ws_bytes = max(ws_bytes, need)
                ztaps = 25 * sp["zk"][0] if sp["wino"] else sp["ztaps"]
                staps = 36 * info["conv0"].kernel[0] if sp["wino_skip"] else info["conv0"].taps
                sp["w_skip"] = torch.empty(conv0.cout * sp["C0"] * conv0.taps, dtype=torch.float32, device=self.device)
                sp["weff"] = torch.empty(sp["P"] * sp["N"] * sp["C1"] * sp["ztaps"], dtype=torch.float32,
                                         device=self.device)
                sp["wp_skip_fwd"] = torch.empty(sp["N"] * staps * sp["C0p"],
                                                dtype=torch.float32, device=self.device)
                sp["wp_z_fwd"] = torch.empty(sp["P"] * sp["N"] * ztaps * sp["C1p"],
                                             dtype=torch.float32, device=self.device)
        if ws_bytes:
            self.workspace = torch.empty(ws_bytes // 4 + 4, dtype=torch.float32, device=self.device)
        # packed weights
        self.wpack_fwd = {}
        self.wpack_dgrad = {}
        for layer in t.convs:
            code = self.algo[layer.name]["fwd"]
            taps = wino_taps(code, layer.kernel) if code else layer.taps
            self.wpack_fwd[layer.name] = torch.empty(
                pad4(layer.cout) * taps * layer.cin_pad, dtype=torch.float32, device=self.device)
            if code and layer.kernel[0] == 1:
                self._register_wplanes(self.wpack_fwd[layer.name], pad4(layer.cout), WINO_TAPS[code], layer.cin_pad)
            elif not code and self._pointwise_sp(layer)[0]:
                self._register_wplanes(self.wpack_fwd[layer.name], pad4(layer.cout), 1, layer.cin_pad)
        for sp in self.subpixel.values():
            if sp["wino"] and sp["zk"][0] == 1:
                self._register_wplanes(sp["wp_z_fwd"], sp["P"] * sp["N"], 25, sp["C1p"])
        if self.precision:
            # scratch for the planes of a 1x1 layer's input (a training plan keeps one buffer per layer instead)
            rows_k = [(self.B * layer.in_shape[0] * layer.in_shape[1] * layer.in_shape[2], layer.cin_pad)
                      for layer in t.convs if self._pointwise_sp(layer)[0]]
            if rows_k:
                self.aplanes = torch.empty(max(int(_clx.load().clx_planes_bytes(r, k)) for r, k in rows_k),
                                           dtype=torch.uint8, device=self.device)
        self._packed_version = None
        self._bwd_ready = False
        self.vcache = {}
        self._vcache_fresh = set()
        self._find_chains()
        # 2 x 2 max-pooling written by the Winograd output transform of the layer that produces the pooled tensor (2-D:
        # a pooling window lies inside one output tile; clx_conv_desc.pool_out).  CLX_FUSED_POOL=0: the separate pass
        self.fused_pool = {}
        if os.environ.get("CLX_FUSED_POOL", "1") != "0":
            by_out = {layer.out: layer for layer in t.convs}
            for pool in t.pools:
                layer = by_out.get(pool.src)
                if (layer is not None and self.algo[layer.name]["fwd"] and layer.kernel[0] == 1 and layer.in_shape[0] == 1
                        and tuple(pool.factor) == (1, 2, 2) and layer.cout % 4 == 0 and layer.name not in self.subpixel
                        and layer.out_shape[1] % 2 == 0 and layer.out_shape[2] % 2 == 0):
                    self.fused_pool[layer.name] = pool

    def share_from(self, other):
        """Use `other`'s packed weights and gradient accumulators (same topology, batch size and switches): this
        plan then never packs, and its weight-gradient kernels add into the accumulators `other` unpacks."""
        assert other._bwd_ready and self._bwd_ready and other.B == self.B and other.algo == self.algo
        assert other.dw_off == self.dw_off and other.dwpack.numel() == self.dwpack.numel()
        self.wpack_fwd, self.wpack_dgrad, self.dwpack = other.wpack_fwd, other.wpack_dgrad, other.dwpack
        self._wplanes = other._wplanes
        for name, sp in self.subpixel.items():
            for key in ("w_skip", "weff", "wp_skip_fwd", "wp_z_fwd", "wp_skip_dgrad", "wp_z_dgrad", "dw_skip", "dw_z"):
                sp[key] = other.subpixel[name][key]

    def share_forward_from(self, other):
        """Forward-only version of share_from: this plan reads `other`'s packed weights and never packs."""
        assert other.B == self.B and other.algo == self.algo and other.keep == self.keep
        self.wpack_fwd = other.wpack_fwd
        self._wplanes = other._wplanes
        for name, sp in self.subpixel.items():
            for key in ("w_skip", "weff", "wp_skip_fwd", "wp_z_fwd"):
                sp[key] = other.subpixel[name][key]
        self._packed_version = other._packed_version

    def _find_chains(self):
        """Pairs of consecutive 64-channel 1x1 layers (conv_pass.2 -> conv_pass.4 of a level, head.0 ->
        head.2) that run as ONE launch each way (csrc/chain64.hip: the intermediate tensor is written once
        and never read back, its gradient never exists in HBM).  CLX_CHAIN64=0 keeps the layer-by-layer path."""
        self.chains, self.chain_second = {}, {}
        if os.environ.get("CLX_CHAIN64", "1") == "0" or self.deterministic:
            return
        for a, b in find_chain_pairs(self.topo, {n: v["fwd"] for n, v in self.algo.items()}, self.B):
            self.chains[a.name] = (a, b)
            self.chain_second[b.name] = (a, b)

    def _alloc_backward(self):
        t = self.topo
        self.gbuf = {}
        for name, (shape, c) in t.shapes.items():
            if name == "raw":
                continue
            n = self.B * shape[0] * shape[1] * shape[2]
            self.gbuf[name] = _clx.zeros((n, pad4(c)), torch.float32, self.device)
        for info in t.r_info:
            layer = info["conv0"]
            n = self.B * layer.in_shape[0] * layer.in_shape[1] * layer.in_shape[2]
            sp = self.subpixel.get(layer.name)
            if sp is None:
                self.gbuf["cat%d" % info["level"]] = _clx.zeros((n, layer.cin_pad), torch.float32, self.device)
            else:
                self.gbuf["dskip%d" % info["level"]] = _clx.zeros((n, sp["C0p"]), torch.float32, self.device)
                self.gbuf[sp["zname"]] = _clx.zeros(tuple(self.buf[sp["zname"]].shape), torch.float32, self.device)
                sp["wp_skip_dgrad"] = torch.empty(sp["C0p"] * (36 * layer.kernel[0] if sp["wino_skip_dgrad"] else layer.taps)
                                                  * sp["N"], dtype=torch.float32,
                                                  device=self.device)
                ztaps = 25 * sp["zk"][0] if sp["wino"] else sp["ztaps"]
                sp["wp_z_dgrad"] = torch.empty(sp["C1p"] * ztaps * sp["P"] * sp["N"],
                                               dtype=torch.float32, device=self.device)
                sp["_dw_skip_n"] = (36 * layer.kernel[0] if sp["wino_skip"] else layer.taps) * sp["N"] * sp["C0p"]
                sp["g_skip"] = torch.empty(layer.cout * sp["C0"] * layer.taps, dtype=torch.float32, device=self.device)
                sp["g_z"] = torch.empty(sp["P"] * sp["N"] * sp["C1"] * sp["ztaps"], dtype=torch.float32,
                                        device=self.device)
                if sp["wino_skip"] and not sp["fused_skip"] and os.environ.get("CLX_WINOGRAD_VCACHE", "1") != "0":
                    tiles = self.B * layer.in_shape[0] * -(-layer.out_shape[1] // 4) * -(-layer.out_shape[2] // 4)
                    sp["vcache_skip"] = torch.empty(self._vfloats(36, tiles, sp["C0p"]), dtype=torch.float32, device=self.device)
                sp["_dw_z_n"] = ztaps * sp["P"] * sp["N"] * sp["C1p"]
                if sp["wino"] and not sp["fused_z"] and os.environ.get("CLX_WINOGRAD_VCACHE", "1") != "0":
                    zs = sp["zshape"]
                    tiles = self.B * (zs[0] + sp["zk"][0] - 1) * -(-zs[1] // 4) * -(-zs[2] // 4)
                    sp["vcache"] = torch.empty(self._vfloats(25, tiles, sp["C1p"]), dtype=torch.float32, device=self.device)
                if sp["wino"] and sp["zk"][0] == 1:
                    self._register_wplanes(sp["wp_z_dgrad"], sp["C1p"], 25, sp["P"] * sp["N"])
        # ReLU gates as bits: written by the epilogue that produces a layer's output, read by the data
        # gradient that passes through that ReLU — 1/32 of the float tensor it would otherwise read (the
        # 64-channel 1x1 layers of the 3-D network are HBM-bound).  Whole words per pixel (channels %
        # 32 == 0) and a producer that knows the bits (not the first-layer kernels); CLX_GATE_BITS=0 = off
        self.gate = {}
        # (the fused 1x1 pairs gate by the float tensors they read anyway: no bits for their input and middle)
        chain_gated = {a.out for a, _b in self.chains.values()} | {a.sources[0].tensor for a, _b in self.chains.values()}
        if os.environ.get("CLX_GATE_BITS", "1") != "0":
            for layer in t.convs:
                if layer.out in chain_gated:
                    continue
                if layer.relu and layer.param_index > 0 and pad4(layer.cout) % 32 == 0:
                    n = self.B * layer.out_shape[0] * layer.out_shape[1] * layer.out_shape[2]
                    self.gate[layer.out] = _clx.zeros((n, pad4(layer.cout) // 32), torch.int32, self.device)
        # F(4x4, 3x3[x3]) layers whose weight AND data gradient are Winograd: the data gradient in its ADJOINT form,
        # dX = sum over tiles of B [U^T (A dY A^T)] B^T — its operand A dY A^T is what the weight gradient has just left
        # in the workspace, so dY is transformed once and the (K-1)-padded input transform of dY is never written
        # (clx_conv_desc.adjoint; CLX_WINO_ADJOINT=0 = the two-transform form)
        self.adjoint = set()
        if os.environ.get("CLX_WINO_ADJOINT", "1") != "0":
            for layer in t.convs:
                a = self.algo[layer.name]
                if (a["wgrad"] == 2 and a["dgrad"] == 2 and tuple(layer.kernel) in ((1, 3, 3), (3, 3, 3)) and layer.param_index > 0
                        and layer.name not in self.subpixel and len(layer.sources) == 1):
                    self.adjoint.add(layer.name)
        # a Winograd layer's weight gradient and data gradient both transform dY: one pass produces both
        # (clx_conv_desc.dy_vcache); the buffer is shared by all layers (written and consumed back to back)
        self.dycache = None
        if os.environ.get("CLX_DY_DUAL", "1") != "0":
            need = 0

            def dual_floats(code, out_shape, k, chans):
                a2, m = (WINO_TAPS[code], WINO_TILE[code]) if k == 3 else (25, 4)
                return self._vfloats(a2, self.B * out_shape[0] * (-(-(out_shape[1] + k - 1) // m)) * (-(-(out_shape[2] + k - 1) // m)),
                                     chans)

            for layer in t.convs:
                a = self.algo[layer.name]
                if a["wgrad"] and a["wgrad"] == a["dgrad"] and layer.name not in self.subpixel and layer.name not in self.adjoint:
                    need = max(need, dual_floats(a["wgrad"], layer.out_shape, 3, pad4(layer.cout)))
            for name, sp in self.subpixel.items():
                if sp["wino"]:
                    need = max(need, dual_floats(2, sp["zshape"], 2, sp["P"] * sp["N"]))
            if need:
                self.dycache = torch.empty(need + 4, dtype=torch.float32, device=self.device)
        # forward and weight gradient of a Winograd layer transform the same input: keep V
        self.vcache = {}
        for layer in t.convs:
            a = self.algo[layer.name]
            if a["fwd"] and a["fwd"] == a["wgrad"] and os.environ.get("CLX_WINOGRAD_VCACHE", "1") != "0":
                m = WINO_TILE[a["fwd"]]
                tiles = self.B * layer.in_shape[0] * -(-layer.out_shape[1] // m) * -(-layer.out_shape[2] // m)
                self.vcache[layer.name] = torch.empty(self._vfloats(WINO_TAPS[a["fwd"]], tiles, layer.cin_pad),
                                                      dtype=torch.float32, device=self.device)
        total = 0
        self.dw_off = {}
        for layer in t.convs:
            self.dw_off[layer.name] = total
            code = self.algo[layer.name]["wgrad"]
            total += (wino_taps(code, layer.kernel) if code else layer.taps) * pad4(layer.cout) * layer.cin_pad
            if layer.param_index > 0:  # first layer needs no data gradient
                code = self.algo[layer.name]["dgrad"]
                taps = wino_taps(code, layer.kernel) if code else layer.taps
                self.wpack_dgrad[layer.name] = torch.empty(
                    layer.cin_pad * taps * pad4(layer.cout), dtype=torch.float32, device=self.device)
                if code and layer.kernel[0] == 1:
                    self._register_wplanes(self.wpack_dgrad[layer.name], layer.cin_pad, WINO_TAPS[code], pad4(layer.cout))
                elif not code and self._pointwise_sp(layer)[1]:
                    self._register_wplanes(self.wpack_dgrad[layer.name], layer.cin_pad, 1, pad4(layer.cout))
        if self.precision:
            # training: the planes of a 1x1 layer's input stay for its weight gradient; one scratch for the planes of dY
            need_dy = 0
            for layer in t.convs:
                fwd_sp, dgrad_sp, wgrad_sp = self._pointwise_sp(layer)
                if layer.name in self.chains or layer.name in self.chain_second:
                    continue
                rows = self.B * layer.in_shape[0] * layer.in_shape[1] * layer.in_shape[2]
                if fwd_sp or wgrad_sp:
                    self.xplanes[layer.name] = self._planes_scratch(rows, layer.cin_pad)
                if (dgrad_sp and layer.param_index > 0) or wgrad_sp:
                    need_dy = max(need_dy, int(_clx.load().clx_planes_bytes(rows, pad4(layer.cout))))
            if need_dy:
                # two of them: a data gradient reads the planes of its dY out of one while its epilogue writes the planes
                # of the next layer's dY into the other
                self.dyplanes = torch.empty(need_dy, dtype=torch.uint8, device=self.device)
                self.dyplanes2 = torch.empty(need_dy, dtype=torch.uint8, device=self.device)
            # tensor -> the 1x1 layer that reads it as its one plain source and keeps planes of it
            for layer in t.convs:
                if layer.name in self.xplanes and pad4(t.shapes[layer.sources[0].tensor][1]) == layer.cin_pad:
                    self._pointwise_reader[layer.sources[0].tensor] = layer
        # the sub-pixel layers' weight-gradient accumulators live behind the others: one fill zeroes all
        sp_off = {}
        for name, sp in self.subpixel.items():
            sp_off[name] = (total, total + sp["_dw_skip_n"])
            total += sp["_dw_skip_n"] + sp["_dw_z_n"]
        self.dwpack = _clx.zeros(total, torch.float32, self.device)
        if self.deterministic:
            lib = _clx.load()
            width = max(pad4(layer.cout) for layer in t.convs)
            self._det_turns = _clx.zeros(1 << 20, torch.int32, self.device)     # 4 MB of turn counters
            self._det_colsum = torch.empty(int(lib.clx_colsum_scratch_bytes(width)) // 4, dtype=torch.float32,
                                           device=self.device)
        for name, sp in self.subpixel.items():
            a, b = sp_off[name]
            sp["dw_skip"] = self.dwpack[a:b]
            sp["dw_z"] = self.dwpack[b:b + sp["_dw_z_n"]]
        self._bwd_ready = True

    # --------------------------------------------------------------- sub-pixel
    def _subpixel_geometry(self, layer: ConvLayer):
        """A convolution over cat(skip, nearest-upsample(low)) equals, exactly,
             conv(skip) + depth_to_space( conv_{2^d taps}(low, phase-summed weights) )
        because the k=3 taps of an output pixel with parity a fall on only two low-res rows.
        Returns the geometry of that rewrite or None when it does not apply (odd crop, odd
        output extent, cropped low-res grid, CLX_SUBPIXEL=0)."""
        if os.environ.get("CLX_SUBPIXEL", "1") == "0" or len(layer.sources) != 2:
            return None
        skip_s, up_s = layer.sources
        f, k, o = up_s.factor, layer.kernel, up_s.crop
        if max(f) != 2 or min(f) < 1 or skip_s.factor != (1, 1, 1):
            return None
        low_shape, low_c = self.topo.shapes[up_s.tensor]
        zshape, zk, zcrop = [], [], []
        for d in range(3):
            if f[d] == 2:
                if k[d] != 3 or o[d] % 2 != 0 or layer.out_shape[d] % 2 != 0:
                    return None
                zshape.append(layer.out_shape[d] // 2)
                zk.append(2)
                zcrop.append(o[d] // 2)
            else:
                zshape.append(layer.out_shape[d])
                zk.append(k[d])
                zcrop.append(o[d])
            # the Z convolution must read the WHOLE low-res grid (its dgrad writes all of it)
            if zcrop[d] != 0 or zshape[d] + zk[d] - 1 != low_shape[d]:
                return None
        level = [i["level"] for i in self.topo.r_info if i["conv0"] is layer][0]
        return dict(fac=f, P=f[0] * f[1] * f[2], N=pad4(layer.cout), C0=skip_s.channels, C0p=pad4(skip_s.channels),
                    C1=up_s.channels, C1p=pad4(up_s.channels), zshape=tuple(zshape), zk=tuple(zk),
                    ztaps=zk[0] * zk[1] * zk[2], zname="Z%d" % level, level=level)

    @staticmethod
    def _phase_sum(w, axis, a):
        """3 taps -> 2 taps along `axis` for output parity `a`: the taps that land on the same
        low-res row are summed (a = 0: {0,1},{2};  a = 1: {0},{1,2}).  Elementwise torch ops."""
        t0, t1, t2 = w.select(axis, 0), w.select(axis, 1), w.select(axis, 2)
        pair = (t0 + t1, t2) if a == 0 else (t0, t1 + t2)
        return torch.stack(pair, dim=axis)

    @staticmethod
    def _phase_spread(g, axis, a):
        """adjoint of _phase_sum: 2 taps -> 3 taps."""
        g0, g1 = g.select(axis, 0), g.select(axis, 1)
        trip = (g0, g0, g1) if a == 0 else (g0, g1, g1)
        return torch.stack(trip, dim=axis)

    def _phase_weights(self, layer, sp, w_up):
        """w_up (cout, C1, kd, kh, kw) -> phase-summed (P*N, C1, zkd, zkh, zkw); rows of padded
        output channels are zero.  Tiny tensors: plain elementwise torch ops."""
        f, N, cout = sp["fac"], sp["N"], layer.cout
        out = w_up.new_zeros((sp["P"], N, sp["C1"]) + sp["zk"])
        for a in range(f[0]):
            for b in range(f[1]):
                for c in range(f[2]):
                    v = w_up
                    for axis, (ff, par) in enumerate(zip(f, (a, b, c))):
                        if ff == 2:
                            v = self._phase_sum(v, 2 + axis, par)
                    out[(a * f[1] + b) * f[2] + c, :cout] = v
        return out.reshape((sp["P"] * N, sp["C1"]) + sp["zk"])

    def _fold_phase_grads(self, layer, sp, dweff):
        """adjoint of _phase_weights: (P*N, C1, zk...) -> (cout, C1, kd, kh, kw)."""
        f, N, cout = sp["fac"], sp["N"], layer.cout
        g = dweff.reshape((sp["P"], N, sp["C1"]) + sp["zk"])
        out = dweff.new_zeros((cout, sp["C1"]) + tuple(layer.kernel))
        for a in range(f[0]):
            for b in range(f[1]):
                for c in range(f[2]):
                    v = g[(a * f[1] + b) * f[2] + c, :cout]
                    for axis, (ff, par) in enumerate(zip(f, (a, b, c))):
                        if ff == 2:
                            v = self._phase_spread(v, 2 + axis, par)
                    out += v
        return out

    def _sp_descs(self, layer, sp):
        """(Z-convolution descriptor over the low-res tensor, skip-convolution descriptor)."""
        t = self.topo
        skip_s, up_s = layer.sources
        dz = ClxConvDesc()
        dz.nsrc = 1
        low_shape, low_c = t.shapes[up_s.tensor]
        src = ClxSrc()
        src.ptr = self.buf[up_s.tensor].data_ptr()
        src.C = sp["C1p"]
        src.ld = pad4(low_c)
        src.D, src.H, src.W = low_shape
        src.oz = src.oy = src.ox = 0
        src.fz = src.fy = src.fx = 1
        dz.src[0] = src
        dz.B = self.B
        dz.ID, dz.IH, dz.IW = low_shape
        dz.KD, dz.KH, dz.KW = sp["zk"]
        dz.PD = dz.PH = dz.PW = 0
        dz.N = sp["P"] * sp["N"]
        ds = ClxConvDesc()
        ds.nsrc = 1
        sshape, sc = t.shapes[skip_s.tensor]
        src2 = ClxSrc()
        src2.ptr = self.buf[skip_s.tensor].data_ptr()
        src2.C = sp["C0p"]
        src2.ld = pad4(sc)
        src2.D, src2.H, src2.W = sshape
        src2.oz, src2.oy, src2.ox = skip_s.crop
        src2.fz = src2.fy = src2.fx = 1
        ds.src[0] = src2
        ds.B = self.B
        ds.ID, ds.IH, ds.IW = layer.in_shape
        ds.KD, ds.KH, ds.KW = layer.kernel
        ds.PD = ds.PH = ds.PW = 0
        for d in (dz, ds):
            d.precision = self.precision
            d.algo = 0
            d.accumulate = 0
            d.workspace = None
            d.workspace_bytes = 0
            d.mask = None
            d.ld_mask = 0
            d.bias = None
            d.relu = 0
        return dz, ds

    @staticmethod
    def _sp_pack_mode(sp, half):
        """clx_pack_mode of a sub-pixel layer's forward weights: plain, F(4x4), or F(4x4) in the fused kernel's layout"""
        if half == "skip":
            return 7 if sp["fused_skip"] else 4 if sp["wino_skip"] else 0
        return 7 if sp["fused_z"] else 4 if sp["wino"] else 0

    def _sp_pack(self, layer, sp, w, need_dgrad, st):
        # one launch: the skip half's weights and the phase-summed weights of the upsampled half
        # (_phase_weights is the same algebra in torch ops, kept as the CPU-testable statement)
        wv = w.detach()
        if not wv.is_contiguous():
            wv = wv.contiguous()
        w_skip, weff = sp["w_skip"], sp["weff"]
        _clx.call("clx_subpixel_split_weights", _clx.ptr(wv), _clx.ptr(w_skip), _clx.ptr(weff), layer.cout,
                  layer.cin, sp["C0"], sp["N"], *layer.kernel, *sp["fac"], st)
        _clx.call("clx_pack_weights", _clx.ptr(w_skip), _clx.ptr(sp["wp_skip_fwd"]), layer.cout, sp["C0"],
                  layer.taps, sp["C0p"], sp["N"], self._sp_pack_mode(sp, "skip"), st)
        _clx.call("clx_pack_weights", _clx.ptr(weff), _clx.ptr(sp["wp_z_fwd"]), sp["P"] * sp["N"], sp["C1"],
                  sp["ztaps"], sp["C1p"], sp["P"] * sp["N"], self._sp_pack_mode(sp, "z"), st)
        if need_dgrad:
            _clx.call("clx_pack_weights", _clx.ptr(w_skip), _clx.ptr(sp["wp_skip_dgrad"]), layer.cout, sp["C0"],
                      layer.taps, sp["C0p"], sp["N"], 5 if sp["wino_skip_dgrad"] else 1, st)
            _clx.call("clx_pack_weights", _clx.ptr(weff), _clx.ptr(sp["wp_z_dgrad"]), sp["P"] * sp["N"],
                      sp["C1"], sp["ztaps"], sp["C1p"], sp["P"] * sp["N"], 5 if sp["wino"] else 1, st)

    def _sp_forward(self, layer, sp, bias, st):
        dz, ds = self._sp_descs(layer, sp)
        zbuf = self.buf[sp["zname"]]
        self._set_wpack(dz, sp["wp_z_fwd"])
        dz.out = zbuf.data_ptr()
        dz.ld_out = sp["P"] * sp["N"]
        sp["_v_fresh"] = False
        if sp["fused_z"]:
            self._use_workspace(dz, 3)
        elif sp["wino"]:
            self._use_workspace(dz, sp["wino"])
            if self.keep and "vcache" in sp:
                dz.vcache = sp["vcache"].data_ptr()
                sp["_v_fresh"] = True
        _clx.call("clx_conv_fwd", ctypes.byref(dz), st)
        out = self.buf[layer.out]
        zs = sp["zshape"]
        _clx.call("clx_depth_to_space", _clx.ptr(zbuf), sp["P"] * sp["N"], _clx.ptr(out), sp["N"], self.B,
                  zs[0], zs[1], zs[2], sp["N"], *sp["fac"], st)
        ds.N = layer.cout
        self._set_wpack(ds, sp["wp_skip_fwd"])
        ds.bias = bias.data_ptr() if bias is not None else None
        ds.relu = 1 if layer.relu else 0
        ds.accumulate = 1
        ds.out = out.data_ptr()
        ds.ld_out = sp["N"]
        if layer.relu and self.keep:
            self._set_gate_out(ds, layer.out)
        sp["_vskip_fresh"] = False
        if sp["fused_skip"]:
            self._use_workspace(ds, 3)
        elif sp["wino_skip"]:
            self._use_workspace(ds, sp["wino_skip"])
            if self.keep and "vcache_skip" in sp:
                ds.vcache = sp["vcache_skip"].data_ptr()
                sp["_vskip_fresh"] = True
        _clx.call("clx_conv_fwd", ctypes.byref(ds), st)

    def _sp_wgrad(self, layer, sp, dy, gw, gb, st):
        """weight/bias gradient of a sub-pixel layer (added into the layer's slices of self.dwpack); returns
        unpack(stream), which unpacks both halves and folds them into `gw`."""
        dz, ds = self._sp_descs(layer, sp)
        zs, PN = sp["zshape"], sp["P"] * sp["N"]
        # dZ = space_to_depth(dY)
        dzbuf = self.gbuf[sp["zname"]]
        _clx.call("clx_space_to_depth", _clx.ptr(dy), sp["N"], _clx.ptr(dzbuf), PN, self.B,
                  zs[0], zs[1], zs[2], sp["N"], *sp["fac"], st)
        # weight gradients (dw_skip / dw_z are slices of self.dwpack: zeroed with it)
        ds.N = sp["N"]
        if sp["wino_skip"]:
            self._use_workspace(ds, sp["wino_skip"])
            if sp.get("_vskip_fresh"):
                ds.vcache = sp["vcache_skip"].data_ptr()
                ds.vcache_valid = 1
        self._wgrad(ds, dy, sp["N"], sp["dw_skip"], gb, layer.cout, st)
        if sp["wino"]:
            self._use_workspace(dz, sp["wino"])
            if sp.get("_v_fresh"):
                dz.vcache = sp["vcache"].data_ptr()
                dz.vcache_valid = 1
            if self.dycache is not None:
                dz.dy_vcache = self.dycache.data_ptr()
        self._wgrad(dz, dzbuf, PN, sp["dw_z"], None, 0, st)
        g_skip, g_z = sp["g_skip"], sp["g_z"]

        def unpack(st):
            if sp["wino_skip"]:
                _clx.call("clx_unpack_wgrad_wino", _clx.ptr(sp["dw_skip"]), _clx.ptr(g_skip), layer.cout, sp["C0"],
                          sp["N"], sp["C0p"], 4, 3, layer.kernel[0], st)
            else:
                _clx.call("clx_unpack_wgrad", _clx.ptr(sp["dw_skip"]), _clx.ptr(g_skip), layer.cout, sp["C0"],
                          layer.taps, sp["N"], sp["C0p"], st)
            if sp["wino"]:
                _clx.call("clx_unpack_wgrad_wino", _clx.ptr(sp["dw_z"]), _clx.ptr(g_z), PN, sp["C1"], PN, sp["C1p"],
                          4, 2, sp["zk"][0], st)
            else:
                _clx.call("clx_unpack_wgrad", _clx.ptr(sp["dw_z"]), _clx.ptr(g_z), PN, sp["C1"], sp["ztaps"], PN,
                          sp["C1p"], st)
            # adjoint of the weight split (one launch; _fold_phase_grads states the same in torch ops)
            _clx.call("clx_subpixel_fold_grads", _clx.ptr(g_skip), _clx.ptr(g_z), _clx.ptr(gw), layer.cout, layer.cin,
                      sp["C0"], sp["N"], *layer.kernel, *sp["fac"], st)
        return unpack

    def _sp_dgrad(self, layer, sp, dy, st):
        """both data gradients of a sub-pixel layer (after _sp_wgrad: it reads the transformed dZ that call left
        in self.dycache); returns the skip gradient buffer (pre-gate, full skip-crop grid)."""
        dzbuf = self.gbuf[sp["zname"]]
        # data gradient of the skip branch (gated later, together with the max-pool gradient)
        dskip = self.gbuf["dskip%d" % sp["level"]]
        dd = self._dgrad_desc(layer, dy)
        dd.N = sp["C0p"]
        self._set_wpack(dd, sp["wp_skip_dgrad"])
        dd.mask = None
        dd.ld_mask = 0
        dd.out = dskip.data_ptr()
        dd.ld_out = sp["C0p"]
        if sp["wino_skip_dgrad"]:
            self._use_workspace(dd, sp["wino_skip_dgrad"])
        _clx.call("clx_conv_fwd", ctypes.byref(dd), st)
        # data gradient of the low-res tensor straight from dZ (replaces upsample backward)
        dl = self._sp_low_dgrad_desc(layer, sp, dzbuf)
        self._set_wpack(dl, sp["wp_z_dgrad"])
        if sp["wino"]:
            self._use_workspace(dl, sp["wino"])
            if self.dycache is not None:       # written by the weight-gradient call on dZ above
                dl.vcache = self.dycache.data_ptr()
                dl.vcache_valid = 1
        _clx.call("clx_conv_fwd", ctypes.byref(dl), st)
        return dskip

    def _sp_low_dgrad_desc(self, layer, sp, dzbuf):
        """dL/d(low-res tensor) as the transposed 2x2(x2) convolution of dZ, ReLU gate fused."""
        t = self.topo
        zs, PN = sp["zshape"], sp["P"] * sp["N"]
        up_s = layer.sources[1]
        low_shape, low_c = t.shapes[up_s.tensor]
        dl = ClxConvDesc()
        dl.nsrc = 1
        src = ClxSrc()
        src.ptr = dzbuf.data_ptr() if dzbuf is not None else 16      # geometry-only queries never dereference
        src.C = PN
        src.ld = PN
        src.D, src.H, src.W = zs
        src.oz = src.oy = src.ox = 0
        src.fz = src.fy = src.fx = 1
        dl.src[0] = src
        dl.B = self.B
        dl.ID, dl.IH, dl.IW = zs
        dl.KD, dl.KH, dl.KW = sp["zk"]
        dl.PD, dl.PH, dl.PW = (k - 1 for k in sp["zk"])
        dl.N = sp["C1p"]
        dl.bias = None
        dl.relu = 0
        dl.accumulate = 0
        dl.algo = 0
        dl.workspace = None
        dl.workspace_bytes = 0
        dl.precision = self.precision
        if dzbuf is not None:
            self._set_mask(dl, up_s.tensor)                 # ReLU gate of the low-res tensor
        else:                                               # geometry-only query
            dl.mask = self.buf[up_s.tensor].data_ptr()
            dl.ld_mask = pad4(low_c)
        dl.out = self.gbuf[up_s.tensor].data_ptr() if dzbuf is not None else None
        dl.ld_out = pad4(low_c)
        return dl

    # ------------------------------------------------------------- descriptors
    def _desc(self, layer: ConvLayer):
        d = ClxConvDesc()
        d.nsrc = len(layer.sources)
        t = self.topo
        for i, s in enumerate(layer.sources):
            shape, c = t.shapes[s.tensor]
            src = ClxSrc()
            src.ptr = self.buf[s.tensor].data_ptr()
            src.C = pad4(s.channels)
            src.ld = pad4(c)
            src.D, src.H, src.W = shape
            src.oz, src.oy, src.ox = s.crop
            src.fz, src.fy, src.fx = s.factor
            d.src[i] = src
        d.B = self.B
        d.ID, d.IH, d.IW = layer.in_shape
        d.KD, d.KH, d.KW = layer.kernel
        d.PD = d.PH = d.PW = 0
        d.accumulate = 0
        d.algo = 0
        d.workspace = None
        d.workspace_bytes = 0
        d.precision = self.precision
        # a raw image with 1-3 channels is stored padded to 4: tell the first-layer kernels
        d.c_real = layer.sources[0].channels if len(layer.sources) == 1 else 0
        return d

    def _set_gate_out(self, d, name):
        """forward epilogue of the layer producing buffer `name`: also emit the ReLU gates as bits"""
        g = getattr(self, "gate", {}).get(name) if self._bwd_ready else None
        if g is not None:
            d.gate_out = g.data_ptr()
            d.ld_gate = g.shape[1]

    def _set_mask(self, d, name, relu=True):
        """data-gradient epilogue: gate by the ReLU of the layer that produced buffer `name`"""
        g = self.gate.get(name)
        if not relu:
            d.mask, d.ld_mask = None, 0
        elif g is not None:
            d.mask, d.ld_mask = None, 0
            d.mask_bits = g.data_ptr()
            d.ld_mask_bits = g.shape[1]
        else:
            d.mask = self.buf[name].data_ptr()
            d.ld_mask = self.buf[name].shape[1]

    def _use_workspace(self, d, code):
        d.algo = code
        if self.workspace is None:      # (only one-launch fused layers: nothing needs scratch)
            return
        d.workspace = self.workspace.data_ptr()
        d.workspace_bytes = self.workspace.numel() * 4

    # ------------------------------------------------- split precision (CLX_PRECISION=f32x3bf16; csrc/gemm_sp.hip)
    def arena_bytes(self):
        """device bytes of this plan's own activations and scratch (not the packed weights and their planes, which plans of one
        model share): what a second plan of the same shape on another stream takes"""
        ts = list(self.buf.values()) + [self.workspace, self.aplanes, self.dyplanes, getattr(self, "dyplanes2", None)]
        ts += list(self.xplanes.values()) + list(self.vcache.values())
        return sum(t.numel() * t.element_size() for t in ts if t is not None)

    def _register_wplanes(self, wp, n, batch, k):
        """planes for the packed weights `wp` seen as `batch` matrices [n][k] (the B operand of a plain product), if
        the split-precision kernels cover that product"""
        if not self.precision or n % 128 or k % 64 or k < 128 or wp.data_ptr() in self._wplanes:
            return
        nbytes = int(_clx.load().clx_planes_bytes(batch * n, k))
        self._wplanes[wp.data_ptr()] = (wp, torch.empty(nbytes, dtype=torch.uint8, device=self.device), batch * n, k)

    def _set_wpack(self, d, wp):
        d.wpack = wp.data_ptr()
        e = self._wplanes.get(wp.data_ptr())
        d.wplanes = e[1].data_ptr() if e is not None else None

    def _split_wplanes(self, st):
        """the planes of every registered packed-weight tensor, after a (re)packing"""
        for wp, planes, rows, k in self._wplanes.values():
            _clx.call("clx_split_planes", _clx.ptr(wp), k, rows, k, _clx.ptr(planes), st)

    def _pointwise_sp(self, layer: ConvLayer):
        """(forward product, data-gradient product, weight-gradient product) of a 1x1 layer over one plain source in the
        split precision?  The rules of clx_sp_applicable / clx_conv_wgrad."""
        if not self.precision or tuple(layer.kernel) != (1, 1, 1) or len(layer.sources) != 1:
            return False, False, False
        s = layer.sources[0]
        shape, _c = self.topo.shapes[s.tensor]
        if tuple(s.crop) != (0, 0, 0) or tuple(s.factor) != (1, 1, 1) or tuple(shape) != tuple(layer.in_shape):
            return False, False, False
        n, c = pad4(layer.cout), layer.cin_pad
        if layer.cout != n:
            return False, False, False
        rows = self.B * layer.in_shape[0] * layer.in_shape[1] * layer.in_shape[2]
        return (n % 128 == 0 and c % 64 == 0 and c >= 128, c % 128 == 0 and n % 64 == 0 and n >= 128,
                # (the weight-gradient product addresses its operand planes with 32-bit offsets: clx_conv_wgrad's rule)
                n % 128 == 0 and c % 128 == 0 and rows * max(n, c) * 6 < (1 << 32) - (1 << 24))

    def _vfloats(self, a2, tiles, chans):
        """floats of a buffer for `a2` transformed tensors [tiles][chans]: float32, or P3 planes (6 bytes per element,
        rows padded to 64) where the layer may run in the split precision"""
        if not self.precision:
            return a2 * tiles * chans
        return a2 * max(128, (tiles + 63) // 64 * 64) * chans * 3 // 2 + 16

    def _planes_scratch(self, rows, k):
        return torch.empty(int(_clx.load().clx_planes_bytes(rows, k)), dtype=torch.uint8, device=self.device)

    def _dgrad_desc(self, layer: ConvLayer, dy):
        """Data gradient as a convolution of dy (zero padding k-1, flipped transposed weights)."""
        dd = ClxConvDesc()
        dd.nsrc = 1
        src = ClxSrc()
        src.ptr = dy.data_ptr() if dy is not None else 16      # geometry-only queries never dereference
        src.C = pad4(layer.cout)
        src.ld = pad4(layer.cout)
        src.D, src.H, src.W = layer.out_shape
        src.oz = src.oy = src.ox = 0
        src.fz = src.fy = src.fx = 1
        dd.src[0] = src
        dd.B = self.B
        dd.ID, dd.IH, dd.IW = layer.out_shape
        dd.KD, dd.KH, dd.KW = layer.kernel
        dd.PD, dd.PH, dd.PW = (k - 1 for k in layer.kernel)
        dd.N = layer.cin_pad
        dd.bias = None
        dd.relu = 0
        dd.accumulate = 0
        dd.algo = 0
        dd.workspace = None
        dd.workspace_bytes = 0
        dd.precision = self.precision
        return dd

    def _expand_cin(self, layer, w):
        """torch weight (cout, cin, taps) -> (cout, cin_gapped, taps) when a concat source is padded."""
        if len(layer.sources) == 1 or all(s.channels % 4 == 0 for s in layer.sources[:-1]):
            return w, layer.cin
        parts, c0 = [], 0
        for s in layer.sources:
            blk = w[:, c0:c0 + s.channels]
            padw = pad4(s.channels) - s.channels
            if padw:
                blk = torch.cat([blk, blk.new_zeros(blk.shape[0], padw, blk.shape[2])], dim=1)
            parts.append(blk)
            c0 += s.channels
        g = torch.cat(parts, dim=1).contiguous()
        return g, g.shape[1]

    def _compress_cin(self, layer, g):
        """inverse of _expand_cin for gradients: (cout, cin_gapped, taps) -> (cout, cin, taps)."""
        parts, c0 = [], 0
        for s in layer.sources:
            parts.append(g[:, c0:c0 + s.channels])
            c0 += pad4(s.channels)
        return torch.cat(parts, dim=1)

    def _wgrad(self, d, dy, ld_dy, dwp, gb, nbias, st):
        """clx_conv_wgrad; in reproducible mode with ordered slices and the bias gradient from ordered column
        sums of dy (rows = output pixels of the layer, `nbias` real channels)."""
        if not self.deterministic:
            _clx.call("clx_conv_wgrad", ctypes.byref(d), _clx.ptr(dy), ld_dy, _clx.ptr(dwp),
                      _clx.ptr(gb) if gb is not None else None, st)
            return
        need = int(_clx.load().clx_conv_wgrad_turns_bytes(ctypes.byref(d)))
        if need > self._det_turns.numel() * 4:
            self._det_turns = _clx.zeros(need // 4 + 1, torch.int32, self.device)
        d.det_turns = self._det_turns.data_ptr()
        _clx.call("clx_conv_wgrad", ctypes.byref(d), _clx.ptr(dy), ld_dy, _clx.ptr(dwp), None, st)
        if gb is not None:
            _clx.call("clx_colsum_ordered", _clx.ptr(dy), ld_dy, dy.shape[0], nbias, _clx.ptr(gb),
                      _clx.ptr(self._det_colsum), st)

    def _unpack_step(self, layer, dwp, gw, wino_w):
        """unpack(stream) for one layer: packed weight gradient `dwp` -> torch-layout gradient `gw`."""
        def unpack(st):
            if wino_w:
                _clx.call("clx_unpack_wgrad_wino", _clx.ptr(dwp), _clx.ptr(gw), layer.cout, layer.cin,
                          pad4(layer.cout), layer.cin_pad, WINO_TILE[wino_w], 3, layer.kernel[0], st)
            elif len(layer.sources) == 1 or all(s.channels % 4 == 0 for s in layer.sources[:-1]):
                _clx.call("clx_unpack_wgrad", _clx.ptr(dwp), _clx.ptr(gw), layer.cout, layer.cin,
                          layer.taps, pad4(layer.cout), layer.cin_pad, st)
            else:       # (torch ops: they run on torch's current stream, which is the caller's `st`)
                tmp = torch.empty((layer.cout, layer.cin_pad, layer.taps), dtype=torch.float32,
                                  device=self.device)
                _clx.call("clx_unpack_wgrad", _clx.ptr(dwp), _clx.ptr(tmp), layer.cout, layer.cin_pad,
                          layer.taps, pad4(layer.cout), layer.cin_pad, st)
                gw.copy_(self._compress_cin(layer, tmp).reshape(gw.shape))
        return unpack

    def _chain_forward(self, a, b, params, st):
        """y1 = relu(x w1^T + b1), y2 = act(y1 w2^T + b2) in one launch (clx_chain64_fwd)."""
        M = self.B * a.in_shape[0] * a.in_shape[1] * a.in_shape[2]
        x = self.buf[a.sources[0].tensor]
        y1, y2 = self.buf[a.out], self.buf[b.out]
        g1 = g2 = None
        if self.keep and self._bwd_ready:
            g1 = self.gate.get(a.out)
            g2 = self.gate.get(b.out) if b.relu else None
        b1, b2 = params[2 * a.param_index + 1], params[2 * b.param_index + 1]
        _clx.call("clx_chain64_fwd", _clx.ptr(x), x.shape[1], M, _clx.ptr(self.wpack_fwd[a.name]), _clx.ptr(b1),
                  _clx.ptr(y1) if self.keep else None, y1.shape[1], _clx.ptr(g1), g1.shape[1] if g1 is not None else 0,
                  _clx.ptr(self.wpack_fwd[b.name]), _clx.ptr(b2), b.cout, 1 if b.relu else 0, _clx.ptr(y2),
                  y2.shape[1], _clx.ptr(g2), g2.shape[1] if g2 is not None else 0, st)

    def _chain_backward(self, a, b, prev, grads, st):
        """Both data gradients, both weight gradients and both bias gradients of the pair in one launch
        (clx_chain64_bwd); the gradient w.r.t. the middle tensor is never written.  Returns unpack(stream)."""
        M = self.B * a.in_shape[0] * a.in_shape[1] * a.in_shape[2]
        dp2 = self.gbuf[b.out]
        x, y1 = self.buf[a.sources[0].tensor], self.buf[a.out]
        dp0 = self.gbuf[prev.out]
        n2p = pad4(b.cout)
        dw2 = self.dwpack[self.dw_off[b.name]:self.dw_off[b.name] + n2p * 64]
        dw1 = self.dwpack[self.dw_off[a.name]:self.dw_off[a.name] + 64 * 64]
        _clx.call("clx_chain64_bwd", _clx.ptr(dp2), dp2.shape[1], b.cout, _clx.ptr(y1), y1.shape[1], _clx.ptr(x),
                  x.shape[1], 1 if prev.relu else 0, M, _clx.ptr(self.wpack_dgrad[b.name]),
                  _clx.ptr(self.wpack_dgrad[a.name]), _clx.ptr(dp0), dp0.shape[1], _clx.ptr(dw2),
                  _clx.ptr(grads[2 * b.param_index + 1]), _clx.ptr(dw1), _clx.ptr(grads[2 * a.param_index + 1]), st)

        def unpack(st):
            for layer, dwp in ((b, dw2), (a, dw1)):
                _clx.call("clx_unpack_wgrad", _clx.ptr(dwp), _clx.ptr(grads[2 * layer.param_index]), layer.cout,
                          layer.cin, 1, pad4(layer.cout), layer.cin_pad, st)
        return unpack

    def _pack_layer(self, layer, w, need_dgrad, st):
        wv = w.detach().reshape(layer.cout, layer.cin, layer.taps)
        if not wv.is_contiguous():
            wv = wv.contiguous()
        wv, cin_eff = self._expand_cin(layer, wv)
        algo = self.algo[layer.name]
        _clx.call("clx_pack_weights", _clx.ptr(wv), _clx.ptr(self.wpack_fwd[layer.name]),
                  layer.cout, cin_eff, layer.taps, layer.cin_pad, pad4(layer.cout),
                  WINO_PACK_FWD.get(algo["fwd"], 0), st)
        if need_dgrad and layer.name in self.wpack_dgrad:
            _clx.call("clx_pack_weights", _clx.ptr(wv), _clx.ptr(self.wpack_dgrad[layer.name]),
                      layer.cout, cin_eff, layer.taps, layer.cin_pad, pad4(layer.cout),
                      6 if layer.name in self.adjoint else WINO_PACK_DGRAD.get(algo["dgrad"], 1), st)

    def _pack_batched(self, params, need_dgrad, st):
        """Every packing of the step in ONE launch (clx_pack_weights_batch): the job table — the arguments of
        the ~40 clx_pack_weights calls — is built once per plan and lives on the device; it is rebuilt when a
        parameter's storage moves.  The sub-pixel layers' weight split runs first (its outputs are sources of
        jobs); layers whose weights need a gapped copy (odd channel counts in a concatenation) keep their calls."""
        from .._clx import ClxPackJob

        sig = (bool(need_dgrad),) + tuple(params[2 * layer.param_index].data_ptr() for layer in self.topo.convs)
        cache = getattr(self, "_pack_table", None)
        if cache is None or cache["sig"] != sig:
            jobs, singles = [], []

            def job(src, dst, cout, cin, taps, cin_pad, cout_pad, mode):
                jobs.append(ClxPackJob(src.data_ptr(), dst.data_ptr(), cout, cin, taps, cin_pad, cout_pad, mode))
                if mode in (0, 1):
                    return (cout if mode == 0 else cin_pad) * taps * (cin_pad if mode == 0 else cout_pad)
                if mode == 7:
                    return cout_pad * cin_pad
                rows, cols = (cin_pad, cout_pad) if mode in (3, 5, 6) else (cout_pad, cin_pad)
                return rows * (3 if taps == 27 else 2 if taps == 8 else 1) * cols

            biggest = 1
            for layer in self.topo.convs:
                w = params[2 * layer.param_index]
                if layer.name in self.subpixel:
                    sp = self.subpixel[layer.name]
                    PN = sp["P"] * sp["N"]
                    biggest = max(biggest, job(sp["w_skip"], sp["wp_skip_fwd"], layer.cout, sp["C0"], layer.taps,
                                               sp["C0p"], sp["N"], self._sp_pack_mode(sp, "skip")),
                                  job(sp["weff"], sp["wp_z_fwd"], PN, sp["C1"], sp["ztaps"], sp["C1p"], PN,
                                      self._sp_pack_mode(sp, "z")))
                    if need_dgrad:
                        biggest = max(biggest, job(sp["w_skip"], sp["wp_skip_dgrad"], layer.cout, sp["C0"], layer.taps,
                                                   sp["C0p"], sp["N"], 5 if sp["wino_skip_dgrad"] else 1),
                                      job(sp["weff"], sp["wp_z_dgrad"], PN, sp["C1"], sp["ztaps"], sp["C1p"], PN,
                                          5 if sp["wino"] else 1))
                    continue
                gapped = len(layer.sources) > 1 and not all(s.channels % 4 == 0 for s in layer.sources[:-1])
                if gapped or not w.is_contiguous():
                    singles.append(layer)
                    continue
                algo = self.algo[layer.name]
                biggest = max(biggest, job(w, self.wpack_fwd[layer.name], layer.cout, layer.cin, layer.taps,
                                           layer.cin_pad, pad4(layer.cout), WINO_PACK_FWD.get(algo["fwd"], 0)))
                if need_dgrad and layer.name in self.wpack_dgrad:
                    biggest = max(biggest, job(w, self.wpack_dgrad[layer.name], layer.cout, layer.cin, layer.taps,
                                               layer.cin_pad, pad4(layer.cout),
                                               6 if layer.name in self.adjoint else WINO_PACK_DGRAD.get(algo["dgrad"], 1)))
            table = None
            if jobs:
                arr = (ClxPackJob * len(jobs))(*jobs)
                host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
                table = host.to(self.device)
            cache = self._pack_table = dict(sig=sig, table=table, njobs=len(jobs), biggest=int(biggest), singles=singles)
        for layer in self.topo.convs:
            if layer.name in self.subpixel:
                sp = self.subpixel[layer.name]
                wv = params[2 * layer.param_index].detach()
                if not wv.is_contiguous():
                    wv = wv.contiguous()
                _clx.call("clx_subpixel_split_weights", _clx.ptr(wv), _clx.ptr(sp["w_skip"]), _clx.ptr(sp["weff"]),
                          layer.cout, layer.cin, sp["C0"], sp["N"], *layer.kernel, *sp["fac"], st)
        if cache["table"] is not None:
            _clx.call("clx_pack_weights_batch", _clx.ptr(cache["table"]), cache["njobs"], cache["biggest"], st)
        for layer in cache["singles"]:
            self._pack_layer(layer, params[2 * layer.param_index], need_dgrad, st)
        self._split_wplanes(st)

    def pack_weights(self, params, version, need_dgrad):
        """(Re)pack weights when the parameters changed (version = tuple of tensor versions)."""
        if need_dgrad and not self._bwd_ready:
            self._alloc_backward()
        key = (version, need_dgrad)
        if self._packed_version is not None and self._packed_version[0] == version and \
                (self._packed_version[1] or not need_dgrad):
            return
        st = _clx.stream_ptr(self.device)
        if os.environ.get("CLX_PACK_BATCH", "1") != "0":
            self._pack_batched(params, need_dgrad, st)
            self._packed_version = key
            return
        for layer in self.topo.convs:
            w = params[2 * layer.param_index]
            if layer.name in self.subpixel:
                self._sp_pack(layer, self.subpixel[layer.name], w, need_dgrad, st)
                continue
            self._pack_layer(layer, w, need_dgrad, st)
        self._split_wplanes(st)
        self._packed_version = key

    # ----------------------------------------------------------------- forward
    # ------------------------------------------------- changed rows of noisy copies (csrc/sparse_rows.hip)
    def pointwise_prefix(self):
        """(first convolution, [the 1x1 layers straight behind it]) if the forward order starts that way with plain
        launches — conv_pass.0 (k x k on the image), conv_pass.2, conv_pass.4 (1 x 1) of the first level — else None.
        These are the layers a noisy copy of an image shares with the clean image outside the window-dilated set of
        its noise pixels (UNetModel._forward_chunks)."""
        ops = self.topo.fwd_order
        if not ops or not isinstance(ops[0], ConvLayer):
            return None
        special = lambda op: op.name in self.chains or op.name in self.chain_second or op.name in self.subpixel
        first = ops[0]
        if special(first) or len(first.sources) != 1 or first.sources[0].tensor != "raw":
            return None
        # On the changed-rows path the copies' tensors between `first` and the LAST layer of the tail are never written
        # (only the last one is rebuilt by broadcast + scatter): every tensor in between must have the next layer of the
        # prefix as its ONLY reader — a topology in which one of them is also a skip connection or pooled would read
        # stale rows (today's build_topology never makes one; find_chain_pairs guards its pairs the same way)
        readers = tensor_consumers(self.topo)
        tail, src = [], first.out
        for op in ops[1:]:
            if not isinstance(op, ConvLayer) or op.taps != 1 or special(op) or self.algo[op.name]["fwd"]:
                break
            s = op.sources[0]
            if len(op.sources) != 1 or s.tensor != src or tuple(s.crop) != (0, 0, 0) or tuple(s.factor) != (1, 1, 1):
                break
            if readers.get(src, 0) != 1:
                break
            tail.append(op)
            src = op.out
        return (first, tail) if tail else None

    def first_layer_on_rows(self):
        """True if the first layer of pointwise_prefix can be computed for a list of output pixels by clx_grey_rows: a
        one-channel image under a 3 x 3 (x 3) kernel — the layers clx_conv_fwd gives to conv_grey_fwd_kernel."""
        prefix = self.pointwise_prefix()
        if prefix is None or os.environ.get("CLX_SPARSE_FIRST", "1") == "0":
            return False
        first = prefix[0]
        return (self.topo.in_channels == 1 and tuple(first.kernel) in ((1, 3, 3), (3, 3, 3)) and first.cout % 4 == 0
                and first.in_shape[0] >= first.kernel[0])

    def tiled_layer_behind_prefix(self):
        """The 2-D Winograd layer that reads the last 1x1 layer of pointwise_prefix (conv_pass.6 of the first level), or
        None: its output tiles can be computed for a list of tiles only (clx_conv_desc.tile_list)."""
        prefix = self.pointwise_prefix()
        if prefix is None:
            return None
        first, tail = prefix
        ops = self.topo.fwd_order
        k = 1 + len(tail)
        if k >= len(ops) or not isinstance(ops[k], ConvLayer):
            return None
        op = ops[k]
        code = self.algo[op.name]["fwd"]
        if (not code or op.kernel[0] != 1 or op.in_shape[0] != 1 or op.name in self.subpixel or op.name in self.chains
                or op.name in self.chain_second or len(op.sources) != 1):
            return None
        s = op.sources[0]
        if s.tensor != tail[-1].out or tuple(s.crop) != (0, 0, 0) or tuple(s.factor) != (1, 1, 1):
            return None
        return op, WINO_TILE[code]

    def _compact(self, slot, rows, width):
        """grow-only scratch for `rows` compact rows of `width` floats"""
        bufs = self.__dict__.setdefault("_compact_bufs", {})
        b = bufs.get(slot)
        if b is None or b.shape[0] < rows or b.shape[1] != width:
            b = bufs[slot] = torch.empty((max(rows, 1), width), dtype=torch.float32, device=self.device)
        return b

    def forward_prefix(self, raw, params, nlayers):
        """the first convolution and the `nlayers` layers behind it on `raw` -> the last one's output rows
        (B * pixels, padded channels), in this plan's buffer (a Winograd layer with a fused pooling also leaves the
        pooled tensor in its buffer)"""
        t = self.topo
        st = _clx.stream_ptr(self.device)
        npix_in = t.in_shape[0] * t.in_shape[1] * t.in_shape[2]
        raw = raw.contiguous()
        _clx.call("clx_planar_to_pixel", _clx.ptr(raw), _clx.ptr(self.buf["raw"]), self.B,
                  t.in_channels, npix_in, pad4(t.in_channels), st)
        for op in t.fwd_order[:1 + nlayers]:
            self._conv_forward(op, params, st)
        return self.buf[t.fwd_order[nlayers].out]

    def _conv_forward(self, op, params, st, tiles=None):
        """one plain convolution layer of the forward pass (tiles = (int32 tensor, count): a Winograd layer computes the
        listed output tiles only)"""
        d = self._desc(op)
        d.N = op.cout
        self._set_wpack(d, self.wpack_fwd[op.name])
        b = params[2 * op.param_index + 1]
        d.bias = b.data_ptr() if b is not None else None
        d.relu = 1 if op.relu else 0
        d.mask = None
        d.ld_mask = 0
        d.out = self.buf[op.out].data_ptr()
        d.ld_out = pad4(op.cout)
        if op.relu and self.keep:
            self._set_gate_out(d, op.out)
        if self.precision and not self.algo[op.name]["fwd"] and self._pointwise_sp(op)[0]:
            xp = self.xplanes.get(op.name) if self.keep and self._bwd_ready and tiles is None else None
            d.aplanes = (xp if xp is not None else self.aplanes).data_ptr()
            if xp is not None:
                # (planes the layer before this one has already written in its epilogue: no split pass)
                d.aplanes_valid = 1 if op.name in self._xplanes_fresh else 0
                self._xplanes_fresh.add(op.name)
                # ... and this layer's epilogue writes the planes of the 1x1 layer that reads its output
                nxt = self._pointwise_reader.get(op.out)
                if nxt is not None and nxt.name in self.xplanes and os.environ.get("CLX_SP_EPILOGUE_PLANES", "1") != "0":
                    d.out_planes = self.xplanes[nxt.name].data_ptr()
                    self._xplanes_fresh.add(nxt.name)
        if self.algo[op.name]["fwd"]:
            self._use_workspace(d, self.algo[op.name]["fwd"])
            if self.keep and self._bwd_ready and op.name in self.vcache:
                d.vcache = self.vcache[op.name].data_ptr()
                self._vcache_fresh.add(op.name)
            pool = self.fused_pool.get(op.name)
            if pool is not None:
                d.pool_out = self.buf[pool.out].data_ptr()
                d.ld_pool = pad4(pool.channels)
        if tiles is not None:
            d.tile_list = tiles[0].data_ptr()
            d.tile_count = int(tiles[1])
        _clx.call("clx_conv_fwd", ctypes.byref(d), st)

    def _tiles_forward(self, op, params, sparse, st):
        """The Winograd layer behind the 1x1 layers for a chunk of noisy copies: the clean image's output (and pooled
        output) under every copy, then the CHANGED tiles of each copy — those whose input window holds a changed row —
        through the transforms and the batched products (clx_conv_desc.tile_list).  A tile's output depends on its own
        input window alone: the dense computation's bits."""
        dense = self.buf[op.out]
        clean = sparse["clean_tile_rows"]
        assert clean.shape[1] == dense.shape[1] and clean.shape[0] * self.B == dense.shape[0]
        _clx.call("clx_broadcast_rows", _clx.ptr(clean), clean.numel(), _clx.ptr(dense), self.B, st)
        pool = self.fused_pool.get(op.name)
        if pool is not None:
            pooled, clean_pooled = self.buf[pool.out], sparse["clean_pool_rows"]
            assert clean_pooled.shape[0] * self.B == pooled.shape[0]
            _clx.call("clx_broadcast_rows", _clx.ptr(clean_pooled), clean_pooled.numel(), _clx.ptr(pooled), self.B, st)
        if int(sparse["ntiles"]) > 0:
            self._conv_forward(op, params, st, tiles=(sparse["tiles"], sparse["ntiles"]))

    def _pointwise_on_rows(self, tail, params, sparse, st):
        """The 1x1 layers `tail` for a chunk of noisy copies: once on the clean image (done by the caller: sparse
        ["clean_rows"]) and here on the copies' CHANGED rows — gathered from the first convolution's dense output, run
        through the layers as a (1, 1, n) image, scattered over the broadcast clean rows.  A row of a 1x1 layer depends on
        its own input row alone, so the tensor is the dense computation's, bit for bit."""
        n, rows = int(sparse["n"]), sparse["rows"]
        src = self.buf[tail[0].sources[0].tensor]
        width = src.shape[1]
        cur = None
        if n > 0:
            cur = self._compact(0, n, width)
            if sparse.get("noisy") is not None:
                # one-channel image: the first layer itself on the changed rows (clx_grey_rows: the dense kernel's
                # arithmetic) — the dense first-layer tensor of the copies is never written
                first = self.topo.fwd_order[0]
                b = params[2 * first.param_index + 1]
                _clx.call("clx_grey_rows", _clx.ptr(sparse["noisy"]), self.B, *first.in_shape, first.kernel[0],
                          _clx.ptr(rows), n, _clx.ptr(self.wpack_fwd[first.name]), _clx.ptr(b) if b is not None else None,
                          1 if first.relu else 0, first.cout, _clx.ptr(cur), width, st)
            else:
                _clx.call("clx_gather_rows", _clx.ptr(src), width, _clx.ptr(rows), n, width, _clx.ptr(cur), width, st)
            for k, op in enumerate(tail):
                y = self._compact(1 + k % 2, n, pad4(op.cout))
                d = ClxConvDesc()
                d.nsrc = 1
                src_d = ClxSrc()
                src_d.ptr = cur.data_ptr()
                src_d.C = src_d.ld = cur.shape[1]
                src_d.D, src_d.H, src_d.W = 1, 1, n
                src_d.oz = src_d.oy = src_d.ox = 0
                src_d.fz = src_d.fy = src_d.fx = 1
                d.src[0] = src_d
                d.B = 1
                d.ID, d.IH, d.IW = 1, 1, n
                d.KD = d.KH = d.KW = 1
                d.PD = d.PH = d.PW = 0
                d.N = op.cout
                self._set_wpack(d, self.wpack_fwd[op.name])
                d.precision = self.precision
                if self.precision and self.aplanes is not None:       # (n <= the dense tensor's rows: the scratch fits)
                    d.aplanes = self.aplanes.data_ptr()
                b = params[2 * op.param_index + 1]
                d.bias = b.data_ptr() if b is not None else None
                d.relu = 1 if op.relu else 0
                d.out = y.data_ptr()
                d.ld_out = y.shape[1]
                _clx.call("clx_conv_fwd", ctypes.byref(d), st)
                cur = y
        dense = self.buf[tail[-1].out]
        clean = sparse["clean_rows"]
        assert clean.shape[1] == dense.shape[1] and clean.shape[0] * self.B == dense.shape[0]
        _clx.call("clx_broadcast_rows", _clx.ptr(clean), clean.numel(), _clx.ptr(dense), self.B, st)
        if n > 0:
            _clx.call("clx_scatter_rows", _clx.ptr(cur), cur.shape[1], _clx.ptr(rows), n, dense.shape[1],
                      _clx.ptr(dense), dense.shape[1], st)

    def forward(self, raw, params, out=None, on_op=None, sparse=None):
        """raw: (B, C, *spatial) f32 on device -> offsets (B, out_channels, *out_spatial), written into `out`
        (contiguous, that shape) when given.  on_op(i): called after the launches of the i-th operation of the forward
        order are enqueued (a second stream can be started behind a chosen point of this one).
        sparse: {"clean_rows", "rows", "n"} — the batch is noisy copies of ONE image whose pointwise prefix
        (pointwise_prefix) was computed on the clean image: the 1x1 layers run on the changed rows only."""
        t = self.topo
        st = _clx.stream_ptr(self.device)
        sparse_tail = self.pointwise_prefix()[1] if sparse is not None else None
        self._vcache_fresh = set()      # Winograd layers whose V this forward left in self.vcache
        self._xplanes_fresh = set()     # 1x1 layers whose input planes this forward left in self.xplanes
        npix_in = t.in_shape[0] * t.in_shape[1] * t.in_shape[2]
        raw = raw.contiguous()
        rows_only_first = sparse is not None and self.first_layer_on_rows()
        if rows_only_first:
            sparse = dict(sparse, noisy=raw)                # the first layer runs on the changed rows only
        else:
            _clx.call("clx_planar_to_pixel", _clx.ptr(raw), _clx.ptr(self.buf["raw"]), self.B,
                      t.in_channels, npix_in, pad4(t.in_channels), st)
        for op_index, op in enumerate(t.fwd_order):
            if on_op is not None and op_index > 0:
                on_op(op_index - 1)
            if op_index == 0 and rows_only_first:
                continue
            if isinstance(op, ConvLayer) and op.name in self.chain_second:
                continue                                    # computed with its predecessor
            if isinstance(op, ConvLayer) and op.name in self.chains:
                self._chain_forward(*self.chains[op.name], params, st)
            elif isinstance(op, ConvLayer) and op.name in self.subpixel:
                self._sp_forward(op, self.subpixel[op.name], params[2 * op.param_index + 1], st)
            elif isinstance(op, ConvLayer) and sparse_tail and op is sparse_tail[0]:
                self._pointwise_on_rows(sparse_tail, params, sparse, st)
            elif isinstance(op, ConvLayer) and sparse_tail and any(op is q for q in sparse_tail):
                continue                                    # computed with the first 1x1 layer of the prefix
            elif isinstance(op, ConvLayer) and sparse is not None and sparse.get("tiles") is not None \
                    and op is sparse["tile_op"]:
                self._tiles_forward(op, params, sparse, st)
            elif isinstance(op, ConvLayer):
                self._conv_forward(op, params, st)
            elif any(p is op for p in self.fused_pool.values()):
                continue                                    # written by the producing layer's output transform
            else:
                D, H, W = op.in_shape
                _clx.call("clx_maxpool_fwd", _clx.ptr(self.buf[op.src]), _clx.ptr(self.buf[op.out]),
                          self.B, D, H, W, pad4(op.channels), *op.factor, st)
        npix_out = t.out_shape[0] * t.out_shape[1] * t.out_shape[2]
        spatial = t.out_shape[3 - t.nd:]
        if out is None:
            out = torch.empty((self.B, t.out_channels) + tuple(spatial), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and tuple(out.shape) == (self.B, t.out_channels) + tuple(spatial)
        _clx.call("clx_pixel_to_planar", _clx.ptr(self.buf["h1"]), _clx.ptr(out), self.B,
                  t.out_channels, npix_out, pad4(t.out_channels), st)
        return out

    # ---------------------------------------------------------------- backward
    def backward(self, dout, params, grads, on_layer_done=None, flat_grad=None):
        """dout: (B, out_channels, *out_spatial) gradient of the loss w.r.t. forward()'s result.
        grads: list aligned with params; every entry is OVERWRITTEN with the gradient.
        flat_grad: the flat buffer the entries of `grads` are views of, if there is one.
        on_layer_done(param_index): called once the kernels that write a layer's weight and bias
        gradient are enqueued (layers finish in reverse forward order: the data-parallel step
        starts reducing the tail of the flat gradient while the rest is still being computed)."""
        st = _clx.stream_ptr(self.device)
        self.zero_gradients(grads, flat_grad)
        for done, unpack in self.backward_steps(dout, params, grads):
            unpack(st)
            if on_layer_done is not None:
                for i in done:
                    on_layer_done(i)

    def train_pass(self, raw, params, grads, flat_grad, loss_fn, after_loss=None, on_layer_done=None):
        """forward -> loss_fn(offsets, lo, hi) (enqueues the loss of crops lo..hi-1 on the current stream, returns the
        gradient w.r.t. those offsets) -> after_loss() -> backward.  Returns the offsets."""
        out = self.forward(raw, params)
        dout = loss_fn(out, 0, self.B)
        if after_loss is not None:
            after_loss()
        self.backward(dout, params, grads, on_layer_done=on_layer_done, flat_grad=flat_grad)
        return out

    def zero_gradients(self, grads, flat_grad=None):
        """The accumulators the backward kernels add into: packed weight gradients and bias gradients."""
        if flat_grad is not None:          # `grads` tile this buffer: ONE launch for both accumulators
            _clx.zero_many(self.dwpack, flat_grad)
        else:
            _clx.zero_many(self.dwpack)
            for g in grads[1::2]:
                if g is not None:
                    g.zero_()

    def backward_steps(self, dout, params, grads):
        """The backward pass as a generator, one step per layer (or fused pair) in reverse order.  Each step
        enqueues the layer's weight/bias-gradient kernels — they ADD into self.dwpack and the bias gradients, which
        the caller has zeroed (zero_gradients) — and yields (param_indices, unpack): unpack(stream) enqueues the
        launches that turn the packed weight gradient of those layers into `grads`; the layer's data gradient is
        enqueued when the generator is resumed.  DualPlan drives two of these on two streams over one accumulator."""
        t = self.topo
        st = _clx.stream_ptr(self.device)
        assert self._bwd_ready, "pack_weights(need_dgrad=True) must run before backward"
        npix_out = t.out_shape[0] * t.out_shape[1] * t.out_shape[2]
        dout = dout.contiguous()
        _clx.call("clx_planar_to_pixel", _clx.ptr(dout), _clx.ptr(self.gbuf["h1"]), self.B,
                  t.out_channels, npix_out, pad4(t.out_channels), st)

        by_out = {layer.out: layer for layer in t.convs}
        pool_by_out = {p.out: p for p in t.pools}
        r_by_conv0 = {info["conv0"].name: info for info in t.r_info}
        pending_skip = {}   # skip tensor name -> (cat gbuf name, conv0 layer)
        dy_planes_of = {}   # tensor name -> buffer that holds the P3 planes of its gradient (split precision)

        # reverse execution order; gbuf[x] holds dL/d(pre-activation of x)
        for op in reversed(t.fwd_order):
            if isinstance(op, PoolOp):
                continue  # handled when its consumer's data gradient is produced
            layer = op
            dy = self.gbuf[layer.out]
            if layer.name in self.chains:
                continue                                    # done with its successor
            if layer.name in self.chain_second:
                a, b = self.chain_second[layer.name]
                yield (b.param_index, a.param_index), self._chain_backward(a, b, by_out[a.sources[0].tensor], grads, st)
                continue
            if layer.name in self.subpixel:
                sp = self.subpixel[layer.name]
                yield (layer.param_index,), self._sp_wgrad(layer, sp, dy, grads[2 * layer.param_index],
                                                           grads[2 * layer.param_index + 1], st)
                dskip = self._sp_dgrad(layer, sp, dy, st)
                pending_skip[layer.sources[0].tensor] = (dskip, sp["C0p"], layer)
                continue
            # ---- weight + bias gradient
            d = self._desc(layer)
            d.N = pad4(layer.cout)
            gb = grads[2 * layer.param_index + 1]
            off = self.dw_off[layer.name]
            wino_w = self.algo[layer.name]["wgrad"]
            wtaps = wino_taps(wino_w, layer.kernel) if wino_w else layer.taps
            dwp = self.dwpack[off:off + wtaps * pad4(layer.cout) * layer.cin_pad]
            adjoint = layer.name in self.adjoint
            dual = (self.dycache is not None and wino_w and layer.param_index > 0
                    and self.algo[layer.name]["dgrad"] == wino_w and not adjoint)
            if wino_w:
                self._use_workspace(d, wino_w)
                if layer.name in self.vcache and layer.name in self._vcache_fresh:
                    d.vcache = self.vcache[layer.name].data_ptr()
                    d.vcache_valid = 1
                if dual:
                    d.dy_vcache = self.dycache.data_ptr()
            # split precision, 1x1 layers: the planes of this layer's dY — written by the epilogue of the data gradient that
            # produced dY (dy_planes_of, with the bias gradient as that epilogue's column sums), or split by the weight
            # gradient below — serve the weight gradient and the data gradient
            dy_planes = dy_planes_of.pop(layer.out, None)
            dy_ready = dy_planes is not None
            if dy_planes is None and self.dyplanes is not None:
                dy_planes = self.dyplanes
            if not wino_w and self._pointwise_sp(layer)[2] and layer.name in self.xplanes and dy_planes is not None:
                d.aplanes = self.xplanes[layer.name].data_ptr()
                d.aplanes_valid = 1 if layer.name in self._xplanes_fresh else 0
                d.dyplanes = dy_planes.data_ptr()
                d.dyplanes_valid = 1 if dy_ready else 0
                if dy_ready:
                    gb = None                       # (the bias gradient is in already)
                dy_ready = True
            elif dy_ready:
                raise AssertionError("planes of dY were written for a layer whose weight gradient does not read them")
            self._wgrad(d, dy, pad4(layer.cout), dwp, gb, layer.cout, st)
            yield (layer.param_index,), self._unpack_step(layer, dwp, grads[2 * layer.param_index], wino_w)
            # ---- data gradient
            if layer.param_index == 0:
                continue
            dd = self._dgrad_desc(layer, dy)
            self._set_wpack(dd, self.wpack_dgrad[layer.name])
            dgrad_sp = not self.algo[layer.name]["dgrad"] and self._pointwise_sp(layer)[1] and dy_planes is not None
            if dgrad_sp:
                dd.aplanes = dy_planes.data_ptr()
                dd.aplanes_valid = 1 if dy_ready else 0          # (left by the weight gradient above, or by the layer behind)
            if self.algo[layer.name]["dgrad"]:
                self._use_workspace(dd, self.algo[layer.name]["dgrad"])
                if adjoint:                    # A dY A^T was left in the workspace by the weight-gradient call above
                    dd.adjoint = 1
                elif dual:                     # V of dY was written by the weight-gradient call above
                    dd.vcache = self.dycache.data_ptr()
                    dd.vcache_valid = 1
            if len(layer.sources) == 2:
                info = r_by_conv0[layer.name]
                cat = self.gbuf["cat%d" % info["level"]]
                dd.mask = None
                dd.ld_mask = 0
                dd.out = cat.data_ptr()
                dd.ld_out = layer.cin_pad
                _clx.call("clx_conv_fwd", ctypes.byref(dd), st)
                skip_s, up_s = layer.sources
                # upsampled branch -> pre-activation gradient of the low-res tensor
                ushape, uc = t.shapes[up_s.tensor]
                LD, LH, LW = layer.in_shape
                _clx.call("clx_upsample_bwd", _clx.ptr(cat), layer.cin_pad, pad4(skip_s.channels),
                          LD, LH, LW, *up_s.crop, _clx.ptr(self.buf[up_s.tensor]),
                          _clx.ptr(self.gbuf[up_s.tensor]), self.B, ushape[0], ushape[1], ushape[2],
                          pad4(uc), *up_s.factor, st)
                pending_skip[skip_s.tensor] = (cat, layer.cin_pad, layer)
            else:
                s = layer.sources[0]
                if s.tensor in pool_by_out:
                    # input is a pooled tensor: dgrad -> gradient of the pool output (no gate),
                    # then route through the max-pool, add the skip gradient, gate by ReLU.
                    pool = pool_by_out[s.tensor]
                    dd.mask = None
                    dd.ld_mask = 0
                    dd.out = self.gbuf[pool.out].data_ptr()
                    dd.ld_out = pad4(pool.channels)
                    _clx.call("clx_conv_fwd", ctypes.byref(dd), st)
                    cat, ld_cat, rl = pending_skip.pop(pool.src)
                    skip_s = rl.sources[0]
                    D, H, W = pool.in_shape
                    SD, SH, SW = rl.in_shape
                    _clx.call("clx_maxpool_bwd", _clx.ptr(self.buf[pool.src]), _clx.ptr(self.buf[pool.out]),
                              _clx.ptr(self.gbuf[pool.out]), _clx.ptr(cat), ld_cat, SD, SH, SW,
                              *skip_s.crop, _clx.ptr(self.gbuf[pool.src]), self.B, D, H, W,
                              pad4(pool.channels), *pool.factor, st)
                else:
                    prev = by_out[s.tensor]
                    self._set_mask(dd, prev.out, relu=prev.relu)
                    dd.out = self.gbuf[prev.out].data_ptr()
                    dd.ld_out = pad4(prev.cout)
                    if (dgrad_sp and prev.name in self.xplanes and self._pointwise_sp(prev)[2] and prev.name not in self.chains
                            and prev.name not in self.chain_second and prev.cout == pad4(prev.cout)
                            and os.environ.get("CLX_SP_EPILOGUE_PLANES", "1") != "0"):
                        # the epilogue writes the planes of prev's dY and adds prev's bias gradient (its column sums)
                        other = self.dyplanes2 if dy_planes is self.dyplanes else self.dyplanes
                        dd.out_planes = other.data_ptr()
                        gbp = grads[2 * prev.param_index + 1]
                        dd.out_colsum = gbp.data_ptr() if gbp is not None else None
                        dy_planes_of[prev.out] = other
                    _clx.call("clx_conv_fwd", ctypes.byref(dd), st)
        assert not pending_skip


def forward_flops(topo, batch):
    """2 M N K over the convolutions of one forward pass (direct form)."""
    total = 0
    for layer in topo.convs:
        m = batch * layer.out_shape[0] * layer.out_shape[1] * layer.out_shape[2]
        total += 2 * m * layer.cout * layer.cin * layer.taps
    return total


class _Rows:
    """Read-only view of a per-tensor buffer dict of the two halves as full-batch tensors (rows are pixels,
    batch-major: the halves concatenate)."""

    def __init__(self, parts, attr):
        self._parts, self._attr = parts, attr

    def __getitem__(self, name):
        return torch.cat([getattr(p, self._attr)[name] for p in self._parts], dim=0)

    def __contains__(self, name):
        return name in getattr(self._parts[0], self._attr)

    def __bool__(self):
        return bool(getattr(self._parts[0], self._attr))

    def get(self, name, default=None):
        return self[name] if name in self else default

    def keys(self):
        return getattr(self._parts[0], self._attr).keys()


def dual_stream_wanted(topo, batch, keep_activations):
    """Two half batches on two streams (DualPlan)?  Training plans with an even batch whose halves are big enough
    to fill the device (CLX_STREAMS_MIN_GFLOP per half-batch forward pass, default 100: below that the step is
    launch-bound and twice the launches cost more than the overlap returns); never in reproducible mode.
    CLX_STREAMS=1 switches it off."""
    if not keep_activations or batch < 2 or batch % 2 or os.environ.get("CLX_STREAMS", "2") == "1":
        return False
    if os.environ.get("CLX_DETERMINISTIC", "0") == "1":
        return False
    return forward_flops(topo, batch // 2) >= float(os.environ.get("CLX_STREAMS_MIN_GFLOP", "100")) * 1e9


class DualPlan:
    """A training batch as two half batches on two HIP streams.

    A step is a strict chain of launches, each either bound by the matrix cores (the GEMMs) or by HBM (Winograd
    transforms, pooling, fills, the first layer): on one stream the two kinds never overlap.  Two independent half
    batches do — the transforms of one half run under the GEMMs of the other, and the partial last round of one
    half's tiles is filled by the other's.  Both halves use ONE set of packed weights and add their weight and bias
    gradients into ONE set of accumulators (the kernels add with atomics anyway); a layer's packed gradient is
    unpacked on the caller's stream once both halves have passed that layer, which is also when on_layer_done
    fires — the data-parallel buckets leave exactly as they do with one stream.
    Same interface as UNetPlan (pack_weights / forward / backward); CLX_STREAMS=1 keeps one stream."""

    def __init__(self, topo, batch, device, keep_activations):
        assert batch % 2 == 0 and keep_activations
        self.topo, self.B, self.device, self.keep = topo, int(batch), device, True
        self.parts = [UNetPlan(topo, batch // 2, device, True) for _ in range(2)]
        self.streams = [torch.cuda.Stream(device=device) for _ in range(2)]
        self._events = [[], [], []]
        self._shared = False
        self.buf = _Rows(self.parts, "buf")

    def __getattr__(self, name):            # algo, chains, subpixel, gate, ... : the halves agree
        if name in ("parts", "streams"):
            raise AttributeError(name)
        if name == "gbuf":
            return _Rows(self.parts, "gbuf")
        return getattr(self.parts[0], name)

    def pack_weights(self, params, version, need_dgrad):
        a, b = self.parts
        a.pack_weights(params, version, need_dgrad)
        if need_dgrad and not self._shared:
            b._alloc_backward()
            b.share_from(a)
            self._shared = True
        b._packed_version = a._packed_version

    def _fork(self):
        main = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(main)
        return main

    def _join(self, main):
        for s in self.streams:
            main.wait_stream(s)

    def forward(self, raw, params, out=None):
        t = self.topo
        assert self._shared, "pack_weights(need_dgrad=True) must run before forward"
        raw = raw.contiguous()
        if out is None:
            out = torch.empty((self.B, t.out_channels) + tuple(t.out_shape[3 - t.nd:]), dtype=torch.float32,
                              device=self.device)
        h = self.B // 2
        main = self._fork()
        for i, (p, s) in enumerate(zip(self.parts, self.streams)):
            with torch.cuda.stream(s):
                p.forward(raw[i * h:(i + 1) * h], params, out=out[i * h:(i + 1) * h])
        self._join(main)
        return out

    def _event(self, i, k):
        ev = self._events[i]
        while len(ev) <= k:
            ev.append(torch.cuda.Event())
        return ev[k]

    def backward(self, dout, params, grads, on_layer_done=None, flat_grad=None):
        dout = dout.contiguous()
        h = self.B // 2
        self.parts[0].zero_gradients(grads, flat_grad)           # the one set of accumulators, on the caller's stream
        main = self._fork()
        self._backward_halves([dout[:h], dout[h:]], params, grads, on_layer_done, main)
        self._join(main)

    def train_pass(self, raw, params, grads, flat_grad, loss_fn, after_loss=None, on_layer_done=None):
        """UNetPlan.train_pass with each half's loss on its own stream (no join between the forward and the backward
        pass); the accumulators are zeroed on the caller's stream while the halves run their forward passes."""
        t = self.topo
        assert self._shared, "pack_weights(need_dgrad=True) must run before train_pass"
        raw = raw.contiguous()
        out = torch.empty((self.B, t.out_channels) + tuple(t.out_shape[3 - t.nd:]), dtype=torch.float32,
                          device=self.device)
        h = self.B // 2
        main = self._fork()
        douts = []
        # (starting the second half behind operation 0 .. 8 of the first — which gains 1.5-3 % on the inference chunks,
        #  models/unet.py — LOSES 0.3-3 % here, round 4: the step ends at a join and the delay is not recovered)
        for i, (p, s) in enumerate(zip(self.parts, self.streams)):
            with torch.cuda.stream(s):
                o = p.forward(raw[i * h:(i + 1) * h], params, out=out[i * h:(i + 1) * h])
                douts.append(loss_fn(o, i * h, (i + 1) * h))
                self._event(i, 0).record(s)
        self.parts[0].zero_gradients(grads, flat_grad)
        zeroed = self._event(2, 0)
        zeroed.record(main)
        for i, s in enumerate(self.streams):
            main.wait_event(self._event(i, 0))
            s.wait_event(zeroed)
        if after_loss is not None:
            after_loss()
        self._backward_halves(douts, params, grads, on_layer_done, main)
        self._join(main)
        return out

    def _backward_halves(self, douts, params, grads, on_layer_done, main):
        st_main = _clx.stream_ptr(self.device)
        gens = []
        for p, s, d in zip(self.parts, self.streams, douts):
            with torch.cuda.stream(s):
                gens.append(p.backward_steps(d, params, grads))
        k = 1
        while True:
            items = []
            for i, (g, s) in enumerate(zip(gens, self.streams)):
                with torch.cuda.stream(s):
                    item = next(g, None)
                    if item is not None:
                        self._event(i, k).record(s)
                items.append(item)
            if items[0] is None:
                assert items[1] is None
                break
            assert items[1] is not None and items[0][0] == items[1][0]
            for i in range(2):
                main.wait_event(self._event(i, k))
            done, unpack = items[0]                  # (the halves share the accumulators: either closure does)
            unpack(st_main)
            if on_layer_done is not None:
                for idx in done:
                    on_layer_done(idx)
            k += 1
