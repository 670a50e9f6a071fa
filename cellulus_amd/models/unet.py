"""``UNetModel`` — drop-in for ``cellulus/models/unet.py`` running on libclx.

Same constructor, ``forward`` (train / infer modes), ``set_infer`` and
``select_and_add_coordinates`` as the reference (unet.py:9-124).  Parameters
carry the reference's ``state_dict`` names (``backbone.l_conv.<i>.conv_pass.<j>``,
``backbone.r_conv.0.<i>.conv_pass.<j>``, ``head.0``, ``head.2``) in torch layout,
so checkpoints interchange; the modules that own them never execute — every
forward/backward is a sequence of HIP kernel launches driven by ``UNetPlan``.
"""

from typing import List, Tuple

import os

import torch
import torch.nn as nn

import ctypes

from .. import _clx, parallel
from .._clx import ClxConvDesc, ClxSrc
from .plan import DualPlan, UNetPlan, build_topology, dual_stream_wanted, forward_flops, pad4


class _ConvPass(nn.Module):
    """Parameter holder named like funlib's ConvPass (conv_pass.{0,2,4,6})."""

    def __init__(self, cin, cout, kernel_sizes, nd):
        super().__init__()
        conv = nn.Conv2d if nd == 2 else nn.Conv3d
        layers = []
        for k in kernel_sizes:
            layers.append(conv(cin, cout, k))
            layers.append(nn.ReLU())
            cin = cout
        self.conv_pass = nn.Sequential(*layers)


class _Backbone(nn.Module):
    """Parameter holder named like funlib.learn.torch.models.UNet (num_heads = 1)."""

    def __init__(self, in_channels, num_fmaps, fmap_inc_factor, downsample_factors, num_fmaps_out, nd):
        super().__init__()
        L = len(downsample_factors)
        ks = [(3,) * nd, (1,) * nd, (1,) * nd, (3,) * nd]
        self.l_conv = nn.ModuleList([
            _ConvPass(in_channels if i == 0 else num_fmaps * fmap_inc_factor ** (i - 1),
                      num_fmaps * fmap_inc_factor ** i, ks, nd)
            for i in range(L + 1)
        ])
        self.r_conv = nn.ModuleList([nn.ModuleList([
            _ConvPass(num_fmaps * fmap_inc_factor ** i + num_fmaps * fmap_inc_factor ** (i + 1),
                      num_fmaps_out if i == 0 else num_fmaps * fmap_inc_factor ** i, ks, nd)
            for i in range(L)
        ])])


class _UNetFunction(torch.autograd.Function):
    """autograd glue: one node for the whole network."""

    @staticmethod
    def forward(ctx, model, raw, *params):
        plan = model._plan_for(raw, keep=True)
        plan.pack_weights(params, model._param_version(), need_dgrad=True)
        out = plan.forward(raw, params)
        ctx.model = model
        ctx.plan = plan
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, dout):
        model, plan = ctx.model, ctx.plan
        grads = model._grad_views()
        flat = model._flat_grad
        if any(p.grad is not None and p.grad.data_ptr() == g.data_ptr() for p, g in zip(ctx.params, grads)):
            # a parameter's .grad IS its slice of the flat buffer (zero_grad(set_to_none=False), or a second
            # backward): autograd is about to ADD what we return to it, so the result must live elsewhere
            flat = torch.empty_like(flat)
            views, off = [], 0
            for g in grads:
                views.append(flat[off:off + g.numel()].view(g.shape))
                off += g.numel()
            grads = views
        plan.backward(dout, ctx.params, grads, flat_grad=flat)
        return (None, None) + tuple(grads)


def _pointwise_desc(x, B, shape3, cin_p, n):
    """clx_conv_desc of a 1x1(x1) convolution over a pixel-major (M, cin_p) tensor."""
    d = ClxConvDesc()
    d.nsrc = 1
    src = ClxSrc()
    src.ptr = x.data_ptr()
    src.C = cin_p
    src.ld = cin_p
    src.D, src.H, src.W = shape3
    src.oz = src.oy = src.ox = 0
    src.fz = src.fy = src.fx = 1
    d.src[0] = src
    d.B = B
    d.ID, d.IH, d.IW = shape3
    d.KD = d.KH = d.KW = 1
    d.PD = d.PH = d.PW = 0
    d.N = n
    return d


class _HeadFunction(torch.autograd.Function):
    """``self.head(backbone_output)`` (unet.py:52-67: Conv1x1 -> ReLU -> Conv1x1) on the
    libclx convolution entry points: forward = two clx_conv_fwd launches, backward = their
    data-gradient form (ReLU gate fused) + two clx_conv_wgrad launches."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1):
        _clx.require_device(x, "backbone_output")
        dev = x.device
        st = _clx.stream_ptr(dev)
        B, C = x.shape[0], x.shape[1]
        spatial = tuple(x.shape[2:])
        shape3 = (1,) + spatial if len(spatial) == 2 else spatial
        n = shape3[0] * shape3[1] * shape3[2]
        F, Dout = w0.shape[0], w1.shape[0]
        Cp, Fp, Dp = pad4(C), pad4(F), pad4(Dout)
        xp = torch.empty((B * n, Cp), dtype=torch.float32, device=dev)
        _clx.call("clx_planar_to_pixel", _clx.ptr(x.contiguous()), _clx.ptr(xp), B, C, n, Cp, st)
        h0 = _clx.zeros((B * n, Fp), torch.float32, dev)
        h1 = _clx.zeros((B * n, Dp), torch.float32, dev)
        packs = []
        for w, cin, cin_p, cout, cout_p, src, dst, bias, relu in (
                (w0, C, Cp, F, Fp, xp, h0, b0, 1), (w1, F, Fp, Dout, Dp, h0, h1, b1, 0)):
            wv = w.detach().reshape(cout, cin, 1).contiguous()
            wp = torch.empty(cout_p * cin_p, dtype=torch.float32, device=dev)
            _clx.call("clx_pack_weights", _clx.ptr(wv), _clx.ptr(wp), cout, cin, 1, cin_p, cout_p, 0, st)
            d = _pointwise_desc(src, B, shape3, cin_p, cout)
            d.wpack = wp.data_ptr()
            d.bias = bias.data_ptr() if bias is not None else None
            d.relu = relu
            d.out = dst.data_ptr()
            d.ld_out = cout_p
            _clx.call("clx_conv_fwd", ctypes.byref(d), st)
            packs.append((wv, wp))
        out = torch.empty((B, Dout) + spatial, dtype=torch.float32, device=dev)
        _clx.call("clx_pixel_to_planar", _clx.ptr(h1), _clx.ptr(out), B, Dout, n, Dp, st)
        ctx.save_for_backward(xp, h0, w0, w1)
        ctx.geom = (B, C, F, Dout, shape3, spatial, b0 is not None, b1 is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        xp, h0, w0, w1 = ctx.saved_tensors
        B, C, F, Dout, shape3, spatial, has_b0, has_b1 = ctx.geom
        dev = dout.device
        st = _clx.stream_ptr(dev)
        n = shape3[0] * shape3[1] * shape3[2]
        Cp, Fp, Dp = pad4(C), pad4(F), pad4(Dout)
        dh1 = torch.empty((B * n, Dp), dtype=torch.float32, device=dev)
        _clx.call("clx_planar_to_pixel", _clx.ptr(dout.contiguous()), _clx.ptr(dh1), B, Dout, n, Dp, st)
        dh0 = _clx.zeros((B * n, Fp), torch.float32, dev)
        dxp = _clx.zeros((B * n, Cp), torch.float32, dev)
        grads = {}
        for tag, w, cin, cin_p, cout, cout_p, src, dy, gate, dsrc, has_b in (
                ("1", w1, F, Fp, Dout, Dp, h0, dh1, h0, dh0, has_b1),
                ("0", w0, C, Cp, F, Fp, xp, dh0, None, dxp, has_b0)):
            dwp = _clx.zeros(cout_p * cin_p, torch.float32, dev)
            gb = _clx.zeros(cout, torch.float32, dev) if has_b else None
            d = _pointwise_desc(src, B, shape3, cin_p, cout_p)
            _clx.call("clx_conv_wgrad", ctypes.byref(d), _clx.ptr(dy), cout_p, _clx.ptr(dwp), _clx.ptr(gb), st)
            gw = torch.empty_like(w)
            _clx.call("clx_unpack_wgrad", _clx.ptr(dwp), _clx.ptr(gw), cout, cin, 1, cout_p, cin_p, st)
            grads[tag] = (gw, gb)
            if tag == "0" and not ctx.needs_input_grad[0]:
                continue
            wv = w.detach().reshape(cout, cin, 1).contiguous()
            wpd = torch.empty(cin_p * cout_p, dtype=torch.float32, device=dev)
            _clx.call("clx_pack_weights", _clx.ptr(wv), _clx.ptr(wpd), cout, cin, 1, cin_p, cout_p, 1, st)
            dd = _pointwise_desc(dy, B, shape3, cout_p, cin_p)
            dd.wpack = wpd.data_ptr()
            dd.bias = None
            dd.relu = 0
            dd.mask = gate.data_ptr() if gate is not None else None
            dd.ld_mask = cin_p if gate is not None else 0
            dd.out = dsrc.data_ptr()
            dd.ld_out = cin_p
            _clx.call("clx_conv_fwd", ctypes.byref(dd), st)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B, C) + spatial, dtype=torch.float32, device=dev)
            _clx.call("clx_pixel_to_planar", _clx.ptr(dxp), _clx.ptr(dx), B, C, n, Cp, st)
        return dx, grads["0"][0], grads["0"][1], grads["1"][0], grads["1"][1]


class UNetModel(nn.Module):  # type: ignore
    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        num_fmaps: int,
        fmap_inc_factor: int,
        features_in_last_layer: int,
        downsampling_factors: List[Tuple[int, ...]],
        num_spatial_dims: int,
    ):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.features_in_last_layer = features_in_last_layer
        self.num_fmaps = num_fmaps
        self.fmap_inc_factor = fmap_inc_factor
        self.downsampling_factors = [tuple(f) for f in downsampling_factors]
        self.num_spatial_dims = num_spatial_dims
        self.mode = "train"
        self.backbone = _Backbone(in_channels, num_fmaps, fmap_inc_factor,
                                  self.downsampling_factors, features_in_last_layer, num_spatial_dims)
        conv = nn.Conv2d if num_spatial_dims == 2 else nn.Conv3d
        self.head = torch.nn.Sequential(
            conv(self.features_in_last_layer, self.features_in_last_layer, 1),
            nn.ReLU(),
            conv(self.features_in_last_layer, out_channels, 1),
        )
        self._plans = {}
        self._flat = None
        self._flat_grad = None
        self._weights_epoch = 0
        self.max_infer_batch = None        # copies per forward of the infer-mode noise loop; None: by tile size (infer_chunk)

    # ------------------------------------------------------------ plumbing
    def _ordered_params(self):
        """[w0, b0, w1, b1, ...] in the plan's layer order."""
        mods = []
        for cp in self.backbone.l_conv:
            mods += [m for m in cp.conv_pass if isinstance(m, nn.modules.conv._ConvNd)]
        for cp in reversed(list(self.backbone.r_conv[0])):
            mods += [m for m in cp.conv_pass if isinstance(m, nn.modules.conv._ConvNd)]
        mods += [self.head[0], self.head[2]]
        out = []
        for m in mods:
            out += [m.weight, m.bias]
        return out

    def _param_version(self):
        return (self._weights_epoch,) + tuple(p._version for p in self._ordered_params())

    def mark_weights_changed(self):
        """Called by optimizers that update the parameters behind torch's back."""
        self._weights_epoch += 1

    def flatten_parameters(self):
        """Makes every parameter (and .grad) a view of one flat f32 buffer so the optimizer
        step and the data-parallel all-reduce are single launches over contiguous memory."""
        params = self._ordered_params()
        dev = params[0].device
        total = sum(p.numel() for p in params)
        ok = self._flat is not None and self._flat.device == dev and self._flat.numel() == total
        if ok:
            off = 0
            for p in params:
                if p.data_ptr() != self._flat.data_ptr() + 4 * off:
                    ok = False
                    break
                off += p.numel()
        if not ok:
            flat = torch.empty(total, dtype=torch.float32, device=dev)
            off = 0
            for p in params:
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
                off += n
            self._flat = flat
            self._flat_grad = _clx.zeros(total, torch.float32, dev)
        return self._flat, self._flat_grad

    def _grad_views(self):
        _, fg = self.flatten_parameters()
        views, off = [], 0
        for p in self._ordered_params():
            n = p.numel()
            views.append(fg[off:off + n].view(p.shape))
            off += n
        return views

    def attach_flat_grads(self):
        """p.grad := view into the flat gradient buffer (used by the fused train step)."""
        views = self._grad_views()
        for p, g in zip(self._ordered_params(), views):
            p.grad = g
        return views

    def _plan_for(self, raw, keep):
        _clx.require_device(raw, "raw")
        if raw.dtype != torch.float32:
            raise TypeError(f"raw must be float32, got {raw.dtype}")
        if raw.ndim != self.num_spatial_dims + 2 or raw.shape[1] != self.in_channels:
            raise ValueError(
                f"raw must have shape (B, {self.in_channels}, *{self.num_spatial_dims} spatial dims), "
                f"got {tuple(raw.shape)}")
        w = self.head[0].weight
        if w.device != raw.device:
            raise RuntimeError(f"model parameters are on {w.device} but raw is on {raw.device}")
        key = (tuple(raw.shape), raw.device, bool(keep))
        plan = self._plans.get(key)
        if plan is None:
            topo = build_topology(self.in_channels, self.out_channels, self.num_fmaps,
                                  self.fmap_inc_factor, self.features_in_last_layer,
                                  self.downsampling_factors, self.num_spatial_dims, raw.shape[2:])
            # one plan (activation arena) per input shape; drop older ones to bound memory
            for k in [k for k in self._plans if k[2] == bool(keep)]:
                del self._plans[k]
            if not keep:
                # (the further plans of the chunk streams and the one-image plan of the changed-rows form belong to the shape
                #  that goes: they must not stay beside the new arena until _forward_chunks replaces them)
                self._infer_pair = None
                self._clean_plan = None
            if dual_stream_wanted(topo, raw.shape[0], keep):
                plan = DualPlan(topo, raw.shape[0], raw.device, keep)
            else:
                plan = UNetPlan(topo, raw.shape[0], raw.device, keep)
            self._plans[key] = plan
        return plan

    def _apply(self, fn, *args, **kwargs):
        # .to()/.cuda() re-allocates parameters: forget flat views and plans
        self._plans = {}
        self._infer_pair = None
        self._clean_plan = None
        self._flat = None
        self._flat_grad = None
        return super()._apply(fn, *args, **kwargs)

    # ------------------------------------------------------------- forward
    def head_forward(self, backbone_output):
        """``self.head(backbone_output)`` (unet.py:65-67) for callers that hold a backbone
        output of their own; ``forward`` itself runs backbone + head as one launch plan."""
        if backbone_output.ndim != self.num_spatial_dims + 2 or \
                backbone_output.shape[1] != self.features_in_last_layer:
            raise ValueError(
                f"backbone_output must be (B, {self.features_in_last_layer}, *{self.num_spatial_dims} "
                f"spatial dims), got {tuple(backbone_output.shape)}")
        if backbone_output.dtype != torch.float32:
            raise TypeError(f"backbone_output must be float32, got {backbone_output.dtype}")
        return _HeadFunction.apply(backbone_output, self.head[0].weight, self.head[0].bias,
                                   self.head[2].weight, self.head[2].bias)

    def _forward_nograd(self, raw):
        params = self._ordered_params()
        plan = self._plan_for(raw, keep=False)
        plan.pack_weights(params, self._param_version(), need_dgrad=False)
        return plan.forward(raw, params)

    def _sparse_prepare(self, plan, noisy, clean, step, params):
        """Noisy copies of ONE image (`clean`, (1, C, *spatial)): per chunk of `step` copies the rows of the first
        convolution's output that differ from the clean image's, and the clean image's rows behind the 1x1 layers that
        follow (plan.pointwise_prefix) — and, where a 2-D Winograd layer reads those, its output tiles whose input window
        holds such a row, with the clean image's output of that layer — -> a list of per-chunk dicts for UNetPlan.forward
        (sparse=...), or None where the dense path is the better one: no such prefix, CLX_SPARSE_NOISE=0, or more than
        CLX_SPARSE_NOISE_MAX (default 0.3) of the rows changed.  One host synchronisation (the counts size the launches);
        DESIGN.md 3.1f."""
        if clean is None or os.environ.get("CLX_SPARSE_NOISE", "1") == "0":
            return None
        prefix = plan.pointwise_prefix()
        if prefix is None:
            return None
        first, tail = prefix
        T = noisy.shape[0]
        if T % step or T < 2 or tuple(clean.shape[1:]) != tuple(noisy.shape[1:]) or clean.shape[0] != 1:
            return None
        npix_out = first.out_shape[0] * first.out_shape[1] * first.out_shape[2]
        nchunks = T // step
        cap = max(1, int(float(os.environ.get("CLX_SPARSE_NOISE_MAX", "0.3")) * step * npix_out))
        dev = noisy.device
        if getattr(self, "_clean_plan", None) is None or self._clean_plan[0] is not plan:
            # room for the one-image plan and the compact rows (per stream) beside everything else, with a margin
            one_image = sum(t.numel() * t.element_size() for t in plan.buf.values()) // max(step, 1)
            compact = 3 * cap * 4 * max(pad4(op.cout) for op in [first] + tail)
            if parallel.free_device_memory(dev) < 1.5 * (one_image + 2 * compact):
                return None
        rows = torch.empty(nchunks * cap, dtype=torch.int32, device=dev)
        counts = torch.empty(2 * nchunks, dtype=torch.int32, device=dev)       # changed rows, changed tiles per chunk
        # the Winograd layer behind the 1x1 layers: its output tiles whose input window holds a changed row
        tiled = plan.tiled_layer_behind_prefix() if os.environ.get("CLX_SPARSE_TILES", "1") != "0" else None
        tiles = None
        if tiled is not None:
            top, tile = tiled
            th, tw = -(-top.out_shape[1] // tile), -(-top.out_shape[2] // tile)
            cap_t = max(1, int(float(os.environ.get("CLX_SPARSE_TILES_MAX", "0.7")) * step * th * tw))
            tiles = torch.empty(nchunks * cap_t, dtype=torch.int32, device=dev)
        noisy = noisy.contiguous()
        clean = clean.contiguous()
        st = _clx.stream_ptr(dev)
        ws = torch.empty(int(_clx.load().clx_changed_rows_workspace(T, *first.in_shape)), dtype=torch.uint8, device=dev)
        _clx.call("clx_changed_rows", _clx.ptr(clean), _clx.ptr(noisy), T, noisy.shape[1], *first.in_shape,
                  *first.kernel, step, _clx.ptr(rows), _clx.ptr(counts), cap, _clx.ptr(ws), st)
        # the counts size the launches: they leave for pinned memory now and the host waits for THAT copy only, after
        # it has enqueued the clean image's pass (the device works on it meanwhile)
        host = getattr(self, "_counts_host", None)
        if host is None or host.numel() < 2 * nchunks:
            host = self._counts_host = torch.empty(max(2 * nchunks, 64), dtype=torch.int32).pin_memory()
        if tiles is not None:
            _clx.call("clx_changed_tiles", _clx.ptr(ws), T, *first.in_shape, *first.kernel, top.kernel[1], top.kernel[2],
                      tile, step, _clx.ptr(tiles), _clx.ptr(counts[nchunks:]), cap_t, st)
        host[:2 * nchunks].copy_(counts, non_blocking=True)
        copied = torch.cuda.Event()
        copied.record()
        # the clean image through the prefix, on a one-image plan that reads the chunk plan's packed weights
        cp = getattr(self, "_clean_plan", None)
        if cp is None or cp[0] is not plan:
            one = UNetPlan(plan.topo, 1, dev, False)
            # (the one-image plan must run these layers the way the chunk plan does: it reads that plan's packed weights)
            same = all(one.algo[op.name] == plan.algo[op.name] for op in [first] + tail)
            same_tiles = tiled is not None and one.algo[tiled[0].name] == plan.algo[tiled[0].name] and \
                bool(one.fused_pool.get(tiled[0].name)) == bool(plan.fused_pool.get(tiled[0].name))
            cp = self._clean_plan = (plan, one if same else None, same_tiles)
        one = cp[1]
        if one is None:
            return None
        if tiles is not None and not cp[2]:
            tiles = None
        one.wpack_fwd = plan.wpack_fwd
        one._wplanes = plan._wplanes
        clean_rows = one.forward_prefix(clean, params, len(tail) + (1 if tiles is not None else 0))
        clean_tile_rows = clean_pool_rows = None
        if tiles is not None:
            clean_tile_rows, clean_rows = clean_rows, one.buf[tail[-1].out]
            if one.fused_pool.get(top.name) is not None:
                clean_pool_rows = one.buf[one.fused_pool[top.name].out]
        copied.synchronize()
        n = host[:nchunks].tolist()
        nt = host[nchunks:2 * nchunks].tolist() if tiles is not None else None
        if nt is not None and max(nt) > cap_t:
            tiles = None                                           # most tiles changed: that layer densely
        self._last_changed_rows = dict(fraction=sum(n) / float(T * npix_out), layers=[op.name for op in tail],
                                       window=tuple(first.kernel), used=max(n) <= cap)
        if max(n) > cap:
            return None
        out = [dict(clean_rows=clean_rows, rows=rows[j * cap:j * cap + n[j]], n=n[j]) for j in range(nchunks)]
        if tiles is not None:
            self._last_changed_rows.update(tile_layer=top.name, tile_fraction=sum(nt) / float(T * th * tw))
            for j, dct in enumerate(out):
                dct.update(tile_op=top, tiles=tiles[j * cap_t:j * cap_t + nt[j]], ntiles=nt[j],
                           clean_tile_rows=clean_tile_rows, clean_pool_rows=clean_pool_rows)
        return out

    def _forward_chunks(self, noisy, step, clean=None):
        """The network on `noisy` (T, C, *spatial) in chunks of `step` copies -> (T, out_channels, *out_spatial).
        clean: the image the copies are noisy versions of, if they are (infer_on_device): the 1x1 layers behind the first
        convolution then run once on the clean image and on the changed rows of the copies (_sparse_prepare; the same
        bits, tests/test_gpu_unet.py).
        With two or more whole chunks of a size that fills the device (>= CLX_STREAMS_MIN_GFLOP of forward pass each, default
        100) and room for a second set of activations, the chunks alternate between two plans on two streams that share one
        set of packed weights: the HBM-bound Winograd transforms of one chunk run under the GEMMs of the other, as in
        plan.DualPlan.  Same kernels on the same rows: bit-identical to the plain loop (tests/test_gpu_unet.py).  Measured at
        the benchmark tile (8 copies of 528 x 528 per chunk): embedding stage 199.7 -> 189.5 ms (round 4, with the second
        stream started behind the first chunk's second layer; 204 -> 198 in round 3's form); with the first level on the
        changed rows / tiles of the copies (`clean`): 171 -> 163 ms.  FOUR streams (all chunks of a tile in flight) bring the
        stage to 158-159 ms (three: 162-166) but infer() end to end from 165.3 to 168.8 ms per sample — the post-processing of
        the previous sample shares the device with four busy queues instead of two — so two stay the default
        (CLX_INFER_STREAMS=4 for the kernel-level figure).  CLX_INFER_STREAMS=1: the plain loop."""
        T = noisy.shape[0]
        nstreams = min(int(os.environ.get("CLX_INFER_STREAMS", "2") or 2), 4, T // max(step, 1))
        if T % step:
            preds = [self._forward_nograd(noisy[i:i + step].contiguous()) for i in range(0, T, step)]
            return torch.cat(preds, dim=0) if len(preds) > 1 else preds[0]
        first = noisy[:step].contiguous()
        plan = self._plan_for(first, keep=False)
        params = self._ordered_params()
        plan.pack_weights(params, self._param_version(), need_dgrad=False)
        big = forward_flops(plan.topo, step) >= float(os.environ.get("CLX_STREAMS_MIN_GFLOP", "100")) * 1e9
        sparse = self._sparse_prepare(plan, noisy, clean, step, params) if big else None

        def plain_loop():
            preds = [plan.forward(noisy[i:i + step].contiguous(), params, sparse=sparse[j] if sparse else None)
                     for j, i in enumerate(range(0, T, step))]
            return torch.cat(preds, dim=0) if len(preds) > 1 else preds[0]

        if nstreams < 2 or not big:
            return plain_loop()
        pair = getattr(self, "_infer_pair", None)
        if pair is not None and pair[0] is plan and len(pair[2]) <= nstreams:
            nstreams = len(pair[2])                      # (built when less memory was free: keep it)
        if pair is None or pair[0] is not plan:
            # every further set of activations must fit beside the first (several ranks may share a device in a rehearsal):
            # as many streams as there is room for
            need = plan.arena_bytes()
            free = parallel.free_device_memory(noisy.device)
            nstreams = min(nstreams, 1 + int(free // (1.25 * need)))
            if nstreams < 2:
                return plain_loop()
            self._infer_pair = None
            others = [UNetPlan(plan.topo, step, noisy.device, False) for _ in range(nstreams - 1)]
            pair = self._infer_pair = (plan, others[0], [torch.cuda.Stream(device=noisy.device) for _ in range(nstreams)],
                                       others)
        _plan, _other, streams, others = pair
        plans = [plan] + list(others)
        for o in others:
            o.share_forward_from(plan)
        t = plan.topo
        preds = torch.empty((T, t.out_channels) + tuple(t.out_shape[3 - t.nd:]), dtype=torch.float32,
                            device=noisy.device)
        main = torch.cuda.current_stream(noisy.device)
        for s in streams:
            s.wait_stream(main)
        # the second stream starts BEHIND the first chunk's second layer (operation 1 of the forward order): two streams
        # that run the same launch sequence from the same instant stay in phase — transforms beside transforms, GEMMs beside
        # GEMMs — and overlap little; out of phase, one stream's HBM-bound transforms fall under the other's GEMMs.
        # Measured at the benchmark tile (ms per tile, two runs): no offset 192.7 / 196.2; behind operation 0: 200.0 /
        # 192.9; 1: 189.1 / 189.9; 2: 192.2 / 193.0; 3: 191.1 / 192.0; 4-8: 191.5-199.6.  CLX_INFER_OFFSET_OP=-1: none
        offset_op = int(os.environ.get("CLX_INFER_OFFSET_OP", "1"))
        # (with more than two streams: each behind its predecessor)
        started = [torch.cuda.Event() for _ in range(nstreams - 1)] if offset_op >= 0 else None

        for j, i in enumerate(range(0, T, step)):
            with torch.cuda.stream(streams[j % nstreams]):
                on_op = None
                if started is not None and j < nstreams:
                    if j > 0:
                        streams[j].wait_event(started[j - 1])
                    if j < nstreams - 1:
                        on_op = (lambda k, j=j: started[j].record(streams[j]) if k == offset_op else None)
                plans[j % nstreams].forward(noisy[i:i + step], params, out=preds[i:i + step], on_op=on_op,
                                            sparse=sparse[j] if sparse else None)
        for s in streams:
            main.wait_stream(s)
        return preds

    def forward(self, raw):
        if self.mode == "train":
            params = self._ordered_params()
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                self.flatten_parameters()
                params = self._ordered_params()
                return _UNetFunction.apply(self, raw, *params)
            return self._forward_nograd(raw)
        elif self.mode == "infer":
            return self.infer_on_device(raw).cpu()

    @torch.no_grad()
    def infer_on_device(self, raw, noise=None, std_minmax=None):
        """Infer-mode forward (unet.py:73-100) without the final D2H copy.

        For every sample: 2 * num_infer_iterations salt/pepper-noised copies
        (value 0.5 then 1.0) -> forward -> per-pixel mean and population std over
        the copies; std summed over channels.  Returns (B, D+1, *out) on device.
        `noise`: optional (B, 2*N, C, *spatial) uniform randoms (CPU or device);
        default draws them with torch.rand on the CPU exactly like the reference
        (one call per noisy copy, in the reference's order).
        `std_minmax`: optional (float32[2] device tensor, reset) — the running minimum / maximum of the std channel
        over the calls of one image's tiles (clx_noise_stats_minmax; reset = True on the first tile): the range the
        Otsu histogram of cellulus/detect.py:88-91 needs, without another pass over the channel."""
        _clx.require_device(raw, "raw")
        n_it = int(self.num_infer_iterations)
        T = 2 * n_it
        B = raw.shape[0]
        st = _clx.stream_ptr(raw.device)
        embeddings = []
        for sample in range(B):
            raw_sample = raw[sample:sample + 1]
            if noise is None:
                rnd = torch.stack([torch.rand(*raw_sample.shape) for _ in range(T)], dim=0)
                rnd = rnd.reshape((T,) + tuple(raw_sample.shape[1:]))
            else:
                rnd = noise[sample]
            rnd = rnd.to(raw.device, non_blocking=True)
            noisy = self._inject_noise(rnd, raw_sample, n_it)
            step = self.infer_chunk(T, raw_sample.shape[2:])
            preds = self._forward_chunks(noisy, step, clean=raw_sample)
            C = preds.shape[1]
            n = preds[0, 0].numel()
            out = torch.empty((C + 1,) + tuple(preds.shape[2:]), dtype=torch.float32, device=raw.device)
            if std_minmax is None:
                _clx.call("clx_noise_stats", _clx.ptr(preds), _clx.ptr(out), T, C, n, st)
            else:
                mm, reset = std_minmax
                _clx.call("clx_noise_stats_minmax", _clx.ptr(preds), _clx.ptr(out), T, C, n, _clx.ptr(mm),
                          1 if (reset and sample == 0) else 0, st)
            embeddings.append(out)
        return torch.stack(embeddings, dim=0)

    def _inject_noise(self, rnd, raw_sample, n_it):
        """The 2 * n_it noisy copies of one sample (unet.py:75-88: `noisy[rnd <= p] = 0.5` for the first n_it draws, 1.0
        for the rest) in one launch (clx_noise_inject); rnd: (T, C, *spatial) float32 on the device."""
        if rnd.dtype != torch.float32 or raw_sample.dtype != torch.float32 or not rnd.is_contiguous():
            vals = torch.tensor([0.5] * n_it + [1.0] * n_it, dtype=raw_sample.dtype, device=raw_sample.device)
            return torch.where(rnd <= self.p_salt_pepper, vals.view((-1,) + (1,) * (raw_sample.ndim - 1)),
                               raw_sample.expand_as(rnd))
        noisy = torch.empty_like(rnd)
        clean = raw_sample.contiguous()
        _clx.call("clx_noise_inject", _clx.ptr(rnd), _clx.ptr(clean), _clx.ptr(noisy), rnd.shape[0], n_it,
                  clean.numel(), float(self.p_salt_pepper), _clx.stream_ptr(rnd.device))
        return noisy

    def infer_chunk(self, T, spatial):
        """Noisy copies per forward of the infer-mode loop (unet.py:75-88 runs them one by one).  ``max_infer_batch`` if
        set; else as many as give the chunk the pixel count of eight 528^2 tiles (2.2 M input pixels: every launch fills
        the device and the batched Winograd operands stay inside 32-bit offsets), a divisor of T: 8 copies of a 528^2 tile, 16
        of a 272^2 one (two chunks for the two streams), all of them for small tiles."""
        if self.max_infer_batch is not None:
            return max(1, min(T, int(self.max_infer_batch)))
        npix = 1
        for s in spatial:
            npix *= int(s)
        want = max(1, (8 * 528 * 528) // max(npix, 1))
        if parallel.ranks_sharing_device() > 1 and torch.cuda.is_available():
            # several ranks on one device (the rehearsal hook CLX_LOCAL_DEVICE): a chunk's plan must fit the rank's share twice
            # (two streams) beside what the rank holds already.  A plan takes ~60 bytes per input pixel and feature map of
            # the first level (activations, the Winograd operands as planes, their products: 34 GB for eight 528^2 copies
            # at 256 feature maps)
            dev = getattr(self, "device", None) or torch.device("cuda", torch.cuda.current_device())
            room = parallel.free_device_memory(dev)
            want = max(1, min(want, int(room / (2.5 * 60.0 * self.num_fmaps * max(npix, 1)))))
        if want >= T:
            return T                   # small tiles: all copies in one forward
        best = 1
        for step in range(1, T + 1):
            if T % step == 0 and step <= want:
                best = step
        return best

    def set_infer(self, p_salt_pepper, num_infer_iterations, device):
        self.mode = "infer"
        self.p_salt_pepper = p_salt_pepper
        self.num_infer_iterations = num_infer_iterations
        self.device: torch.device = device

    @staticmethod
    def select_and_add_coordinates(outputs, coordinates):
        """sel[b, p, c] = outputs[b, c, (z,) y, x] + coordinates[b, p, c]  (unet.py:108-124);
        coordinate column 0 is x (last axis)."""
        from ..criterions.oce_loss import gather_add

        return gather_add(outputs, coordinates)
