"""CLX_ALGO_WINOGRAD4_FUSED (csrc/wino_fused.hip) through the C ABI: the one-launch F(4x4, 3x3) / F(4x4, 2x2) forward —
nn.Conv2d 3x3 + ReLU (+ MaxPool2d) of funlib's ConvPass, cellulus/models/unet.py:24-51 — against float64 torch
convolutions on the CPU and against the three-launch form (CLX_ALGO_WINOGRAD4), on extents that are not multiples of
the tile or of the block, with every epilogue the kernel knows: bias + ReLU, accumulate, gate bits, fused 2 x 2
pooling, tile lists, cropped sources with a pixel stride above the channel count."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5          # of the output range: F(4x4) in float32 (tests/test_gpu_unet.py holds the three-launch form to the same)


def _desc(x_d, B, SH, SW, H, W, C, ld, k, N, oy=0, ox=0):
    from cellulus_amd._clx import ClxConvDesc, ClxSrc

    d = ClxConvDesc()
    d.nsrc = 1
    s = ClxSrc()
    s.ptr, s.C, s.ld = x_d.data_ptr(), C, ld
    s.D, s.H, s.W = 1, SH, SW
    s.oz, s.oy, s.ox = 0, oy, ox
    s.fz = s.fy = s.fx = 1
    d.src[0] = s
    d.B = B
    d.ID, d.IH, d.IW = 1, H, W
    d.KD, d.KH, d.KW = 1, k, k
    d.PD = d.PH = d.PW = 0
    d.N = N
    return d


@pytest.mark.parametrize("form", ["one_launch", "two_launches"])
@pytest.mark.parametrize("k,B,H,W,C,N", [(3, 2, 23, 30, 16, 64), (3, 1, 41, 38, 72, 128), (2, 2, 22, 27, 24, 64),
                                         (3, 3, 10, 150, 8, 192), (2, 1, 35, 34, 40, 128)])
def test_fused_winograd_vs_f64_convolution(k, B, H, W, C, N, form, device):
    """form: layers with N > 64 given the scratch clx_conv_fused_workspace_bytes asks for run as input transform +
    product kernel (both operands as ready-made fragments from global memory); without it, and always at N = 64, as
    one launch."""
    from cellulus_amd import _clx

    if form == "two_launches" and N <= 64:
        pytest.skip("N = 64 has one form only")

    torch.manual_seed(1000 * k + H + W + C)
    OH, OW = H - k + 1, W - k + 1
    # the logical input is a crop of a larger stored grid whose pixels carry 8 more floats than the layer reads
    oy, ox, SH, SW, ld = 2, 1, H + 3, W + 2, C + 8
    stored = torch.randn(B, SH, SW, ld)
    x = stored[:, oy:oy + H, ox:ox + W, :C]
    w = torch.randn(N, C, k, k) * 0.2
    bias = torch.randn(N)
    prev = torch.randn(B, OH, OW, N)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double()).permute(0, 2, 3, 1)
    st = _clx.stream_ptr(device)
    lib = _clx.load()
    x_d = stored.to(device).contiguous()
    w_d = w.reshape(N, C, k * k).to(device).contiguous()
    nxi = (4 + k - 1) ** 2
    wf = torch.empty(nxi * N * C, device=device)
    _clx.call("clx_pack_weights", _clx.ptr(w_d), _clx.ptr(wf), N, C, k * k, C, N, 7, st)
    # the fused layout is a permutation of the plain F(4x4) pack
    wp = torch.empty(nxi * N * C, device=device)
    _clx.call("clx_pack_weights", _clx.ptr(w_d), _clx.ptr(wp), N, C, k * k, C, N, 4, st)
    assert torch.equal(torch.sort(wf)[0], torch.sort(wp)[0])

    d0 = _desc(x_d, B, SH, SW, H, W, C, ld, k, N, oy, ox)
    need3 = int(lib.clx_conv_fused_workspace_bytes(ctypes.byref(d0)))
    assert (need3 > 0) == (N > 64)
    ws3 = torch.full((need3 // 4 + 4,), float("nan"), device=device) if form == "two_launches" else None

    def desc():
        d = _desc(x_d, B, SH, SW, H, W, C, ld, k, N, oy, ox)
        d.algo = 3
        d.wpack = wf.data_ptr()
        if ws3 is not None:
            d.workspace, d.workspace_bytes = ws3.data_ptr(), need3
        return d

    assert int(lib.clx_conv_fused_applicable(ctypes.byref(desc()))) == 1
    scale = max(1.0, ref.abs().max().item())
    outs = {}
    for mode in ("plain", "bias_relu", "accumulate"):
        d = desc()
        out = prev.to(device).clone().contiguous() if mode == "accumulate" else torch.full((B, OH, OW, N), float("nan"), device=device)
        d.out, d.ld_out = out.data_ptr(), N
        want = ref
        if mode == "bias_relu":
            b_d = bias.to(device)
            d.bias, d.relu = b_d.data_ptr(), 1
            gate = torch.full((B, OH, OW, N // 32), -1, dtype=torch.int32, device=device)
            d.gate_out, d.ld_gate = gate.data_ptr(), N // 32
            want = torch.relu(ref + bias.double())
        elif mode == "accumulate":
            b_d = bias.to(device)
            d.bias, d.relu, d.accumulate = b_d.data_ptr(), 1, 1
            want = torch.relu(ref + bias.double() + prev.double())
        _clx.call("clx_conv_fwd", ctypes.byref(d), st)
        err = (out.cpu().double() - want).abs().max().item()
        assert err < TOL * scale, (mode, err)
        outs[mode] = out
        if mode == "bias_relu":       # bit (n & 31) of word n >> 5 = (out > 0)
            bits = (gate.cpu().long().unsqueeze(-1) >> torch.arange(32)) & 1
            assert torch.equal(bits.reshape(B, OH, OW, N).bool(), out.cpu() > 0)
    # against the three-launch form: the same algorithm, other summation orders
    d = _desc(x_d, B, SH, SW, H, W, C, ld, k, N, oy, ox)
    d.algo = 2
    need = int(lib.clx_conv_workspace_bytes(ctypes.byref(d), 0))
    ws = torch.empty(need // 4 + 4, device=device)
    d.wpack, d.workspace, d.workspace_bytes = wp.data_ptr(), ws.data_ptr(), ws.numel() * 4
    out3 = torch.empty(B, OH, OW, N, device=device)
    d.out, d.ld_out = out3.data_ptr(), N
    _clx.call("clx_conv_fwd", ctypes.byref(d), st)
    assert (out3 - outs["plain"]).abs().max().item() < TOL * scale

    # ---- a tile list: the listed tiles get the dense launch's bits, everything else stays as it was
    th, tw = -(-OH // 4), -(-OW // 4)
    g = torch.Generator().manual_seed(5)
    pick = torch.nonzero(torch.rand(B * th * tw, generator=g) < 0.4).flatten().to(torch.int32)
    pick = pick[torch.randperm(pick.numel(), generator=g)]          # any order
    assert 0 < pick.numel() < B * th * tw
    d = desc()
    b_d = bias.to(device)
    d.bias, d.relu = b_d.data_ptr(), 1
    out = torch.full((B, OH, OW, N), -7.0, device=device)
    d.out, d.ld_out = out.data_ptr(), N
    pick_d = pick.to(device)
    d.tile_list, d.tile_count = pick_d.data_ptr(), pick.numel()
    _clx.call("clx_conv_fwd", ctypes.byref(d), st)
    listed = torch.zeros(B, th, tw, dtype=torch.bool)
    listed.view(-1)[pick.long()] = True
    px = listed.repeat_interleave(4, 1).repeat_interleave(4, 2)[:, :OH, :OW]
    got, dense = out.cpu(), outs["bias_relu"].cpu()
    assert torch.equal(got[px], dense[px])
    assert (got[~px] == -7.0).all()
    d.tile_count = 0                                                # an empty list is a no-op
    _clx.call("clx_conv_fwd", ctypes.byref(d), st)
    assert torch.equal(out.cpu(), got)


@pytest.mark.parametrize("with_list", [False, True])
def test_fused_winograd_pooling_epilogue(with_list, device):
    """pool_out: the 2 x 2 max-pooled output (funlib's Downsample behind a level's last convolution) from the same
    launch — the values clx_maxpool_fwd gives on the launch's own output."""
    from cellulus_amd import _clx

    torch.manual_seed(3)
    B, H, W, C, N = 2, 24, 30, 16, 64
    OH, OW = H - 2, W - 2
    x = torch.randn(B, H, W, C)
    w = torch.randn(N, C, 3, 3) * 0.2
    bias = torch.randn(N)
    st = _clx.stream_ptr(device)
    x_d, b_d = x.to(device), bias.to(device)
    w_d = w.reshape(N, C, 9).to(device).contiguous()
    wf = torch.empty(36 * N * C, device=device)
    _clx.call("clx_pack_weights", _clx.ptr(w_d), _clx.ptr(wf), N, C, 9, C, N, 7, st)
    d = _desc(x_d, B, H, W, H, W, C, C, 3, N)
    d.algo, d.wpack, d.bias, d.relu = 3, wf.data_ptr(), b_d.data_ptr(), 1
    out = torch.full((B, OH, OW, N), -7.0, device=device)
    pooled = torch.full((B, OH // 2, OW // 2, N), -7.0, device=device)
    d.out, d.ld_out, d.pool_out, d.ld_pool = out.data_ptr(), N, pooled.data_ptr(), N
    th, tw = -(-OH // 4), -(-OW // 4)
    listed = torch.ones(B, th, tw, dtype=torch.bool)
    if with_list:
        listed = torch.rand(B, th, tw, generator=torch.Generator().manual_seed(1)) < 0.5
        pick = torch.nonzero(listed.flatten()).flatten().to(torch.int32).to(device)
        d.tile_list, d.tile_count = pick.data_ptr(), pick.numel()
    _clx.call("clx_conv_fwd", ctypes.byref(d), st)
    ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double()).permute(0, 2, 3, 1) + bias.double())
    px = listed.repeat_interleave(4, 1).repeat_interleave(4, 2)[:, :OH, :OW]
    assert (out.cpu().double() - ref)[px].abs().max().item() < TOL * ref.abs().max().item()
    want = F.max_pool2d(out.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).cpu()
    ppx = listed.repeat_interleave(2, 1).repeat_interleave(2, 2)[:, :OH // 2, :OW // 2]
    assert torch.equal(pooled.cpu()[ppx], want[ppx])
    assert (pooled.cpu()[~ppx] == -7.0).all() and (out.cpu()[~px] == -7.0).all()


def test_fused_winograd_rejects_what_it_does_not_know(device):
    from cellulus_amd import _clx

    lib = _clx.load()
    x_d = torch.zeros(1, 12, 12, 16, device=device)
    d = _desc(x_d, 1, 12, 12, 12, 12, 16, 16, 3, 64)
    assert int(lib.clx_conv_fused_applicable(ctypes.byref(d))) == 1
    for field, value in (("N", 32), ("N", 96), ("KD", 3), ("PH", 2)):
        d2 = _desc(x_d, 1, 12, 12, 12, 12, 16, 16, 3, 64)
        setattr(d2, field, value)
        assert int(lib.clx_conv_fused_applicable(ctypes.byref(d2))) == 0, field
    d.algo = 3
    d.wpack = x_d.data_ptr()
    out = torch.zeros(1, 10, 10, 64, device=device)
    d.out, d.ld_out = out.data_ptr(), 64
    d.mask, d.ld_mask = out.data_ptr(), 64
    with pytest.raises(_clx.ClxError, match="mask"):
        _clx.call("clx_conv_fwd", ctypes.byref(d), _clx.stream_ptr(device))
    # pool_out / tile_list with the direct algorithm: an error, not a silently unwritten buffer
    d = _desc(x_d, 1, 12, 12, 12, 12, 16, 16, 3, 64)
    d.algo, d.wpack, d.out, d.ld_out = 0, x_d.data_ptr(), out.data_ptr(), 64
    d.pool_out, d.ld_pool = out.data_ptr(), 64
    with pytest.raises(_clx.ClxError, match="pool_out"):
        _clx.call("clx_conv_fwd", ctypes.byref(d), _clx.stream_ptr(device))
