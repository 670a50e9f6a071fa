"""GPU: the GEMM kernels' staging paths without vector address arithmetic, through the C ABI against
float64 torch: implicit-GEMM convolution with rows clamped at ragged M / N tile edges (no padding, channels % 32 == 0),
with range-checked buffer loads (padding), two concatenated sources (crop + nearest upsample), every epilogue option;
weight gradient with linear rows (1x1) and with the general decode (3x3), ragged extents."""

import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _conv_desc(x_d, B, D, H, W, C, kd, k, pad, N):
    from cellulus_amd._clx import ClxConvDesc, ClxSrc

    d = ClxConvDesc()
    d.nsrc = 1
    s = ClxSrc()
    s.ptr, s.C, s.ld = x_d.data_ptr(), C, C
    s.D, s.H, s.W = D, H, W
    s.oz = s.oy = s.ox = 0
    s.fz = s.fy = s.fx = 1
    d.src[0] = s
    d.B = B
    d.ID, d.IH, d.IW = D, H, W
    d.KD, d.KH, d.KW = kd, k, k
    d.PD, d.PH, d.PW = (kd - 1 if pad else 0), (k - 1 if pad else 0), (k - 1 if pad else 0)
    d.N = N
    d.algo = 0
    return d


@pytest.mark.parametrize("C,N,k,kd,pad", [(32, 136, 1, 1, False), (64, 36, 1, 1, False), (96, 200, 3, 1, False),
                                         (64, 72, 3, 1, True), (32, 64, 1, 3, True), (64, 130, 3, 3, False),
                                         (40, 64, 3, 1, False)])      # (the last: channels % 32 != 0 -> the general path)
def test_implicit_gemm_staging_paths_vs_f64(C, N, k, kd, pad, device):
    from cellulus_amd import _clx

    torch.manual_seed(C + 7 * N + k + 3 * kd + int(pad))
    B, D, H, W = 2, (4 if kd > 1 else 1), 21, 19                # M = B * OD * OH * OW: never a multiple of 128
    P, PZ = (k - 1 if pad else 0), (kd - 1 if pad else 0)
    OD, OH, OW = D + 2 * PZ - kd + 1, H + 2 * P - k + 1, W + 2 * P - k + 1
    x = torch.randn(B, D, H, W, C)
    w = torch.randn(N, C, kd, k, k) * 0.1
    bias = torch.randn(N)
    ref = F.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.double(), padding=(PZ, P, P)).permute(0, 2, 3, 4, 1)
    st = _clx.stream_ptr(device)
    x_d = x.to(device).contiguous()
    w_d = w.reshape(N, C, kd * k * k).to(device).contiguous()
    Np = (N + 3) // 4 * 4
    wp = torch.empty(Np * kd * k * k * C, device=device)
    _clx.call("clx_pack_weights", _clx.ptr(w_d), _clx.ptr(wp), N, C, kd * k * k, C, Np, 0, st)
    prev = torch.randn(B, OD, OH, OW, Np)
    gate = torch.randn(B, OD, OH, OW, Np)
    for mode in ("plain", "bias_relu", "accumulate", "mask"):
        d = _conv_desc(x_d, B, D, H, W, C, kd, k, pad, N)
        d.wpack = wp.data_ptr()
        out = prev.to(device).clone() if mode == "accumulate" else torch.full((B, OD, OH, OW, Np), 7.0, device=device)
        d.out, d.ld_out = out.data_ptr(), Np
        want = ref
        if mode == "bias_relu":
            b_d = bias.to(device)
            d.bias, d.relu = b_d.data_ptr(), 1
            want = torch.relu(ref + bias.double())
        elif mode == "accumulate":
            d.accumulate = 1
            want = ref + prev[..., :N].double()
        elif mode == "mask":
            g_d = gate.to(device).contiguous()
            d.mask, d.ld_mask = g_d.data_ptr(), Np
            want = ref * (gate[..., :N] > 0)
        _clx.call("clx_conv_fwd", ctypes.byref(d), st)
        got = out.cpu().double()
        err = (got[..., :N] - want).abs().max().item()
        assert err < 3e-5 * max(1.0, want.abs().max().item()), (mode, err)
        if mode != "accumulate" and Np > N:
            assert torch.all(got[..., N:] == 7.0), "columns past N must not be written"


def test_two_concatenated_sources_with_crop_and_upsample(device):
    """conv over cat(crop(skip), nearest-upsample(low)) — the right path's first layer in its general (not sub-pixel)
    form — with 32- and 64-channel sources: both take the clamped-row loads."""
    from cellulus_amd import _clx
    from cellulus_amd._clx import ClxConvDesc, ClxSrc

    torch.manual_seed(3)
    B, C0, C1, N, k = 2, 32, 64, 48, 3
    SH, SW, LH, LW = 30, 28, 12, 11                      # skip grid, low-res grid
    IH, IW = 2 * LH - 2, 2 * LW - 2                      # the upsampled grid cropped by 1 low-res pixel on each side
    cy, cx = (SH - IH) // 2, (SW - IW) // 2
    skip, low = torch.randn(B, SH, SW, C0), torch.randn(B, LH, LW, C1)
    w = torch.randn(N, C0 + C1, k, k) * 0.1
    up = low.repeat_interleave(2, 1).repeat_interleave(2, 2)[:, 1:1 + IH, 1:1 + IW]
    cat = torch.cat([skip[:, cy:cy + IH, cx:cx + IW], up], dim=-1)
    ref = F.conv2d(cat.permute(0, 3, 1, 2).double(), w.double()).permute(0, 2, 3, 1)
    st = _clx.stream_ptr(device)
    s_d, l_d = skip.to(device).contiguous(), low.to(device).contiguous()
    w_d = w.reshape(N, C0 + C1, k * k).to(device).contiguous()
    wp = torch.empty(N * k * k * (C0 + C1), device=device)
    _clx.call("clx_pack_weights", _clx.ptr(w_d), _clx.ptr(wp), N, C0 + C1, k * k, C0 + C1, N, 0, st)
    d = ClxConvDesc()
    d.nsrc = 2
    a = ClxSrc()
    a.ptr, a.C, a.ld, a.D, a.H, a.W = s_d.data_ptr(), C0, C0, 1, SH, SW
    a.oz, a.oy, a.ox, a.fz, a.fy, a.fx = 0, cy, cx, 1, 1, 1
    b = ClxSrc()
    b.ptr, b.C, b.ld, b.D, b.H, b.W = l_d.data_ptr(), C1, C1, 1, LH, LW
    b.oz, b.oy, b.ox, b.fz, b.fy, b.fx = 0, 1, 1, 1, 2, 2
    d.src[0], d.src[1] = a, b
    d.B, d.ID, d.IH, d.IW = B, 1, IH, IW
    d.KD, d.KH, d.KW = 1, k, k
    d.PD = d.PH = d.PW = 0
    d.N, d.algo = N, 0
    d.wpack = wp.data_ptr()
    out = torch.empty(B, IH - 2, IW - 2, N, device=device)
    d.out, d.ld_out = out.data_ptr(), N
    _clx.call("clx_conv_fwd", ctypes.byref(d), st)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 3e-5 * ref.abs().max().item(), err


@pytest.mark.parametrize("C,N,k", [(64, 136, 1), (200, 36, 1), (96, 128, 3), (32, 260, 1)])
def test_weight_gradient_linear_and_decoded_rows_vs_f64(C, N, k, device):
    """1x1: output pixel m reads input pixel m (rows through per-chunk buffer descriptors, the last chunk ragged);
    3x3: the decoded rows; dY always through the descriptors.  N, C not multiples of the 64 / 128 tiles."""
    from cellulus_amd import _clx

    torch.manual_seed(C + N + k)
    B, H, W = 2, 23, 17
    OH, OW = H - k + 1, W - k + 1
    x = torch.randn(B, 1, H, W, C)
    dy = torch.randn(B, OH, OW, N)
    wr = torch.zeros(N, C, k, k, dtype=torch.float64, requires_grad=True)
    (F.conv2d(x[:, 0].permute(0, 3, 1, 2).double(), wr) * dy.permute(0, 3, 1, 2).double()).sum().backward()
    st = _clx.stream_ptr(device)
    x_d, dy_d = x.to(device).contiguous(), dy.to(device).contiguous()
    d = _conv_desc(x_d, B, 1, H, W, C, 1, k, False, N)
    dwp = torch.zeros(k * k * N * C, device=device)
    db = torch.zeros(N, device=device)
    _clx.call("clx_conv_wgrad", ctypes.byref(d), _clx.ptr(dy_d), N, _clx.ptr(dwp), _clx.ptr(db), st)
    dw = torch.empty(N, C, k * k, device=device)
    _clx.call("clx_unpack_wgrad", _clx.ptr(dwp), _clx.ptr(dw), N, C, k * k, N, C, st)
    err = (dw.cpu().double().reshape(N, C, k, k) - wr.grad).abs().max().item()
    assert err < 3e-5 * wr.grad.abs().max().item(), err
    assert (db.cpu().double() - dy.double().sum((0, 1, 2))).abs().max().item() < 1e-3


_FLUSH_SCRIPT = r"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
from cellulus_amd import _clx
from cellulus_amd._clx import ClxConvDesc, ClxSrc
dev = torch.device("cuda:0")
st = _clx.stream_ptr(dev)
res = {}
for C in (32, 256):                       # one K chunk of 32 / eight of them
    torch.manual_seed(C)
    B, H, W, N = 2, 40, 36, 64
    x = torch.relu(torch.randn(B, H, W, C)).to(dev).contiguous()
    w = (torch.randn(N, C, 1) * 0.2).to(dev).contiguous()
    wp = torch.empty(N * C, device=dev)
    _clx.call("clx_pack_weights", _clx.ptr(w), _clx.ptr(wp), N, C, 1, C, N, 0, st)
    out = torch.empty(B, H, W, N, device=dev)
    d = ClxConvDesc()
    d.nsrc = 1
    s = ClxSrc()
    s.ptr, s.C, s.ld = x.data_ptr(), C, C
    s.D, s.H, s.W = 1, H, W
    s.fz = s.fy = s.fx = 1
    d.src[0] = s
    d.B = B
    d.ID, d.IH, d.IW = 1, H, W
    d.KD = d.KH = d.KW = 1
    d.N = N
    d.algo = 0
    d.wpack = wp.data_ptr()
    d.out, d.ld_out = out.data_ptr(), N
    _clx.call("clx_conv_fwd", ctypes.byref(d), st)
    torch.cuda.synchronize()
    res[f"out{C}"] = out.cpu().numpy()
    res[f"ref{C}"] = (x.double().reshape(-1, C) @ w.double().reshape(N, C).t()).reshape(B, H, W, N).cpu().numpy()
np.savez(sys.argv[2], **res)
"""


def test_accumulator_flush_against_one_chain(tmp_path, device):
    """CLX_IGEMM_FLUSH (read once per process: two processes): fresh accumulators every 64 products (the default, 2) against one
    chain over all of K (0), float32 MFMA kernels.  One 32-deep chunk: the flush never fires, the bits are the same; eight
    chunks: both within the float32 bar of the float64 product, and the flushed sum no further from it than the chain."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "flush.py"
    script.write_text(_FLUSH_SCRIPT)
    got = {}
    for flush in ("0", "2"):
        env = dict(os.environ, CLX_IGEMM_FLUSH=flush, CLX_PRECISION="f32")
        out = tmp_path / f"flush{flush}.npz"
        p = subprocess.run([sys.executable, str(script), root, str(out)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
        got[flush] = np.load(out)
    assert np.array_equal(got["0"]["out32"], got["2"]["out32"])
    ref = got["0"]["ref256"]
    scale = np.abs(ref).max()
    e0 = np.abs(got["0"]["out256"] - ref).max() / scale
    e2 = np.abs(got["2"]["out256"] - ref).max() / scale
    assert e0 < 2e-6 and e2 < 2e-6, (e0, e2)
    r0 = np.sqrt(((got["0"]["out256"] - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean())
    r2 = np.sqrt(((got["2"]["out256"] - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean())
    assert r2 <= 1.1 * r0, (r0, r2)
