"""GPU: the fused pair of 64-channel 1x1 convolutions (clx_chain64_fwd / clx_chain64_bwd) through
the C ABI against float64 autograd of the same two layers (funlib ConvPass's kernel sizes [3, 1, 1, 3]
and the head of cellulus/models/unet.py:52-63), incl. ragged pixel counts, the head's narrow last
layer, the ReLU gate bits and a padded leading dimension."""

import os

import numpy as np
import pytest
import torch

from cellulus_amd import _clx

pytestmark = pytest.mark.gpu


def _pack_fwd(w, rows_pad):          # clx_pack_weights(CLX_PACK_FWD), one tap: [pad4(cout)][64]
    out = torch.zeros(rows_pad, 64)
    out[:w.shape[0]] = w
    return out


def _pack_dgrad(w, cols_pad):        # CLX_PACK_DGRAD, one tap: [64][pad4(cout)]
    out = torch.zeros(64, cols_pad)
    out[:, :w.shape[0]] = w.t()
    return out


def _bits(gate, M, C):
    """(M, C/32) int32 words -> (M, C) bool"""
    g = gate.cpu().numpy().astype(np.uint32)
    return ((g[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool).reshape(M, -1)[:, :C]


@pytest.mark.parametrize("M,N2,relu2,ld_x", [(1000, 64, 1, 64), (128 * 7 + 5, 64, 1, 72), (33, 64, 0, 64),
                                             (4099, 3, 0, 64), (257, 2, 0, 64), (128 * 40, 64, 1, 64), (777, 7, 0, 68)])
def test_chain_forward_and_backward_match_float64_autograd(M, N2, relu2, ld_x, device):
    torch.manual_seed(M + N2)
    n2p = (N2 + 3) // 4 * 4
    x = torch.relu(torch.randn(M, 64))                 # the input IS a ReLU output (gate_x = 1)
    w1, b1 = torch.randn(64, 64) / 8, torch.randn(64) * 0.1
    w2, b2 = torch.randn(N2, 64) / 8, torch.randn(N2) * 0.1
    xd = torch.zeros(M, ld_x, device=device)
    xd[:, :64] = x.to(device)
    y1 = torch.full((M, 64), float("nan"), device=device)
    y2 = torch.full((M, n2p), float("nan"), device=device)
    gate1 = torch.zeros((M, 2), dtype=torch.int32, device=device)
    gate2 = torch.zeros((M, 2), dtype=torch.int32, device=device) if (relu2 and N2 == 64) else None
    w1p, w2p = _pack_fwd(w1, 64).to(device), _pack_fwd(w2, n2p).to(device)
    b1d, b2d = b1.to(device), b2.to(device)
    st = _clx.stream_ptr(device)
    _clx.call("clx_chain64_fwd", _clx.ptr(xd), ld_x, M, _clx.ptr(w1p), _clx.ptr(b1d), _clx.ptr(y1), 64,
              _clx.ptr(gate1), 2, _clx.ptr(w2p), _clx.ptr(b2d), N2, relu2, _clx.ptr(y2), n2p,
              _clx.ptr(gate2), 2, st)

    x64 = x.double().requires_grad_(True)
    p64 = [t.double().requires_grad_(True) for t in (w1, b1, w2, b2)]
    r1 = torch.relu(x64 @ p64[0].t() + p64[1])
    pre2 = r1 @ p64[2].t() + p64[3]
    r2 = torch.relu(pre2) if relu2 else pre2
    assert (y1.cpu().double() - r1.detach()).abs().max().item() < 2e-5
    assert (y2.cpu().double()[:, :N2] - r2.detach()).abs().max().item() < 2e-5
    if n2p > N2:
        assert not y2[:, N2:].any()                    # padded channels stay zero
    np.testing.assert_array_equal(_bits(gate1, M, 64), (y1 > 0).cpu().numpy())
    if gate2 is not None:
        np.testing.assert_array_equal(_bits(gate2, M, 64), (y2 > 0).cpu().numpy())
    # inference form: nothing kept of layer 1
    y2b = torch.empty_like(y2)
    _clx.call("clx_chain64_fwd", _clx.ptr(xd), ld_x, M, _clx.ptr(w1p), _clx.ptr(b1d), None, 0, None, 0,
              _clx.ptr(w2p), _clx.ptr(b2d), N2, relu2, _clx.ptr(y2b), n2p, None, 0, st)
    assert torch.equal(y2b, y2)

    # ---- backward: dp2 = gradient w.r.t. layer 2's pre-activation
    if N2 == 64 or N2 <= 8:
        dp2 = torch.randn(M, N2)
        if relu2:
            dp2 = dp2 * (pre2.detach() > 0).float()
        pre2.backward(dp2.double())
        dp2d = torch.zeros(M, n2p, device=device)
        dp2d[:, :N2] = dp2.to(device)
        dp0 = torch.full((M, 64), float("nan"), device=device)
        dw2 = torch.zeros(n2p, 64, device=device)
        dw1 = torch.zeros(64, 64, device=device)
        db2 = torch.zeros(N2, device=device)
        db1 = torch.zeros(64, device=device)
        w2t, w1t = _pack_dgrad(w2, n2p).to(device), _pack_dgrad(w1, 64).to(device)
        # y1 as the GPU forward wrote it: the gate decisions are the forward's own
        _clx.call("clx_chain64_bwd", _clx.ptr(dp2d), n2p, N2, _clx.ptr(y1), 64, _clx.ptr(xd), ld_x, 1, M,
                  _clx.ptr(w2t), _clx.ptr(w1t), _clx.ptr(dp0), 64, _clx.ptr(dw2), _clx.ptr(db2), _clx.ptr(dw1),
                  _clx.ptr(db1), st)
        # autograd's dL/dx has no gate (x is a leaf); the kernel's dp0 is gated by x > 0
        ref_dp0 = x64.grad * (x64.detach() > 0)
        scale = ref_dp0.abs().max().item()
        assert (dp0.cpu().double() - ref_dp0).abs().max().item() < 1e-5 * max(scale, 1.0)

        def rel(a, b):
            return ((a.cpu().double() - b).norm() / (b.norm() + 1e-30)).item()

        assert rel(dw2[:N2], p64[2].grad) < 1e-5 and rel(db2, p64[3].grad) < 1e-5
        assert rel(dw1, p64[0].grad) < 1e-5 and rel(db1, p64[1].grad) < 1e-5
        if n2p > N2:
            assert not dw2[N2:].any()
        # without a data gradient (dp0 NULL) and without bias gradients the weight gradients are the same sums
        dw2b, dw1b = torch.zeros_like(dw2), torch.zeros_like(dw1)
        _clx.call("clx_chain64_bwd", _clx.ptr(dp2d), n2p, N2, _clx.ptr(y1), 64, _clx.ptr(xd), ld_x, 1, M,
                  _clx.ptr(w2t), _clx.ptr(w1t), None, 0, _clx.ptr(dw2b), None, _clx.ptr(dw1b), None, st)
        assert rel(dw2b[:N2], p64[2].grad) < 1e-5 and rel(dw1b, p64[0].grad) < 1e-5


def test_chain_rejects_what_it_cannot_do(device):
    x = torch.zeros(64, 64, device=device)
    st = _clx.stream_ptr(device)
    with pytest.raises(_clx.ClxError):              # 48 output channels: neither 64 nor <= 32
        _clx.call("clx_chain64_fwd", _clx.ptr(x), 64, 64, _clx.ptr(x), None, None, 0, None, 0, _clx.ptr(x), None,
                  48, 0, _clx.ptr(x), 64, None, 0, st)
    with pytest.raises(_clx.ClxError):              # leading dimension shorter than the channels
        _clx.call("clx_chain64_fwd", _clx.ptr(x), 32, 64, _clx.ptr(x), None, None, 0, None, 0, _clx.ptr(x), None,
                  64, 0, _clx.ptr(x), 64, None, 0, st)
    with pytest.raises(_clx.ClxError):
        _clx.call("clx_chain64_bwd", _clx.ptr(x), 64, 16, _clx.ptr(x), 64, _clx.ptr(x), 64, 1, 64, _clx.ptr(x),
                  _clx.ptr(x), None, 0, _clx.ptr(x), None, _clx.ptr(x), None, st)


@pytest.mark.parametrize("M,N,K", [(1000, 256, 256), (517, 768, 96), (129, 132, 36), (4100, 100, 260), (300, 260, 68),
                                   (128 * 9, 128, 32), (50, 65, 8)])
def test_plain_products_every_epilogue(M, N, K, device, monkeypatch):
    """Plain products (the 1x1 convolutions and the Winograd-domain GEMMs) through clx_conv_fwd: every epilogue of
    the implicit-GEMM kernel — plain, bias + ReLU with gate bits out, float ReLU-gate mask, gate bits in,
    accumulate — on pixel counts, channel counts and contraction lengths that are not multiples of the tile,
    against float64; and bit-identical gate semantics."""
    import ctypes

    from cellulus_amd._clx import ClxConvDesc, ClxSrc

    torch.manual_seed(M + N + K)
    st = _clx.stream_ptr(device)
    ldx = K + 4                                        # a padded leading dimension
    x = torch.randn(M, K)
    xd = torch.zeros(M, ldx, device=device)
    xd[:, :K] = x.to(device)
    w = torch.randn(N, K) / K ** 0.5
    wd = w.to(device).contiguous()
    bias = torch.randn(N)
    ldo = (N + 31) // 32 * 32
    ref = x.double() @ w.double().t()

    def desc(out):
        d = ClxConvDesc()
        d.nsrc = 1
        s = ClxSrc()
        s.ptr, s.C, s.ld = xd.data_ptr(), K, ldx
        s.D, s.H, s.W = 1, 1, M
        s.oz = s.oy = s.ox = 0
        s.fz = s.fy = s.fx = 1
        d.src[0] = s
        d.B = 1
        d.ID, d.IH, d.IW = 1, 1, M
        d.KD = d.KH = d.KW = 1
        d.PD = d.PH = d.PW = 0
        d.N = N
        d.wpack = wd.data_ptr()
        d.out, d.ld_out = out.data_ptr(), ldo
        return d

    def run(setup):
        out = torch.full((M, ldo), 7.0, device=device)
        d = desc(out)
        keep = setup(d, out)
        _clx.call("clx_conv_fwd", ctypes.byref(d), st)
        torch.cuda.synchronize()
        return out, keep

    tol = 2e-5 * max(1.0, ref.abs().max().item())
    out, _ = run(lambda d, o: None)
    assert (out[:, :N].cpu().double() - ref).abs().max().item() < tol
    assert (out[:, N:] == 7.0).all()                   # nothing written beyond N

    b_d = bias.to(device)
    gate = torch.zeros((M, ldo // 32), dtype=torch.int32, device=device)

    def bias_relu(d, o):
        d.bias, d.relu = b_d.data_ptr(), 1
        d.gate_out, d.ld_gate = gate.data_ptr(), ldo // 32
    out, _ = run(bias_relu)
    want = torch.relu(ref + bias.double())
    assert (out[:, :N].cpu().double() - want).abs().max().item() < tol
    bits = ((gate.cpu().numpy().astype(np.uint32)[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)
    bits = bits.reshape(M, -1)
    np.testing.assert_array_equal(bits[:, :N], (out[:, :N] > 0).cpu().numpy())
    assert not bits[:, N:].any()

    mask = torch.randn(M, ldo, device=device)
    def float_mask(d, o):
        d.mask, d.ld_mask = mask.data_ptr(), ldo
    out, _ = run(float_mask)
    assert (out[:, :N].cpu().double() - ref * (mask[:, :N] > 0).cpu()).abs().max().item() < tol

    def bit_mask(d, o):                                # the gates written above, read back as the mask
        d.mask_bits, d.ld_mask_bits = gate.data_ptr(), ldo // 32
    out, _ = run(bit_mask)
    assert (out[:, :N].cpu().double() - ref * torch.from_numpy(bits[:, :N])).abs().max().item() < tol

    prev = torch.randn(M, ldo, device=device)
    def accumulate(d, o):
        o.copy_(prev)
        d.accumulate, d.bias, d.relu = 1, b_d.data_ptr(), 1
    out, _ = run(accumulate)
    want = torch.relu(ref + bias.double() + prev[:, :N].cpu().double())
    assert (out[:, :N].cpu().double() - want).abs().max().item() < tol

    # (the launch profile proves which kernel ran)
    _clx.call("clx_profile_enable", 2)
    run(lambda d, o: None)
    n_l, ms_l, fl_l = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    _clx.load().clx_profile_read(0 if N > 64 and -(-N // 128) * 128 / N <= 1.2 else 1, ctypes.byref(n_l), ctypes.byref(ms_l),
                                 ctypes.byref(fl_l))
    _clx.call("clx_profile_enable", 0)
    assert n_l.value == 1 and fl_l.value == 2.0 * M * N * K
