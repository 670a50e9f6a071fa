"""Split-precision operands (csrc/gemm_sp.hip) through the C ABI: the P3 planes hold an EXACT three-way bfloat16 split
of the float32 operand, and the product made from them meets the float32 kernels' bars against float64.

Replaces the same reference arithmetic as the default kernels: nn.Conv{2,3}d in float32 (cellulus/models/unet.py:24-63).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from cellulus_amd import _clx

    return _clx, _clx.load()


def _planes(x):
    _clx, lib = _lib()
    rows, K = x.shape
    buf = torch.empty(lib.clx_planes_bytes(rows, K), dtype=torch.uint8, device=x.device)
    _clx.call("clx_split_planes", _clx.ptr(x), x.stride(0), rows, K, _clx.ptr(buf), _clx.stream_ptr(x.device))
    return buf


@pytest.mark.parametrize("rows,K,ld", [(64, 64, 64), (1000, 256, 256), (33, 16, 20), (4097, 768, 772)])
def test_planes_are_an_exact_split(rows, K, ld):
    _clx, lib = _lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(rows)
    full = torch.randn(rows, ld, device=dev) * torch.logspace(-20, 20, ld, device=dev)[None, :]
    full[0, :4] = torch.tensor([0.0, -0.0, 1.0, -1.5], device=dev)
    x = full[:, :K]
    buf = _planes(x)
    prows = max(128, (rows + 63) // 64 * 64)
    assert buf.numel() == prows // 32 * (K // 16) * 3072
    # the inverse (h0 + h1 + h2 in float32) gives the operand back bit for bit
    back = torch.full((rows, K), float("nan"), device=dev)
    _clx.call("clx_join_planes", _clx.ptr(buf), rows, K, _clx.ptr(back), K, _clx.stream_ptr(dev))
    assert torch.equal(back, x.contiguous())            # (values: -0.0 comes back as +0.0)
    # ... and the layout is the documented one: fragment (rb, ks, p), 16 bytes at 512 h + 16 r
    h = buf.cpu().numpy().view(np.uint16).reshape(prows // 32, K // 16, 3, 2, 32, 8)
    pieces = (h.astype(np.uint32) << 16).view(np.float32)                    # bf16 -> f32
    xs = np.zeros((prows, K), np.float32)
    xs[:rows] = x.cpu().numpy()
    want = xs.reshape(-1, 32, K // 16, 2, 8).transpose(0, 2, 3, 1, 4)        # [rb][ks][h][r][8]
    got = (pieces[:, :, 0].astype(np.float64) + pieces[:, :, 1] + pieces[:, :, 2])
    assert np.array_equal(got.astype(np.float32), want)
    # (every piece has at most 8 significant bits by construction: it IS a bfloat16); the padding rows are zero
    assert not np.isnan(pieces).any()
    assert np.all(h.transpose(0, 4, 1, 2, 3, 5).reshape(-1, K // 16 * 48)[rows:] == 0)


@pytest.mark.parametrize("kernel", ["tile256", "tile128", "mfma16"])
@pytest.mark.parametrize("M,N,K,relu", [(256, 128, 128, 0), (1000, 256, 256, 1), (70000, 256, 768, 1), (257, 768, 2304, 0)])
def test_product_from_planes_against_float64(M, N, K, relu, kernel, monkeypatch):
    """The three forward kernels on every shape (the library picks by K; the switches are read per launch): 256 x 128 tiles,
    one workgroup per CU (gemm_sp_kernel<0>: K > 1024); 128 x 128 tiles, two per CU (gemm_sp2_kernel: K <= 1024); the opt-in
    one on v_mfma_f32_16x16x32_bf16 (CLX_SP_MFMA=16): the same products, the same bars"""
    if kernel == "mfma16":
        monkeypatch.setenv("CLX_SP_MFMA", "16")
    else:
        monkeypatch.delenv("CLX_SP_MFMA", raising=False)
        monkeypatch.setenv("CLX_SP_TILE", kernel[4:])
    _clx, lib = _lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(M + K)
    x = torch.relu(torch.randn(M, K, device=dev))
    w = torch.randn(N, K, device=dev) / K ** 0.5
    bias = torch.randn(N, device=dev)
    out = torch.full((M, N + 4), float("nan"), device=dev)
    pa, pb = _planes(x), _planes(w)
    _clx.call("clx_gemm_planes", _clx.ptr(pa), _clx.ptr(pb), M, N, K, _clx.ptr(bias), relu, _clx.ptr(out), N + 4,
              _clx.stream_ptr(dev))
    ref = x.double() @ w.double().t() + bias.double()
    if relu:
        ref = torch.relu(ref)
    got = out[:, :N].double()
    assert torch.isnan(out[:, N:]).all()                           # nothing written past N
    rms = ref.pow(2).mean().sqrt().item()
    err = (got - ref).abs().max().item()
    rel = ((got - ref).pow(2).mean().sqrt() / rms).item()
    bias_err = abs(((got - ref).mean() / rms).item())
    # the float32 FMA chain of the default kernel sits at 1e-7 .. 3e-7 relative L2 on these shapes
    assert rel < 3e-7, (rel, err)
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err
    assert bias_err < 2e-8, bias_err                               # the alternating sign removes the accumulation bias


@pytest.mark.parametrize("rows,N,C", [(128, 128, 128), (5, 128, 128), (1000, 256, 128), (70001, 768, 256), (33000, 128, 768)])
def test_weight_gradient_from_planes_against_float64(rows, N, C):
    _clx, lib = _lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(rows + N)
    x = torch.relu(torch.randn(rows, C, device=dev))
    dy = torch.randn(rows, N, device=dev) * (torch.rand(rows, N, device=dev) < 0.5)          # a gated gradient
    dw = torch.full((N, C + 4), 1.0, device=dev)                                              # += into what is there
    pdy, px = _planes(dy), _planes(x)              # (kept alive: a temporary's memory would be handed to the next allocation)
    _clx.call("clx_wgrad_planes", _clx.ptr(pdy), _clx.ptr(px), rows, N, C, _clx.ptr(dw), C + 4, _clx.stream_ptr(dev))
    ref = dy.double().t() @ x.double() + 1.0
    got = dw[:, :C].double()
    assert torch.equal(dw[:, C:], torch.ones(N, 4, device=dev))
    rms = (ref - 1.0).pow(2).mean().sqrt().item()
    rel = ((got - ref).pow(2).mean().sqrt() / rms).item()
    assert rel < 3e-7, rel
    assert abs(((got - ref).mean() / rms).item()) < 3e-8


def test_rejects_what_it_cannot_do():
    _clx, lib = _lib()
    dev = torch.device("cuda:0")
    a = torch.zeros(3072 * 8, dtype=torch.uint8, device=dev)
    out = torch.zeros(64, 128, device=dev)
    with pytest.raises(_clx.ClxError):
        _clx.call("clx_gemm_planes", _clx.ptr(a), _clx.ptr(a), 64, 96, 64, None, 0, _clx.ptr(out), 128, _clx.stream_ptr(dev))
    with pytest.raises(_clx.ClxError):
        _clx.call("clx_gemm_planes", _clx.ptr(a), _clx.ptr(a), 64, 128, 48, None, 0, _clx.ptr(out), 128, _clx.stream_ptr(dev))
    assert lib.clx_planes_bytes(100, 40) == 0
