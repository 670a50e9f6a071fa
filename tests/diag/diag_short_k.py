"""Short-K plain GEMMs (N = 64: the 1x1x1 layers and Winograd-domain products of the 3-D configuration) timed
alone through clx_conv_fwd, next to a device copy of the same bytes and rocBLAS on the same product.
Usage: python tests/diag/diag_short_k.py"""
import ctypes
import sys

sys.path.insert(0, ".")
import torch  # noqa: E402

from cellulus_amd import _clx  # noqa: E402
from cellulus_amd._clx import ClxConvDesc, ClxSrc  # noqa: E402

dev = torch.device("cuda:0")
st = _clx.stream_ptr(dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (B, D, plane, kd, C, N, relu) in [(8, 1, 62 ** 3, 1, 64, 64, 1), (8, 1, 50 ** 3, 1, 64, 64, 0), (8 * 36, 62, 16 * 16, 3, 64, 64, 0),
                                      (8, 1, 28 ** 3, 1, 192, 64, 0), (8, 1, 58 ** 3, 1, 192, 64, 0), (8, 1, 58 ** 3, 1, 128, 64, 0), (8 * 36, 62, 16 * 16, 1, 64, 64, 0)]:
    x = torch.randn(B, D, plane, C, device=dev)
    w = torch.randn(N, kd * C, device=dev) * 0.1
    OD = D - kd + 1
    M = B * OD * plane
    out = torch.empty(M, N, device=dev)
    d = ClxConvDesc()
    d.nsrc = 1
    s = ClxSrc()
    s.ptr, s.C, s.ld = x.data_ptr(), C, C
    s.D, s.H, s.W = D, 1, plane
    s.fz = s.fy = s.fx = 1
    d.src[0] = s
    d.B, d.ID, d.IH, d.IW = B, D, 1, plane
    d.KD, d.KH, d.KW = kd, 1, 1
    d.N = N
    d.relu = relu
    d.wpack = w.data_ptr()
    d.out, d.ld_out = out.data_ptr(), N
    ms = timed(lambda: _clx.call("clx_conv_fwd", ctypes.byref(d), st))
    byts = (x.numel() + out.numel()) * 4
    flops = 2.0 * M * N * kd * C
    line = f"M={M:8d} K={kd * C:4d} N={N}: clx {ms:.3f} ms  {flops / ms / 1e9:6.1f} TF/s  {byts / ms / 1e9:5.2f} TB/s (in+out once)"
    src = torch.empty(byts // 8, device=dev)
    dst = torch.empty_like(src)
    cms = timed(lambda: dst.copy_(src))
    line += f" | copy of the same bytes {cms:.3f} ms {byts / cms / 1e9:5.2f} TB/s"
    if kd == 1:
        x2 = x.view(M, C)
        bms = timed(lambda: torch.mm(x2, w.t(), out=out))
        line += f" | rocBLAS {bms:.3f} ms"
    print(line)
