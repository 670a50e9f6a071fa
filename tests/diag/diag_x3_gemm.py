"""Accuracy of the plain GEMM (1x1 convolution through clx_conv_fwd) in both precisions against float64:
relative L2 error and the signed mean error (bias) for zero-mean and for non-negative (post-ReLU) inputs.
Usage: python tests/diag/diag_x3_gemm.py"""
import ctypes
import sys

sys.path.insert(0, ".")
import torch  # noqa: E402

from cellulus_amd import _clx  # noqa: E402
from cellulus_amd._clx import ClxConvDesc, ClxSrc  # noqa: E402

dev = torch.device("cuda:0")
st = _clx.stream_ptr(dev)
M, N = 128 * 128, 256
for K in (64, 256, 768, 2304):
    for kind in ("normal", "relu"):
        torch.manual_seed(K)
        x = torch.randn(M, K)
        if kind == "relu":
            x = torch.relu(x)
        w = torch.randn(N, K) / K ** 0.5
        ref = x.double() @ w.double().t()
        x_d, w_d = x.to(dev), w.to(dev)
        line = f"K={K:5d} {kind:6s}"
        for prec in (0, 1):
            d = ClxConvDesc()
            d.nsrc = 1
            s = ClxSrc()
            s.ptr, s.C, s.ld = x_d.data_ptr(), K, K
            s.D, s.H, s.W = 1, 128, 128
            s.fz = s.fy = s.fx = 1
            d.src[0] = s
            d.B, d.ID, d.IH, d.IW = 1, 1, 128, 128
            d.KD = d.KH = d.KW = 1
            d.N = N
            d.wpack = w_d.data_ptr()
            out = torch.empty(M, N, device=dev)
            d.out, d.ld_out = out.data_ptr(), N
            d.precision = prec
            _clx.call("clx_conv_fwd", ctypes.byref(d), st)
            e = out.cpu().double() - ref
            line += f" | prec {prec}: rel L2 {(e.norm() / ref.norm()).item():.2e} mean err/rms {(e.mean() / ref.pow(2).mean().sqrt()).item():+.2e} max {e.abs().max().item():.2e}"
        e32 = (x @ w.t()).double() - ref
        line += f" | cpu f32: {(e32.norm() / ref.norm()).item():.2e}"
        print(line)
