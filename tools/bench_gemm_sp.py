"""Times the split-precision product (clx_gemm_planes, csrc/gemm_sp.hip) and its split pass against the float32-MFMA
implicit-GEMM kernel (clx_conv_fwd, 1x1 layer) on the plain-product shapes of the benchmark network, and reports both
kernels' error against a float64 product of the same rows.

    python tools/bench_gemm_sp.py [M N K ...]        # default: the 2-D training step's shapes
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellulus_amd import _clx
from cellulus_amd._clx import ClxConvDesc, ClxSrc

if os.environ.get("CLX_LIB"):
    _clx.LIB_PATH = os.path.abspath(os.environ["CLX_LIB"])
dev = torch.device("cuda:0")
st = _clx.stream_ptr(dev)
lib = _clx.load()


def planes_of(x):
    rows, K = x.shape
    nbytes = lib.clx_planes_bytes(rows, K)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _clx.call("clx_split_planes", _clx.ptr(x), x.stride(0), rows, K, _clx.ptr(buf), st)
    return buf


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def one(M, N, K, relu_in=True):
    torch.manual_seed(0)
    x = torch.randn(M, K, device=dev)
    if relu_in:
        x = torch.relu(x)                     # what a layer really sees: non-negative activations
    w = torch.randn(N, K, device=dev) * (1.0 / K ** 0.5)
    bias = torch.randn(N, device=dev)
    out_sp = torch.empty(M, N, device=dev)
    out_f32 = torch.empty(M, N, device=dev)
    pa = planes_of(x)
    pb = planes_of(w)

    def run_sp():
        _clx.call("clx_gemm_planes", _clx.ptr(pa), _clx.ptr(pb), M, N, K, _clx.ptr(bias), 1, _clx.ptr(out_sp), N, st)

    def run_split():
        _clx.call("clx_split_planes", _clx.ptr(x), K, M, K, _clx.ptr(pa), st)

    wp = w.contiguous()
    d = ClxConvDesc()
    d.nsrc = 1
    s = ClxSrc()
    s.ptr = x.data_ptr(); s.C = K; s.ld = K; s.D, s.H, s.W = 1, 1, M; s.oz = s.oy = s.ox = 0; s.fz = s.fy = s.fx = 1
    d.src[0] = s
    d.B = 1; d.ID, d.IH, d.IW = 1, 1, M; d.KD = d.KH = d.KW = 1; d.PD = d.PH = d.PW = 0; d.N = N
    d.wpack = wp.data_ptr(); d.bias = bias.data_ptr(); d.relu = 1; d.mask = None; d.ld_mask = 0
    d.out = out_f32.data_ptr(); d.ld_out = N; d.accumulate = 0; d.algo = 0; d.workspace = None; d.workspace_bytes = 0

    def run_f32():
        _clx.call("clx_conv_fwd", ctypes.byref(d), st)

    t_sp, t_split, t_f32 = timed(run_sp), timed(run_split), timed(run_f32)
    if os.environ.get("SP_TIME_ONLY"):
        print(f"M={M:7d} N={N:4d} K={K:5d}  sp {t_sp:7.3f} ms {2.0 * M * N * K / t_sp / 1e9:6.1f} TF/s", flush=True)
        return
    rows = min(M, 4096)
    ref = torch.relu(x[:rows].double() @ w.double().t() + bias.double())
    ref_t = torch.relu(x[M - rows:].double() @ w.double().t() + bias.double())
    scale = ref.abs().max().item()
    e_sp = max((out_sp[:rows].double() - ref).abs().max().item(), (out_sp[M - rows:].double() - ref_t).abs().max().item())
    e_f32 = max((out_f32[:rows].double() - ref).abs().max().item(), (out_f32[M - rows:].double() - ref_t).abs().max().item())
    r_sp = ((out_sp[:rows].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    r_f32 = ((out_f32[:rows].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    b_sp = ((out_sp[:rows].double() - ref).mean() / ref.pow(2).mean().sqrt()).item()
    b_f32 = ((out_f32[:rows].double() - ref).mean() / ref.pow(2).mean().sqrt()).item()
    fl = 2.0 * M * N * K
    print(f"M={M:7d} N={N:4d} K={K:5d}  sp {t_sp:7.3f} ms {fl / t_sp / 1e9:6.1f} TF/s | split(A) {t_split:6.3f} ms "
          f"{M * K * 10 / t_split / 1e9:5.2f} TB/s | f32 {t_f32:7.3f} ms {fl / t_f32 / 1e9:6.1f} TF/s | max err sp {e_sp:.2e} f32 {e_f32:.2e} "
          f"(range {scale:.1f}) rel-L2 sp {r_sp:.2e} f32 {r_f32:.2e} mean err / rms sp {b_sp:+.1e} f32 {b_f32:+.1e}", flush=True)


if __name__ == "__main__":
    args = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [
        (516128, 256, 256), (258064, 256, 256), (123008, 768, 768), (61504, 768, 768),
        (129032, 256, 256), (31752, 768, 768), (30976, 768, 256), (29768, 256, 768)]
    for M, N, K in shapes:
        one(M, N, K)


def one_wgrad(rows, N, C):
    """the weight-gradient product from planes against conv_wgrad_kernel on the same 1x1 layer"""
    torch.manual_seed(1)
    x = torch.relu(torch.randn(rows, C, device=dev))
    dy = torch.randn(rows, N, device=dev)
    pdy, px = planes_of(dy), planes_of(x)
    dw = torch.zeros(N, C, device=dev)
    dw32 = torch.zeros(N, C, device=dev)

    def run_sp():
        _clx.call("clx_wgrad_planes", _clx.ptr(pdy), _clx.ptr(px), rows, N, C, _clx.ptr(dw), C, st)

    d = ClxConvDesc()
    d.nsrc = 1
    s = ClxSrc()
    s.ptr = x.data_ptr(); s.C = C; s.ld = C; s.D, s.H, s.W = 1, 1, rows; s.oz = s.oy = s.ox = 0; s.fz = s.fy = s.fx = 1
    d.src[0] = s
    d.B = 1; d.ID, d.IH, d.IW = 1, 1, rows; d.KD = d.KH = d.KW = 1; d.PD = d.PH = d.PW = 0; d.N = N; d.algo = 0

    def run_f32():
        _clx.call("clx_conv_wgrad", ctypes.byref(d), _clx.ptr(dy), N, _clx.ptr(dw32), None, st)

    t_sp, t_f32 = timed(run_sp), timed(run_f32)
    fl = 2.0 * rows * N * C
    print(f"wgrad rows={rows:7d} N={N:4d} C={C:4d}  sp {t_sp:7.3f} ms {fl / t_sp / 1e9:6.1f} TF/s | f32 {t_f32:7.3f} ms {fl / t_f32 / 1e9:6.1f} TF/s", flush=True)


if __name__ == "__main__" and os.environ.get("SP_WGRAD"):
    for rows, N, C in [(516128, 256, 256), (258064, 256, 256), (123008, 768, 768), (61504, 768, 768)]:
        one_wgrad(rows, N, C)
