"""Times clx_conv_wgrad (implicit-GEMM MFMA kernel) on one layer shape: B x H x W input, C -> N, k x k."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellulus_amd import _clx
from cellulus_amd._clx import ClxConvDesc, ClxSrc

if os.environ.get("CLX_LIB"):
    _clx.LIB_PATH = os.path.abspath(os.environ["CLX_LIB"])
dev = torch.device("cuda:0")
B, H, W, C, N, k = (int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (8, 124, 124, 768, 768, 1)))
x = torch.randn(B * H * W, C, device=dev)
OH, OW = H - k + 1, W - k + 1
M = B * OH * OW
dy = torch.randn(M, N, device=dev)
dw = torch.zeros(k * k * N * C, device=dev); db = torch.zeros(N, device=dev)
d = ClxConvDesc(); d.nsrc = 1
s = ClxSrc(); s.ptr = x.data_ptr(); s.C = C; s.ld = C; s.D, s.H, s.W = 1, H, W; s.oz = s.oy = s.ox = 0; s.fz = s.fy = s.fx = 1
d.src[0] = s; d.B = B; d.ID, d.IH, d.IW = 1, H, W; d.KD, d.KH, d.KW = 1, k, k; d.PD = d.PH = d.PW = 0; d.N = N
d.accumulate = 0; d.algo = 0; d.workspace = None; d.workspace_bytes = 0
st = _clx.stream_ptr(dev)
def run():
    _clx.call("clx_conv_wgrad", ctypes.byref(d), _clx.ptr(dy), N, _clx.ptr(dw), _clx.ptr(db), st)
run(); torch.cuda.synchronize()
if k == 1:
    ref = dy.double().t() @ x.double()
    got = dw.view(N, C).double()
    print("max rel err", ((got - ref).abs().max() / ref.abs().max()).item())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"B={B} {H}x{W} C={C} N={N} k={k}: {ms:.3f} ms  {2.0 * M * N * C * k * k / ms / 1e9:.1f} TF/s")
