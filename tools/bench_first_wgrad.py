"""Times clx_conv_wgrad on the first-layer shape (2-D cfg: x [8,256,256,4 (1 real)], dy [8*254*254, 256])."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellulus_amd import _clx
from cellulus_amd._clx import ClxConvDesc, ClxSrc

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
CIN = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B, H, W = (int(v) for v in sys.argv[3].split("x")) if len(sys.argv) > 3 else (8, 256, 256)
x = torch.zeros(B * H * W, 4, device=dev); x[:, :CIN] = torch.rand(B * H * W, CIN, device=dev)
M = B * (H - 2) * (W - 2)
dy = torch.randn(M, N, device=dev)
dw = torch.zeros(9 * N * 4, device=dev); db = torch.zeros(N, device=dev)
d = ClxConvDesc(); d.nsrc = 1
s = ClxSrc(); s.ptr = x.data_ptr(); s.C = 4; s.ld = 4; s.D, s.H, s.W = 1, H, W; s.oz = s.oy = s.ox = 0; s.fz = s.fy = s.fx = 1
d.src[0] = s; d.B = B; d.ID, d.IH, d.IW = 1, H, W; d.KD, d.KH, d.KW = 1, 3, 3; d.PD = d.PH = d.PW = 0; d.N = N
d.accumulate = 0; d.algo = 0; d.workspace = None; d.workspace_bytes = 0
st = _clx.stream_ptr(dev)
def run():
    _clx.call("clx_conv_wgrad", ctypes.byref(d), _clx.ptr(dy), N, _clx.ptr(dw), _clx.ptr(db), st)
run(); torch.cuda.synchronize()
# reference check on a few entries
ref = torch.zeros(9, N, 4, device=dev, dtype=torch.float64)
xi = x.view(B, H, W, 4).double(); dyi = dy.view(B, H - 2, W - 2, N).double()
for ty in range(3):
    for tx in range(3):
        ref[ty * 3 + tx] = torch.einsum("bhwc,bhwn->nc", xi[:, ty:ty + H - 2, tx:tx + W - 2], dyi)
got = dw.view(9, N, 4).double()
print("max rel err", ((got - ref).abs().max() / ref.abs().max()).item(), "bias err", (db.double() - dyi.sum((0, 1, 2))).abs().max().item())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"N={N}: {ms:.3f} ms  dy read {M * N * 4 / ms / 1e9:.2f} TB/s")
