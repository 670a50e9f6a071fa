"""Layout probe of clx_wgrad_planes: one-hot dY, x[p][c] = 1 + c + 1000 p (exact in bf16 pieces), prints where things land."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellulus_amd import _clx
dev = torch.device("cuda:0"); st = _clx.stream_ptr(dev); lib = _clx.load()
def planes(x):
    buf = torch.empty(lib.clx_planes_bytes(x.shape[0], x.shape[1]), dtype=torch.uint8, device=dev)
    _clx.call("clx_split_planes", _clx.ptr(x), x.stride(0), x.shape[0], x.shape[1], _clx.ptr(buf), st)
    return buf
rows, N, C = 128, 128, 128
x = (1 + torch.arange(C, device=dev)[None, :] + 1000 * torch.arange(rows, device=dev)[:, None]).float()
px = planes(x)
for (p0, n0) in [(0, 0), (1, 0), (0, 1), (5, 3), (17, 40), (70, 100), (127, 127)]:
    dy = torch.zeros(rows, N, device=dev); dy[p0, n0] = 1.0
    dw = torch.zeros(N, C, device=dev)
    _clx.call("clx_wgrad_planes", _clx.ptr(planes(dy)), _clx.ptr(px), rows, N, C, _clx.ptr(dw), C, st)
    nz = dw.nonzero()
    rows_nz = sorted(set(nz[:, 0].tolist()))
    print(f"dy one-hot at pixel {p0} channel {n0}: nonzero rows of dw {rows_nz[:8]}{'...' if len(rows_nz) > 8 else ''} ({len(nz)} entries)")
    if len(rows_nz):
        r = rows_nz[0]
        vals = dw[r]
        print("   row", r, "first 8:", vals[:8].tolist(), " -> pixel", ((vals[0] - 1) // 1000).item(), "c of col0:", ((vals[0] - 1) % 1000).item(),
              "| col 1:", ((vals[1]-1) % 1000).item(), ((vals[1]-1)//1000).item(), "| col 16:", ((vals[16]-1) % 1000).item(), "| col 32:", ((vals[32]-1)%1000).item())
