"""debug: dense vs tile-list runs of the fused kernel, k = 2 and 3: where do the bits differ?"""
import ctypes, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellulus_amd import _clx
from cellulus_amd._clx import ClxConvDesc, ClxSrc
dev = torch.device("cuda:0")
st = _clx.stream_ptr(dev)
for k, B, H, W, C, N in [(2, 2, 22, 27, 24, 64), (3, 2, 23, 30, 16, 64), (2, 1, 35, 34, 40, 128), (2, 2, 22, 27, 8, 64)]:
    torch.manual_seed(1)
    OH, OW = H - k + 1, W - k + 1
    x = torch.randn(B, H, W, C, device=dev)
    w = (torch.randn(N, C, k * k, device=dev) * 0.2).contiguous()
    nxi = (3 + k) ** 2
    wf = torch.empty(nxi * N * C, device=dev)
    _clx.call("clx_pack_weights", _clx.ptr(w), _clx.ptr(wf), N, C, k * k, C, N, 7, st)
    ws = torch.empty(64 << 20, device=dev)
    def run(tiles=None, fill=float("nan")):
        d = ClxConvDesc(); d.nsrc = 1
        s = ClxSrc(); s.ptr, s.C, s.ld = x.data_ptr(), C, C; s.D, s.H, s.W = 1, H, W; s.fz = s.fy = s.fx = 1
        d.src[0] = s; d.B = B; d.ID, d.IH, d.IW = 1, H, W; d.KD, d.KH, d.KW = 1, k, k; d.N = N; d.algo = 3
        d.wpack = wf.data_ptr()
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        out = torch.full((B, OH, OW, N), fill, device=dev)
        d.out, d.ld_out = out.data_ptr(), N
        if tiles is not None:
            d.tile_list, d.tile_count = tiles.data_ptr(), tiles.numel()
        _clx.call("clx_conv_fwd", ctypes.byref(d), st)
        torch.cuda.synchronize()
        return out
    a, b = run(), run()
    print(k, (B, H, W, C, N), "dense twice equal:", torch.equal(a, b))
    th, tw = -(-OH // 4), -(-OW // 4)
    allt = torch.arange(B * th * tw, dtype=torch.int32, device=dev)
    c = run(allt, -7.0)
    print("   full list == dense:", torch.equal(a, c))
    perm = allt[torch.randperm(allt.numel(), device=dev)]
    e = run(perm, -7.0)
    diff = (a != e)
    print("   permuted list == dense:", torch.equal(a, e), "mismatches", int(diff.sum()), "max abs", float((a - e).abs().max()))
    if diff.any():
        idx = diff.nonzero()
        tiles_bad = set(((i[0] * th + i[1] // 4) * tw + i[2] // 4).item() for i in idx)
        pos = {int(t): int((perm == t).nonzero()[0]) for t in tiles_bad}
        print("   bad tiles -> position in list:", sorted(pos.items(), key=lambda kv: kv[1])[:40])
        print("   channels:", sorted(set(int(i[3]) for i in idx))[:20])
