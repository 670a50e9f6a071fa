"""One wide layer, realistic operands (post-ReLU input, Kaiming weights): max |y - f64| / max|y| of
the HIP direct kernel, HIP F(4x4) (three launches and fused), and PyTorch-CPU float32 (oneDNN)."""
import ctypes, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellulus_amd import _clx
from cellulus_amd._clx import ClxConvDesc, ClxSrc
dev = torch.device("cuda:0"); st = _clx.stream_ptr(dev); lib = _clx.load()
for (C, N, k, H) in [(768, 768, 3, 40), (256, 256, 3, 64), (768, 768, 1, 64), (256, 256, 1, 96), (1024, 64, 3, 48)]:
    torch.manual_seed(C + N + k)
    x = torch.relu(torch.randn(1, H, H, C))
    w = torch.randn(N, C, k, k) * (2.0 / (C * k * k)) ** 0.5
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double()).permute(0, 2, 3, 1)
    cpu = F.conv2d(x.permute(0, 3, 1, 2), w).permute(0, 2, 3, 1)
    scale = ref.abs().max().item()
    res = {"cpu f32": (cpu.double() - ref).abs().max().item() / scale}
    rms = {"cpu f32": ((cpu.double() - ref) ** 2).mean().sqrt().item() / scale}
    xd = x.to(dev).contiguous(); wd = w.reshape(N, C, k * k).to(dev).contiguous()
    OH = H - k + 1
    def run(algo, mode):
        nx = 36 if algo else k * k
        wp = torch.empty(nx * N * C, device=dev)
        _clx.call("clx_pack_weights", _clx.ptr(wd), _clx.ptr(wp), N, C, k * k, C, N, mode, st)
        d = ClxConvDesc(); d.nsrc = 1
        s = ClxSrc(); s.ptr, s.C, s.ld = xd.data_ptr(), C, C; s.D, s.H, s.W = 1, H, H; s.fz = s.fy = s.fx = 1
        d.src[0] = s; d.B = 1; d.ID, d.IH, d.IW = 1, H, H; d.KD, d.KH, d.KW = 1, k, k; d.N = N; d.algo = algo
        d.wpack = wp.data_ptr()
        if algo == 2:
            need = int(lib.clx_conv_workspace_bytes(ctypes.byref(d), 0)); ws = torch.empty(need // 4 + 4, device=dev)
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        out = torch.empty(1, OH, OH, N, device=dev); d.out, d.ld_out = out.data_ptr(), N
        _clx.call("clx_conv_fwd", ctypes.byref(d), st); torch.cuda.synchronize()
        return out.cpu().double()
    todo = [("hip direct", 0, 0)] + ([("hip F(4x4) 3 launches", 2, 4), ("hip F(4x4) fused", 3, 7)] if k == 3 and N % 64 == 0 else [])
    for name, algo, mode in todo:
        o = run(algo, mode)
        res[name] = (o - ref).abs().max().item() / scale; rms[name] = ((o - ref) ** 2).mean().sqrt().item() / scale
    print(f"C={C} N={N} k={k} K={C*k*k}: " + "  ".join(f"{n}: max {res[n]:.2e} rms {rms[n]:.2e}" for n in res))
