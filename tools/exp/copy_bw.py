"""Practical HBM ceilings on this box: torch copy / in-place add / read-only reduction at a few sizes."""
import torch
dev = torch.device("cuda:0")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for mb in (128, 512, 2048):
    n = mb * 1024 * 1024 // 8
    a = torch.randn(n, dtype=torch.float64, device=dev); b = torch.empty_like(a)
    s = t(lambda: b.copy_(a)); print(f"{mb} MB copy      : {2 * n * 8 / s / 1e12:.2f} TB/s  ({s * 1e6:.0f} us)")
    s = t(lambda: a.add_(1.0)); print(f"{mb} MB add_ in place: {2 * n * 8 / s / 1e12:.2f} TB/s  ({s * 1e6:.0f} us)")
    s = t(lambda: a.sum()); print(f"{mb} MB sum       : {n * 8 / s / 1e12:.2f} TB/s  ({s * 1e6:.0f} us)")
    s = t(lambda: b.zero_()); print(f"{mb} MB fill      : {n * 8 / s / 1e12:.2f} TB/s  ({s * 1e6:.0f} us)")
