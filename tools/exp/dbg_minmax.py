import sys; sys.path.insert(0, ".")
import numpy as np, torch
from cellulus_amd import _clx
dev = torch.device("cuda:0")
st = _clx.stream_ptr(dev)
for n in (25, 4096, 262144, 262147):
    x = (torch.rand(n, dtype=torch.float64, device=dev) * 0.9 + 0.05)
    mm = torch.full((_clx.MINMAX_DOUBLES,), -7.0, dtype=torch.float64, device=dev)
    _clx.call("clx_minmax_f64", _clx.ptr(x), n, _clx.ptr(mm), st)
    torch.cuda.synchronize()
    h = mm.cpu().numpy()
    written = np.nonzero(h != -7.0)[0]
    print(n, "result", h[:2], "true", x.min().item(), x.max().item(), "written slots", len(written), written[:12], h[2:8])
    from cellulus_amd.utils.otsu import histogram_on_device
    c, e = histogram_on_device(x)
    rc, re_ = np.histogram(x.cpu().numpy(), bins=256)
    print("   hist equal", np.array_equal(c, rc), np.array_equal(e, re_), int(c.sum()), n)
