"""Per-parameter gradient error at BASELINE cfg-2 / cfg-4 (batch 1) against the float64 oracle:
HIP default plan, HIP direct kernels (CLX_WINOGRAD=0), and the float32 CPU oracle itself.
Usage: python tests/diag/diag_fullsize_grads.py [cfg2|cfg4] [mode]   (mode: default|direct|cpu32)"""
import os
import sys

sys.path.insert(0, ".")
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
mode = sys.argv[2] if len(sys.argv) > 2 else "default"
if mode == "direct":
    os.environ["CLX_WINOGRAD"] = "0"
if mode == "wino2":
    os.environ["CLX_WINOGRAD_TILE"] = "2"
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import unet_oracle as O  # noqa: E402
from tests.test_gpu_fullsize_oracle import CFG2, CFG4, _blobs, _kaiming  # noqa: E402

cfg, crop = (CFG2, (256, 256)) if which == "cfg2" else (CFG4, (64, 64, 64))
torch.manual_seed(0)
oracle = O.OracleUNetModel(**cfg)
_kaiming(oracle)
raw = _blobs(crop, seed=3)
o64 = O.OracleUNetModel(**cfg).double()
o64.load_state_dict({k: v.double() for k, v in oracle.state_dict().items()})
with O.gemm_convolutions(o64):
    ref64 = o64(raw.double())
    torch.manual_seed(2)
    dout = torch.randn(ref64.shape)
    if os.environ.get("DIAG_LOSS") == "oce":
        pass
    ref64.backward(dout.double())
if mode == "cpu32":
    out = oracle(raw)
    out.backward(dout)
    grads = [p.grad.double() for p in oracle.parameters()]
else:
    from cellulus_amd.models import get_model

    dev = torch.device("cuda:0")
    model = get_model(**cfg)
    model.load_state_dict(oracle.state_dict(), strict=True)
    model = model.to(dev)
    out = model(raw.to(dev))
    out.backward(dout.to(dev))
    grads = [p.grad.cpu().double() for p in model.parameters()]
print(f"{which} {mode}: |out - f64| {(out.detach().cpu().double() - ref64.detach()).abs().max().item():.2e}")
for (n, po), g in zip(o64.named_parameters(), grads):
    gr = po.grad
    l2 = ((g - gr).norm() / (gr.norm() + 1e-30)).item()
    print(f"  {n:45s} |g| {gr.norm().item():10.3e}  rel L2 {l2:.2e}  max rel {(g - gr).abs().max().item() / gr.abs().max().item():.2e}")
