"""The MFMA kernels of one inference tile, two ways: the rows of the kernel trace's per-tile table
(tools/infer_gaps.py --digest, argument 1) against what libclx's own launch events report (bench_infer.embed_stage, the numbers
behind infer.roofline.all_mfma_kernels).  Prints both sums and fails if they differ by more than 3 %."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CLX_INFER_STREAMS", "1")          # the pass the rooflines are timed in: every kernel alone on the device
import torch  # noqa: E402

from bench_infer import embed_stage  # noqa: E402
from cellulus_amd.models import get_model  # noqa: E402

MFMA = ("conv_igemm_kernel", "conv_wgrad_kernel", "gemm_sp_kernel", "gemm_sp2_kernel", "chain64_", "wino_fused_kernel", "wino_pre_kernel")
trace_us = 0.0
for line in open(sys.argv[1]):
    m = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if m and any(k in m.group(1) for k in MFMA):
        trace_us += float(m.group(3))
dev = torch.device("cuda:0")
cfg = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64,
           downsampling_factors=[[2, 2]], num_spatial_dims=2)
torch.manual_seed(0)
model = get_model(**cfg).to(dev)
for _n, layer in model.named_modules():
    if isinstance(layer, torch.nn.modules.conv._ConvNd):
        torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
model.eval()
model.set_infer(p_salt_pepper=0.01, num_infer_iterations=16, device=dev)
_t, _emb, prof = embed_stage(model, dev, 512, 16, 2)
events_ms = sum(v[1] for v in prof.values())
print(f"MFMA kernels of one tile: kernel trace {trace_us / 1e3:.2f} ms, libclx launch events {events_ms:.2f} ms "
      f"(ratio {events_ms / (trace_us / 1e3):.3f})")
for k, v in prof.items():
    if v[0]:
        print(f"    {k:36s} {int(round(v[0])):4d} launches {v[1]:8.2f} ms per tile")
assert abs(events_ms / (trace_us / 1e3) - 1.0) < 0.03, "infer.roofline's MFMA time disagrees with the kernel trace"
