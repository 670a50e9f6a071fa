"""Full-size (BASELINE cfg-2) fused train steps on ONE rank of the real RCCL backend with the
multi-rank code path forced: cost of issuing the gradient buckets on RCCL's stream between the
backward kernels (CLX_GRAD_BUCKET_MB = 4 default, 0 = one all-reduce after the backward pass)
against the plain single-rank step.  Usage: python tests/diag/diag_rccl_overlap.py"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellulus_amd import parallel  # noqa: E402
from cellulus_amd.criterions import get_loss  # noqa: E402
from cellulus_amd.models import get_model  # noqa: E402
from cellulus_amd.optim import Adam  # noqa: E402
from cellulus_amd.train import _fused_step  # noqa: E402

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29631"), WORLD_SIZE="1",
                  RANK="0", LOCAL_RANK="0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
cfg = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64,
           downsampling_factors=[[2, 2]], num_spatial_dims=2)
rng = np.random.default_rng(0)
B, out, kappa = 8, 240, 10
raw = torch.rand(B, 1, 256, 256, device=dev)
a = np.repeat(rng.integers(kappa, out - kappa + 1, size=(B, 4840, 2)), 31, axis=1)
o = rng.integers(-kappa + 1, kappa, size=a.shape)
o[np.abs(o).sum(-1) == 0] = 1
anchor = torch.from_numpy(a.astype(np.int64)).to(dev)
reference = torch.from_numpy((a + o).astype(np.int64)).to(dev)
real_world = parallel.world_size
for label, world, mb in (("single rank", 1, "4"), ("buckets 4 MB", 2, "4"), ("one all-reduce at the end", 2, "0"),
                         ("buckets 1 MB", 2, "1")):
    os.environ["CLX_GRAD_BUCKET_MB"] = mb
    parallel.world_size = (lambda: 2) if world == 2 else real_world
    torch.manual_seed(0)
    model = get_model(**cfg).to(dev)
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=2, device=dev)
    opt = Adam(model.parameters(), lr=4e-5, weight_decay=0.01)
    for _ in range(3):
        _fused_step(model, crit, opt, raw, anchor, reference)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        loss = _fused_step(model, crit, opt, raw, anchor, reference)[0]
    torch.cuda.synchronize()
    print(f"{label:28s} {(time.perf_counter() - t0) / 8 * 1e3:7.2f} ms/step  loss {loss:.3f}", flush=True)
    del model, opt
dist.destroy_process_group()
