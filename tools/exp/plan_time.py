import sys, time; sys.path.insert(0, ".")
import torch
from cellulus_amd.models import get_model
from cellulus_amd.models.plan import UNetPlan, build_topology
dev = torch.device("cuda:0")
cfg = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64, downsampling_factors=[[2, 2]], num_spatial_dims=2)
topo = build_topology(**cfg, spatial=(528, 528))
torch.zeros(1, device=dev); torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter(); p = UNetPlan(topo, 8, dev, False); torch.cuda.synchronize(); t1 = time.perf_counter()
    nbytes = sum(t.numel() * t.element_size() for t in p.buf.values())
    print(f"plan {i}: {1e3 * (t1 - t0):.1f} ms, buffers {nbytes / 2**30:.1f} GiB, workspace {0 if p.workspace is None else p.workspace.numel() * 4 / 2**30:.1f} GiB")
    t0 = time.perf_counter(); x = torch.empty(nbytes // 4, dtype=torch.float32, device=dev); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"   torch.empty of the same size: {1e3 * (t1 - t0):.1f} ms")
    t0 = time.perf_counter(); x.zero_(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"   zero_: {1e3 * (t1 - t0):.1f} ms")
    del x
