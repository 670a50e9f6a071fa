"""clx_ms_prepare_f32 alone at size^2 (kernel time by libclx's event stamps): python tools/exp/prepare_f32_time.py 8192"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from cellulus_amd import _clx  # noqa: E402

if os.environ.get("CLX_LIB"):
    _clx.LIB_PATH = os.path.abspath(os.environ["CLX_LIB"])
from bench_infer import PROF_KIND, _kernel_time, synthetic_embeddings  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
device = torch.device("cuda:0")
lib = _clx.load()
st = _clx.stream_ptr(device)
mean, std = synthetic_embeddings((512, 512), spacing=48, radius=12, noise=0.3, seed=1)
reps = size // 512
emb32 = torch.from_numpy(np.tile(mean[0], (1, reps, reps))).to(device).float()
sd32 = torch.from_numpy(np.tile(std, (reps, reps))).to(device).float()
npix = size * size
ws = torch.empty(int(lib.clx_ms_prepare_workspace(npix)), dtype=torch.uint8, device=device)
pts = torch.empty((npix, 2), dtype=torch.float64, device=device)
nfg = torch.zeros(1, dtype=torch.int32, device=device)
t = _kernel_time(lambda: _clx.call("clx_ms_prepare_f32", _clx.ptr(emb32), _clx.ptr(sd32), 0.5, 2, 1, size, size,
                                   _clx.ptr(pts), None, _clx.ptr(nfg), _clx.ptr(ws), st), PROF_KIND["ms_prepare"])
print(size, "kernel ms", round(t[1] * 1e3, 4), "nfg", int(nfg.item()))
