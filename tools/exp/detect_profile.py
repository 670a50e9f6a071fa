"""Host-side profile of the detect stage on one synthetic 512^2 sample (cProfile: where the ~1 ms goes)."""
import cProfile
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from bench_infer import synthetic_embeddings  # noqa: E402
from cellulus_amd.utils.mean_shift import mean_shift_on_device  # noqa: E402
from cellulus_amd.utils.otsu import threshold_otsu  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
device = torch.device("cuda:0")
mean, std = synthetic_embeddings((size, size), spacing=48, radius=12, noise=0.3, seed=1)
mean_d = torch.from_numpy(mean[0]).to(device)
std_d = torch.from_numpy(std).to(device)


def once():
    np.random.seed(1)
    thr = threshold_otsu(std_d)
    return mean_shift_on_device(mean_d.clone(), std_d, 15.0, 0.1, thr, None)


for _ in range(5):
    once()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    once()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
