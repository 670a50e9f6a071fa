"""Where the HOST side of train() spends an iteration at the benchmark configuration (8 loader processes): waiting for
the loader's next batch, enqueueing the step, logging.  python tools/exp/e2e_host_breakdown.py [workers=8] [iterations=140]"""
import contextlib
import io
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from cellulus_amd import train as T  # noqa: E402
from cellulus_amd.configs import ExperimentConfig  # noqa: E402
from cellulus_amd.utils import zarr_io  # noqa: E402

# what ran in this process before train(): bench.py's CPU legs change the loaders' environment (CLX_HB_PRE=sklearn|oracle)
pre = os.environ.get("CLX_HB_PRE", "")
if "sklearn" in pre:
    from sklearn.cluster import MeanShift

    MeanShift(bandwidth=15.0, cluster_all=False).fit(np.random.rand(2000, 2) * 200).predict(np.random.rand(50000, 2) * 200)
if "oracle" in pre:
    from bench_infer import synthetic_embeddings
    from oracle import infer_oracle as IO

    mean_, std_ = synthetic_embeddings((512, 512), spacing=48, radius=12, noise=0.3, seed=1)
    IO.mean_shift_segmentation(mean_.copy(), std_, 15.0, 70, 0.1, IO.threshold_otsu(std_), None)
if "threads" in pre:
    torch.set_num_threads(32)
    a = torch.randn(2048, 2048)
    (a @ a).sum()
workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iterations = int(sys.argv[2]) if len(sys.argv) > 2 else 140
wl = bench.WORKLOADS["train2d"]
crop = list(wl["crop"])
tmp = tempfile.mkdtemp(prefix="clx_hb_")
cwd = os.getcwd()
os.chdir(tmp)
f = zarr_io.open("data.zarr")
big = tuple(int(c * 1.5) for c in crop)
f["train/raw"] = np.concatenate([bench.synthetic_raw(1, big, s).numpy() for s in range(16)])
f["train/raw"].attrs["axis_names"] = ["s", "c", "y", "x"]
m = wl["model"]
cfg = ExperimentConfig(
    normalization_factor=1.0, object_size=30,
    model_config=dict(num_fmaps=m["num_fmaps"], fmap_inc_factor=m["fmap_inc_factor"],
                      features_in_last_layer=m["features_in_last_layer"],
                      downsampling_factors=[list(x) for x in m["downsampling_factors"]]),
    train_config=dict(crop_size=crop, batch_size=wl["batch"], max_iterations=iterations, num_workers=workers,
                      kappa=wl["kappa"], density=wl["density"], device="cuda:0",
                      save_model_every=10 ** 6, save_best_model_every=10 ** 6, save_snapshot_every=10 ** 6,
                      train_data_config=dict(container_path="data.zarr", dataset_name="train/raw")))
acc = {"stage": [], "step": [], "stamp": []}
real_stage, real_iter = T._DevicePrefetcher._stage, T.train_iteration


def stage(self):
    t = time.perf_counter()
    real_stage(self)
    acc["stage"].append(time.perf_counter() - t)


def step(*a, **k):
    t = time.perf_counter()
    out = real_iter(*a, **k)
    now = time.perf_counter()
    acc["step"].append(now - t)
    acc["stamp"].append(now)
    return out


T._DevicePrefetcher._stage = stage
T.train_iteration = step
try:
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        T.train(cfg)
finally:
    os.chdir(cwd)
    shutil.rmtree(tmp, ignore_errors=True)
s = 40
it = np.diff(acc["stamp"][s:])
st = np.array(acc["stage"][s + 1:len(it) + s + 1])
sp = np.array(acc["step"][s + 1:])
print(f"pre [{pre}] workers {workers}: iteration {it.mean() * 1e3:.2f} ms (p95 {np.percentile(it, 95) * 1e3:.2f}, max {it.max() * 1e3:.2f}) | "
      f"host in train_iteration {sp.mean() * 1e3:.2f} ms | in _stage (next batch + H2D enqueue) {st.mean() * 1e3:.2f} ms "
      f"(p95 {np.percentile(st, 95) * 1e3:.2f}, max {st.max() * 1e3:.2f}) | "
      f"rest of the loop (print, logger) {(it.mean() - sp.mean() - st.mean()) * 1e3:.2f} ms | host cores {os.cpu_count()}")
