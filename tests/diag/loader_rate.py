"""Loader-inclusive training rate at the benchmark configuration: train() on a synthetic zarr (zarr
reads, random crops, pair sampling in the loader processes — or on the device with CLX_DEVICE_PAIRS=1 —
H2D, logging).  Usage: [CLX_DEVICE_PAIRS=1] python tests/diag/loader_rate.py [iterations] [workers] [elastic: 0|1]"""
import contextlib
import io
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
elastic = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
tmp = tempfile.mkdtemp(prefix="clx_lr_")
os.chdir(tmp)
from bench import synthetic_raw  # noqa: E402
from cellulus_amd.utils import zarr_io  # noqa: E402

f = zarr_io.open("data.zarr")
f["train/raw"] = np.concatenate([synthetic_raw(1, (256, 256), s).numpy() for s in range(16)])
f["train/raw"].attrs["axis_names"] = ["s", "c", "y", "x"]
import cellulus_amd.train as T  # noqa: E402
from cellulus_amd.configs import ExperimentConfig  # noqa: E402

cfg = ExperimentConfig(normalization_factor=1.0, model_config=dict(num_fmaps=256, fmap_inc_factor=3),
                       train_config=dict(crop_size=[256, 256], batch_size=8, max_iterations=iters, num_workers=workers,
                                         elastic_deform=elastic, save_model_every=10 ** 6, save_best_model_every=10 ** 6,
                                         save_snapshot_every=10 ** 6,
                                         train_data_config=dict(container_path="data.zarr", dataset_name="train/raw")))
stamps = []
real = T.train_iteration


def spy(*a, **k):
    out = real(*a, **k)
    stamps.append(time.perf_counter())
    return out


T.train_iteration = spy
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
    T.train(cfg)
dt = time.perf_counter() - t0
steady = (stamps[-1] - stamps[50]) / (len(stamps) - 51)
print(f"CLX_DEVICE_PAIRS={os.environ.get('CLX_DEVICE_PAIRS', '0')} workers={workers} elastic={int(elastic)}: {iters} iterations in {dt:.1f} s "
      f"({iters * 8 / dt:.1f} crops/s incl. start-up); steady state {steady * 1e3:.1f} ms per iteration = "
      f"{8 / steady:.1f} crops/s")
shutil.rmtree(tmp, ignore_errors=True)
