"""End-to-end rate of cellulus_amd.train.train() at the 2-D benchmark configuration WITH the input
pipeline (zarr reads, random crops, pair sampling in loader processes, H2D), against bench.py's
resident-input figure.  Usage: python tests/diag/loader_rate.py [iterations] [num_workers] [elastic 0|1]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellulus_amd.configs import ExperimentConfig  # noqa: E402
from cellulus_amd.train import train  # noqa: E402
from cellulus_amd.utils import zarr_io  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
elastic = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
with tempfile.TemporaryDirectory() as tmp:
    os.chdir(tmp)
    rng = np.random.default_rng(0)
    f = zarr_io.open(os.path.join(tmp, "data.zarr"))
    f["train/raw"] = rng.random((64, 1, 512, 512)).astype(np.float32)
    f["train/raw"].attrs["axis_names"] = ["s", "c", "y", "x"]
    cfg = ExperimentConfig(
        object_size=30, normalization_factor=1.0,
        model_config=dict(num_fmaps=256, fmap_inc_factor=3, downsampling_factors=[[2, 2]]),
        train_config=dict(
            train_data_config=dict(container_path=os.path.join(tmp, "data.zarr"), dataset_name="train/raw"),
            crop_size=[256, 256], batch_size=8, max_iterations=iters, num_workers=workers, elastic_deform=elastic,
            save_model_every=10 ** 9, save_snapshot_every=10 ** 9, save_best_model_every=10 ** 9, device="cuda:0"))
    t0 = time.perf_counter()
    train(cfg)
    dt = time.perf_counter() - t0
    print(f"train(): {iters} iterations of 8 crops in {dt:.1f} s incl. start-up = {iters * 8 / dt:.1f} crops/s "
          f"(workers {workers}, elastic {elastic})")
