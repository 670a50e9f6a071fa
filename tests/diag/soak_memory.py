"""Leak check: train() at the benchmark configuration for N iterations (loader processes, device prefetcher,
snapshots / checkpoints at their default cadence switched off), then infer() over S samples; prints device memory
(allocated / reserved) and the host RSS of this process every `every` iterations / samples.
Usage: python tests/diag/soak_memory.py [iterations] [samples]"""
import contextlib
import io
import os
import resource
import shutil
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 64
tmp = tempfile.mkdtemp(prefix="clx_soak_")
os.chdir(tmp)
from bench import synthetic_raw  # noqa: E402
from cellulus_amd.utils import zarr_io  # noqa: E402

f = zarr_io.open("data.zarr")
f["train/raw"] = np.concatenate([synthetic_raw(1, (256, 256), s).numpy() for s in range(16)])
f["train/raw"].attrs["axis_names"] = ["s", "c", "y", "x"]
import cellulus_amd.train as T  # noqa: E402
from cellulus_amd.configs import ExperimentConfig  # noqa: E402

cfg = ExperimentConfig(normalization_factor=1.0, model_config=dict(num_fmaps=256, fmap_inc_factor=3),
                       train_config=dict(crop_size=[256, 256], batch_size=8, max_iterations=iters, num_workers=8,
                                         elastic_deform=True, save_model_every=10 ** 6, save_best_model_every=10 ** 6,
                                         save_snapshot_every=10 ** 6,
                                         train_data_config=dict(container_path="data.zarr", dataset_name="train/raw")))
rows = []
real = T.train_iteration
count = [0]


def rss_mb():
    with open("/proc/self/statm") as fh:
        return int(fh.read().split()[1]) * resource.getpagesize() / 2 ** 20


def spy(*a, **k):
    out = real(*a, **k)
    count[0] += 1
    if count[0] % max(1, iters // 10) == 0:
        rows.append((count[0], torch.cuda.memory_allocated() / 2 ** 20, torch.cuda.memory_reserved() / 2 ** 20, rss_mb()))
    return out


T.train_iteration = spy
with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
    T.train(cfg)
print("train(): iteration, device MB allocated, reserved, host RSS MB")
for r in rows:
    print("  %6d %10.1f %10.1f %10.1f" % r)
grow = rows[-1][1] - rows[1][1], rows[-1][3] - rows[1][3]
print(f"growth between the 2nd and the last reading: device {grow[0]:+.1f} MB, host {grow[1]:+.1f} MB")
shutil.rmtree(tmp, ignore_errors=True)
