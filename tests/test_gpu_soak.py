"""GPU: nothing grows while the drivers run at benchmark size (promoted from tests/diag/soak_memory.py).

180 iterations (CLX_SOAK_ITERATIONS; 300 and more for a long soak) of the real ``train()`` at BASELINE configs[1] (loader processes with the default elastic
augmentation and the np.random pair stream, device prefetcher, logging) and two passes of ``infer()`` over a
512^2 container: device memory and host RSS after the start-up must be flat."""

import contextlib
import io
import os
import resource

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rss_mb():
    with open("/proc/self/statm") as fh:
        return int(fh.read().split()[1]) * resource.getpagesize() / 2 ** 20


def test_train_loop_holds_its_memory_at_benchmark_size(device, tmp_path, monkeypatch):
    import cellulus_amd.train as T
    from bench import synthetic_raw
    from cellulus_amd.configs import ExperimentConfig
    from cellulus_amd.utils import zarr_io

    monkeypatch.chdir(tmp_path)
    f = zarr_io.open("data.zarr")
    f["train/raw"] = np.concatenate([synthetic_raw(1, (384, 384), s).numpy() for s in range(8)])
    f["train/raw"].attrs["axis_names"] = ["s", "c", "y", "x"]
    iters = max(140, int(os.environ.get("CLX_SOAK_ITERATIONS", "180")))    # (under -m gpu: the suite has a time limit)
    cfg = ExperimentConfig(
        normalization_factor=1.0, model_config=dict(num_fmaps=256, fmap_inc_factor=3),
        train_config=dict(crop_size=[256, 256], batch_size=8, max_iterations=iters, num_workers=8,
                          save_model_every=10 ** 6, save_best_model_every=10 ** 6, save_snapshot_every=10 ** 6,
                          train_data_config=dict(container_path="data.zarr", dataset_name="train/raw")))
    assert cfg.train_config.elastic_deform is True          # the reference's default (train_config.py:124)
    rows = []
    real = T.train_iteration

    def spy(*a, **k):
        out = real(*a, **k)
        rows.append((torch.cuda.memory_allocated(device) / 2 ** 20, torch.cuda.memory_reserved(device) / 2 ** 20,
                     _rss_mb(), float(out[0])))
        return out

    monkeypatch.setattr(T, "train_iteration", spy)
    with contextlib.redirect_stdout(io.StringIO()):
        T.train(cfg)
    assert len(rows) == iters
    alloc, reserved, rss, loss = (np.array(c) for c in zip(*rows))
    # a reading may or may not include the prefetched next batch (40 MB): compare window maxima
    early, late = slice(60, 100), slice(iters - 40, iters)
    print(f"device MB allocated {alloc[early].max():.1f} -> {alloc[late].max():.1f}, reserved "
          f"{reserved[early].max():.1f} -> {reserved[late].max():.1f}, host RSS {rss[early].max():.1f} -> "
          f"{rss[late].max():.1f}; loss {loss[:10].mean():.4g} -> {loss[-10:].mean():.4g}")
    assert alloc[late].max() - alloc[early].max() < 1.0
    assert reserved[late].max() - reserved[early].max() < 64.0
    assert rss[late].max() - rss[early].max() < 128.0
    assert np.isfinite(loss).all() and loss[-10:].mean() < loss[:10].mean()


def test_infer_holds_its_memory(device, tmp_path, monkeypatch):
    from bench import synthetic_raw
    from cellulus_amd.configs import ExperimentConfig
    from cellulus_amd.infer import infer
    from cellulus_amd.models import get_model
    from cellulus_amd.utils import zarr_io

    monkeypatch.chdir(tmp_path)
    mcfg = dict(num_fmaps=32, fmap_inc_factor=3, features_in_last_layer=64, downsampling_factors=[[2, 2]])
    torch.manual_seed(0)
    model = get_model(in_channels=1, out_channels=2, num_spatial_dims=2, **mcfg)
    for layer in model.modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    os.makedirs("models")
    torch.save({"model_state_dict": model.state_dict()}, "models/best_loss.pth")
    readings = []
    for run in range(3):
        container = str(tmp_path / f"data{run}.zarr")
        f = zarr_io.open(container)
        f["test/raw"] = np.concatenate([synthetic_raw(1, (512, 512), s).numpy() for s in range(6)])
        f["test/raw"].attrs["axis_names"] = ["s", "c", "y", "x"]
        cfg = ExperimentConfig(
            model_config=dict(checkpoint="models/best_loss.pth", **mcfg), object_size=30, normalization_factor=1.0,
            inference_config=dict(
                dataset_config=dict(container_path=container, dataset_name="test/raw"),
                prediction_dataset_config=dict(container_path=container, dataset_name="embeddings"),
                detection_dataset_config=dict(container_path=container, dataset_name="detection",
                                              secondary_dataset_name="embeddings"),
                segmentation_dataset_config=dict(container_path=container, dataset_name="segmentation",
                                                 secondary_dataset_name="detection"),
                crop_size=[272, 272], num_infer_iterations=4, device="cuda:0"))
        with contextlib.redirect_stdout(io.StringIO()):
            infer(cfg)
        torch.cuda.synchronize()
        readings.append((torch.cuda.memory_allocated(device) / 2 ** 20, _rss_mb()))
    print("after each infer(): device MB allocated / host RSS MB:", readings)
    assert readings[2][0] - readings[1][0] < 1.0
    assert readings[2][1] - readings[1][1] < 64.0
