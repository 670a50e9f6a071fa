"""Generates tests/golden/*.npz by running the REAL reference code in the build
container (it cannot travel to the GPU box):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        python tests/golden/make_golden.py

Importable reference modules: cellulus.criterions.oce_loss,
cellulus.utils.mean_shift (+ installed scikit-learn), cellulus.configs, and
cellulus.models.unet once `funlib.learn.torch.models.UNet` is stubbed with the
oracle's backbone restatement (the head, noise loop, std_mean and gather are
then genuine reference code).  The fixtures hold inputs (or the seeds that
regenerate them) and the reference's outputs — data only.
"""

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import infer_oracle as IO  # noqa: E402  (synthetic input generator only)
from oracle.unet_oracle import OracleUNet  # noqa: E402

# ---- stub the un-vendored backbone so cellulus.models.unet imports
funlib = types.ModuleType("funlib")
learn = types.ModuleType("funlib.learn")
ftorch = types.ModuleType("funlib.learn.torch")
models = types.ModuleType("funlib.learn.torch.models")
models.UNet = OracleUNet
sys.modules.update({"funlib": funlib, "funlib.learn": learn, "funlib.learn.torch": ftorch,
                    "funlib.learn.torch.models": models})

from cellulus.configs import ExperimentConfig  # noqa: E402
from cellulus.criterions import get_loss  # noqa: E402
from cellulus.models.unet import UNetModel  # noqa: E402
from cellulus.utils.mean_shift import mean_shift_segmentation  # noqa: E402


def g1_oce():
    out = {}
    for nd in (2, 3):
        torch.manual_seed(nd)
        a = (torch.randn(3, 200, nd) * 4)
        r = a + torch.randn(3, 200, nd) * 3
        a[0, 0] = 0.0
        r[0, 1] = a[0, 1]
        a.requires_grad_(True)
        crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1,
                        num_spatial_dims=nd, device=torch.device("cpu"))
        loss, oce, reg = crit(a, r)
        loss.backward()
        out[f"a{nd}"] = a.detach().numpy()
        out[f"r{nd}"] = r.numpy()
        out[f"sums{nd}"] = np.array([loss.item(), oce.item(), reg.item()], dtype=np.float64)
        out[f"grad{nd}"] = a.grad.numpy()
    # known-answer vector of SURVEY.md §8c
    a = torch.tensor([[[3.0, 4.0], [1.0, 1.0], [0.0, 0.0]]], requires_grad=True)
    r = torch.tensor([[[0.0, 0.0], [1.0, 1.0], [2.0, 0.0]]])
    crit = get_loss(10.0, 1e-5, 0.1, 2, torch.device("cpu"))
    loss, oce, reg = crit(a, r)
    loss.backward()
    out["kat_sums"] = np.array([loss.item(), oce.item(), reg.item()])
    out["kat_grad"] = a.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g1_oce_loss.npz"), **out)


def g2_gather():
    out = {}
    rng = np.random.default_rng(0)
    for nd, shape in ((2, (9, 11)), (3, (5, 6, 7))):
        offsets = torch.from_numpy(rng.standard_normal((2, nd) + shape).astype(np.float32))
        coords = np.stack([rng.integers(0, s, size=(2, 50)) for s in shape[::-1]], axis=2).astype(np.int64)
        sel = UNetModel.select_and_add_coordinates(offsets, torch.from_numpy(coords))
        out[f"offsets{nd}"] = offsets.numpy()
        out[f"coords{nd}"] = coords
        out[f"sel{nd}"] = sel.numpy()
    np.savez_compressed(os.path.join(HERE, "g2_gather.npz"), **out)


def g3_unet():
    """Reference UNetModel wrapper (real head / infer loop) around the stub backbone."""
    out = {}
    for nd, spatial in ((2, (36, 40)), (3, (20, 20, 24))):
        cfg = dict(in_channels=1, out_channels=nd, num_fmaps=4, fmap_inc_factor=2,
                   features_in_last_layer=8, downsampling_factors=[(2,) * nd], num_spatial_dims=nd)
        torch.manual_seed(100 + nd)
        model = UNetModel(**cfg)
        for _n, layer in model.named_modules():
            if isinstance(layer, torch.nn.modules.conv._ConvNd):
                torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
        raw = torch.rand(2, 1, *spatial)
        with torch.no_grad():
            train_out = model(raw)
        model.set_infer(p_salt_pepper=0.05, num_infer_iterations=2, device=torch.device("cpu"))
        torch.manual_seed(7)
        with torch.no_grad():
            infer_out = model(raw)
        for k, v in model.state_dict().items():
            out[f"w{nd}/{k}"] = v.numpy()
        out[f"raw{nd}"] = raw.numpy()
        out[f"train{nd}"] = train_out.numpy()
        out[f"infer{nd}"] = infer_out.numpy()
    np.savez_compressed(os.path.join(HERE, "g3_unet.npz"), **out)


def g4_mean_shift():
    """Real reference mean_shift_segmentation (+ sklearn) on synthetic disc/ball embeddings."""
    out = {}
    cases = [
        ("2d_rp1", (96, 96), dict(spacing=32, radius=9), 1.0, None, 10.0),
        ("2d_rp05", (96, 128), dict(spacing=32, radius=9), 0.5, None, 10.0),
        ("2d_rp02", (144, 144), dict(spacing=48, radius=12), 0.2, None, 15.0),
        ("2d_seeds", (96, 96), dict(spacing=32, radius=9), 0.5, "grid", 10.0),
        ("3d_rp05", (24, 40, 40), dict(spacing=20, radius=6), 0.5, None, 7.0),
        ("2d_empty", (32, 32), dict(spacing=64, radius=0), 0.5, None, 10.0),
    ]
    for name, shape, kw, rp, seeds, bw in cases:
        mean, std = IO.synthetic_embeddings(shape, seed=3, **kw)
        if name == "2d_empty":
            std[:] = 1.0
        if seeds == "grid":
            nd = len(shape)
            seeds = np.stack(np.meshgrid(*[np.arange(16, s, 32) for s in shape[::-1]], indexing="ij"),
                             -1).reshape(-1, nd)      # (x, y) order, integers
        np.random.seed(11)
        m = mean.copy()
        labels = mean_shift_segmentation(m, std, bandwidth=bw, min_size=10,
                                         reduction_probability=rp, threshold=0.5, seeds=seeds)
        out[f"{name}/mean"] = mean
        out[f"{name}/std"] = std
        out[f"{name}/mean_after"] = m
        out[f"{name}/labels"] = labels
        out[f"{name}/params"] = np.array([bw, rp, 0.5, 11])
        if seeds is not None:
            out[f"{name}/seeds"] = seeds
    np.savez_compressed(os.path.join(HERE, "g4_mean_shift.npz"), **out)


def g7_configs():
    import tomli

    train_toml = b"""
[model_config]
num_fmaps = 256
fmap_inc_factor = 3
downsampling_factors = [[2,2],]

[train_config.train_data_config]
container_path = "skin.zarr"
dataset_name = "train/raw"
"""
    infer_toml = b"""
[model_config]
num_fmaps = 256
fmap_inc_factor = 3
checkpoint = "models/best_loss.pth"

[inference_config.dataset_config]
container_path = "skin.zarr"
dataset_name = "test/raw"

[inference_config.prediction_dataset_config]
container_path = "skin.zarr"
dataset_name = "embeddings"

[inference_config.detection_dataset_config]
container_path = "skin.zarr"
dataset_name = "detection"
secondary_dataset_name = "embeddings"

[inference_config.segmentation_dataset_config]
container_path = "skin.zarr"
dataset_name = "segmentation"
secondary_dataset_name = "detection"
"""
    reprs = []
    for t in (train_toml, infer_toml):
        cfg = ExperimentConfig(experiment_name="golden", **tomli.loads(t.decode()))
        reprs.append(repr(cfg))
    np.savez_compressed(os.path.join(HERE, "g7_configs.npz"), train_toml=np.frombuffer(train_toml, dtype=np.uint8),
                        infer_toml=np.frombuffer(infer_toml, dtype=np.uint8), reprs=np.array(reprs))




def g6_greedy():
    """Real reference Cluster2d / Cluster3d (device="cpu") on synthetic disc/ball embeddings."""
    from cellulus.utils.greedy_cluster import Cluster2d, Cluster3d

    out = {}
    for name, shape, kw, bw in (("2d", (96, 128), dict(spacing=32, radius=9), 6.0),
                                ("3d", (24, 40, 40), dict(spacing=20, radius=6), 4.0)):
        nd = len(shape)
        mean, std = IO.synthetic_embeddings(shape, seed=5, **kw)
        rng = np.random.RandomState(2)
        std = std + rng.uniform(0, 0.05, size=std.shape)        # distinct seed scores
        pred = np.concatenate([mean[0], std[None]], 0)           # float64, like the zarr data
        fg = std < 0.5
        if nd == 2:
            seg = Cluster2d(width=shape[1], height=shape[0], fg_mask=fg, device="cpu").cluster(
                prediction=pred, bandwidth=bw, min_object_size=20)
        else:
            seg = Cluster3d(width=shape[2], height=shape[1], depth=shape[0], fg_mask=fg, device="cpu").cluster(
                prediction=pred, bandwidth=bw, min_object_size=20)
        out[f"{name}/pred"] = pred
        out[f"{name}/fg"] = fg
        out[f"{name}/seg"] = seg.numpy()
        out[f"{name}/params"] = np.array([bw, 20])
    np.savez_compressed(os.path.join(HERE, "g6_greedy.npz"), **out)


def g9_train_iteration():
    """REAL cellulus.train.train_iteration (train.py:160-180) with the real get_model / get_loss
    and the optimizer exactly as train.py:80-82 builds it, four iterations on fresh batches.
    `zarr` and `gunpowder` (imported at module level by cellulus.train / cellulus.datasets,
    never touched by train_iteration) are empty stub modules; the backbone is the stub above."""
    for name in ("zarr", "gunpowder"):
        sys.modules.setdefault(name, types.ModuleType(name))
    from cellulus.models import get_model
    from cellulus.train import train_iteration

    out = {}
    lr = 1e-3
    for nd, spatial, npairs in ((2, (36, 40), 60), (3, (20, 20, 24), 40)):
        torch.manual_seed(200 + nd)
        model = get_model(in_channels=1, out_channels=nd, num_fmaps=4, fmap_inc_factor=2,
                          features_in_last_layer=8, downsampling_factors=[(2,) * nd], num_spatial_dims=nd)
        for _n, layer in model.named_modules():                       # train.py:65-68
            if isinstance(layer, torch.nn.modules.conv._ConvNd):
                torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
        criterion = get_loss(regularizer_weight=1e-5, temperature=10.0, density=0.1,
                             num_spatial_dims=nd, device=torch.device("cpu"))
        optimizer = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0.01)   # train.py:80-82
        for k, v in model.state_dict().items():
            out[f"w{nd}/init/{k}"] = v.numpy().copy()
        out_shape = tuple(s - 16 for s in spatial)
        rng = np.random.RandomState(300 + nd)
        losses = []
        for it in range(4):
            raw = torch.from_numpy(rng.rand(2, 1, *spatial).astype(np.float32))
            # coordinates (x, y[, z]) order, column 0 indexes the last axis (unet.py:113-118)
            hi = np.array(out_shape[::-1])
            anchor = torch.from_numpy(rng.randint(0, hi, size=(2, npairs, nd)).astype(np.int64))
            reference = torch.from_numpy(rng.randint(0, hi, size=(2, npairs, nd)).astype(np.int64))
            loss, oce, offsets = train_iteration((raw, anchor, reference), model, criterion, optimizer,
                                                 torch.device("cpu"))
            losses.append([loss, oce])
            out[f"b{nd}/{it}/raw"] = raw.numpy()
            out[f"b{nd}/{it}/anchor"] = anchor.numpy()
            out[f"b{nd}/{it}/reference"] = reference.numpy()
            out[f"b{nd}/{it}/offsets"] = offsets.detach().numpy().copy()
        out[f"losses{nd}"] = np.array(losses, dtype=np.float64)
        for k, v in model.state_dict().items():
            out[f"w{nd}/final/{k}"] = v.numpy().copy()
    out["lr"] = np.array(lr)
    np.savez_compressed(os.path.join(HERE, "g9_train_iteration.npz"), **out)


def g10_evaluate():
    """REAL cellulus.evaluate.compute_pairwise_IoU / compute_F1 (evaluate.py:72-105) on seeded label
    images (zarr / gunpowder, imported at module level only, are empty stubs)."""
    for name in ("zarr", "gunpowder"):
        sys.modules.setdefault(name, types.ModuleType(name))
    from cellulus.evaluate import compute_F1, compute_pairwise_IoU

    out = {}
    rng = np.random.RandomState(5)
    for i, shape in enumerate([(48, 56), (40, 40), (16, 16, 24)]):
        gt = np.kron(rng.randint(0, 6, size=tuple(s // 8 for s in shape)), np.ones((8,) * len(shape), dtype=np.int64))
        pred = np.roll(gt, 2, axis=-1) * (rng.rand(*shape) < 0.9)
        pred[pred == 3] = 9                        # relabelled instance
        pred[tuple(slice(0, 6) for _ in shape)] = 11     # a false positive
        gt, pred = gt.astype(np.uint16), pred.astype(np.uint16)
        iou, seg, n = compute_pairwise_IoU(pred, gt)
        f1, tp, fp, fn = compute_F1(iou)
        out[f"{i}/gt"], out[f"{i}/pred"] = gt, pred
        out[f"{i}/iou"] = iou
        out[f"{i}/scalars"] = np.array([seg, n, f1, tp, fp, fn], dtype=np.float64)
    out["none_for_empty_gt"] = np.array(compute_pairwise_IoU(pred, np.zeros_like(gt)) is None)
    np.savez_compressed(os.path.join(HERE, "g10_evaluate.npz"), **out)


def g11_pair_sampler():
    """REAL ZarrDataset.sample_coordinates / sample_offsets_within_radius (zarr_dataset.py:163-248)
    called on an instance built without __init__ (which would need gunpowder), np.random seeded."""
    for name in ("zarr", "gunpowder"):
        sys.modules.setdefault(name, types.ModuleType(name))
    from cellulus.datasets.zarr_dataset import ZarrDataset

    out = {}
    for tag, nd, output_shape, kappa, density, seed in (("2d", 2, (60, 72), 10.0, 0.1, 3), ("3d", 3, (24, 28, 32), 6.0, 0.2, 4),
                                                         ("2d_small", 2, (30, 30), 3.0, 0.3, 5)):
        ds = object.__new__(ZarrDataset)
        ds.num_spatial_dims, ds.kappa, ds.density = nd, kappa, density
        ds.output_shape = output_shape
        ds.unbiased_shape = tuple(int(s - 2 * kappa) for s in output_shape)
        np.random.seed(seed)
        anchors, references = ds.sample_coordinates()
        out[f"{tag}/params"] = np.array([nd, kappa, density, seed], dtype=np.float64)
        out[f"{tag}/output_shape"] = np.array(output_shape)
        out[f"{tag}/anchors"], out[f"{tag}/references"] = anchors, references
        out[f"{tag}/counts"] = np.array([ds.get_num_anchors(), ds.get_num_references(), ds.get_num_samples()])
    np.savez_compressed(os.path.join(HERE, "g11_pair_sampler.npz"), **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    if only:
        for fn in only:
            globals()[fn]()
        sys.exit(0)
    g9_train_iteration()
    g10_evaluate()
    g11_pair_sampler()
    g1_oce()
    g2_gather()
    g3_unet()
    g4_mean_shift()
    g6_greedy()
    g7_configs()
    print("golden vectors written to", HERE)
