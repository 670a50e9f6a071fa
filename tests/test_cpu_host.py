"""CPU tier: host-side logic of cellulus_amd (no kernel is launched): configs, zarr IO,
topology, pair sampler, centre de-duplication, evaluation, the C-ABI symbol table,
loud failure without a HIP device, and the data-parallel wiring over gloo."""

import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------ C ABI
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "clx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(clx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cellulus_amd import _build, _clx

    _build.build()
    lib = ctypes.CDLL(_clx.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/clx.h but not exported"
    assert sorted(_clx.PROTOTYPES) == declared, "ctypes prototypes and clx.h disagree"
    assert _clx.load().clx_abi_version() == 13


def test_argument_validation_without_gpu():
    """Validation happens before any launch, so it is testable on the CPU."""
    from cellulus_amd import _clx

    lib = _clx.load()
    d = _clx.ClxConvDesc()
    d.nsrc = 3
    assert lib.clx_conv_fwd(ctypes.byref(d), None) == -1
    assert b"nsrc" in lib.clx_last_error()
    with pytest.raises(_clx.ClxError, match="nsrc"):
        _clx.call("clx_conv_fwd", ctypes.byref(d), None)
    assert lib.clx_maxpool_fwd(None, None, 1, 1, 4, 4, 4, 1, 2, 2, None) == -1
    assert lib.clx_adam_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) == -1
    # the row / tile movers of the infer-mode prefix
    assert lib.clx_changed_rows(None, None, 4, 1, 1, 16, 16, 1, 3, 3, 2, None, None, 8, None, None) == -1
    assert b"null pointer" in lib.clx_last_error()
    assert lib.clx_changed_tiles(None, 4, 1, 16, 16, 1, 3, 3, 3, 3, 4, 2, None, None, 8, None) == -1
    assert lib.clx_gather_rows(None, 8, None, 0, 8, None, 8, None) == 0          # nothing to move
    assert lib.clx_gather_rows(None, 8, None, 3, 8, None, 8, None) == -1
    assert lib.clx_scatter_rows(None, 8, None, -1, 8, None, 8, None) == -1
    assert lib.clx_broadcast_rows(None, 16, None, 2, None) == -1
    assert lib.clx_grey_rows(None, 1, 1, 16, 16, 1, None, 0, None, None, 1, 8, None, 8, None) == 0
    assert lib.clx_grey_rows(None, 1, 1, 16, 16, 1, None, 5, None, None, 1, 8, None, 8, None) == -1
    assert lib.clx_changed_rows_workspace(4, 1, 16, 130) == 2 * 4 * 16 * 3 * 8


def test_product_fails_loudly_without_hip_device():
    from cellulus_amd._clx import ClxError
    from cellulus_amd.models import get_model
    from cellulus_amd.utils.misc import size_filter

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    model = get_model(1, 2, 8, 3, 16, [[2, 2]], 2)
    with pytest.raises(ClxError, match="no CPU path"):
        model(torch.zeros(1, 1, 44, 44))
    with pytest.raises(ClxError, match="no CPU path"):
        size_filter(np.ones((4, 4), dtype=np.int32), 2)
    from cellulus_amd.train import _require_hip_device

    with pytest.raises(RuntimeError, match="no CPU path"):
        _require_hip_device("cpu")


def test_product_never_imports_oracle():
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "cellulus_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


# ----------------------------------------------------------------- configs
def test_configs_match_reference_repr_and_validators():
    import tomli

    from cellulus_amd.configs import ExperimentConfig

    g = np.load(os.path.join(G, "g7_configs.npz"))
    for key, ref in zip(("train_toml", "infer_toml"), g["reprs"]):
        cfg = ExperimentConfig(experiment_name="golden", **tomli.loads(bytes(g[key]).decode()))
        assert repr(cfg) == str(ref)
    with pytest.raises(TypeError):      # object_size must be an int (tests/train.toml:2 trips this)
        ExperimentConfig(model_config=dict(num_fmaps=8, fmap_inc_factor=2), object_size=10.0)
    with pytest.raises(ValueError):
        ExperimentConfig(model_config=dict(num_fmaps=8, fmap_inc_factor=2),
                         inference_config=dict(clustering="kmeans"))
    with pytest.raises(TypeError):
        ExperimentConfig(model_config=dict(num_fmaps=8.0, fmap_inc_factor=2))


# -------------------------------------------------------------------- zarr
def test_zarr_roundtrip_and_metadata(tmp_path):
    from cellulus_amd.configs import DatasetConfig
    from cellulus_amd.datasets import DatasetMetaData
    from cellulus_amd.utils import zarr_io

    f = zarr_io.open(tmp_path / "c.zarr")
    a = np.random.default_rng(0).random((3, 2, 10, 12, 14)).astype(np.float32)
    f["train/raw"] = a
    f["train/raw"].attrs["axis_names"] = ["s", "c", "z", "y", "x"]
    m = DatasetMetaData.from_dataset_config(DatasetConfig(container_path=tmp_path / "c.zarr", dataset_name="train/raw"))
    assert (m.num_samples, m.num_channels, m.num_spatial_dims, m.spatial_array) == (3, 2, 3, (10, 12, 14))
    np.testing.assert_array_equal(zarr_io.open(tmp_path / "c.zarr", "r")["train/raw"][1, :, 2:5], a[1, :, 2:5])
    ds = f.create_dataset("u16", shape=(2, 1, 9, 9), dtype=np.uint16, chunks=(1, 1, 4, 4),
                          compressor={"id": "gzip", "level": 1})
    ds[1, 0, 2:7, 3:9] = np.arange(30).reshape(5, 6)
    assert ds[1, 0, 6, 8] == 29 and ds[0].sum() == 0
    # zarr-python semantics: create_dataset refuses an existing path, group[name] = value replaces
    with pytest.raises(zarr_io.ContainsArrayError):
        f.create_dataset("u16", shape=(2, 1, 9, 9), dtype=np.uint16)
    assert f["u16"][1, 0, 6, 8] == 29
    assert f.create_dataset("u16", shape=(1, 1, 3, 3), dtype=np.uint16, overwrite=True).shape == (1, 1, 3, 3)
    f["u16"] = np.ones((2, 2), dtype=np.uint8)
    assert f["u16"].dtype == np.uint8
    del f["u16"]
    assert "u16" not in f
    f["noattr"] = np.zeros((1, 1, 4, 4))
    with pytest.raises(RuntimeError, match="axis_names"):
        DatasetMetaData.from_dataset_config(DatasetConfig(container_path=tmp_path / "c.zarr", dataset_name="noattr"))
    with pytest.raises(RuntimeError, match="does not contain"):
        DatasetMetaData.from_dataset_config(DatasetConfig(container_path=tmp_path / "c.zarr", dataset_name="missing"))
    with pytest.raises(RuntimeError, match="sample dimension"):
        DatasetMetaData((4, 4), ["y", "x"])


# ---------------------------------------------------------- entry points
def test_console_scripts_install_and_alias_package(tmp_path):
    """pyproject.toml declares the reference's console scripts (reference pyproject.toml:49-51:
    `train`, `infer`); an editable install into a scratch prefix produces them and they parse
    their argument like cellulus/cli.py.  `install_as_cellulus()` makes `import cellulus` resolve here."""
    import tomli

    doc = tomli.load(open(os.path.join(ROOT, "pyproject.toml"), "rb"))
    assert doc["project"]["scripts"] == {"train": "cellulus_amd.cli:train", "infer": "cellulus_amd.cli:infer"}
    prefix = tmp_path / "prefix"
    res = subprocess.run([sys.executable, "-m", "pip", "install", "--no-build-isolation", "--no-deps", "--no-index",
                          "--prefix", str(prefix), "-e", ROOT], capture_output=True, text=True)
    if res.returncode != 0:
        pytest.skip("pip cannot install here: " + res.stderr[-300:])
    env = dict(os.environ, PYTHONPATH=ROOT)
    for script in ("train", "infer"):
        exe = prefix / "bin" / script
        assert exe.exists()
        out = subprocess.run([str(exe), "--help"], capture_output=True, text=True, env=env, cwd=tmp_path)
        assert out.returncode == 0 and "CONFIG_FILE" in out.stdout, out.stderr
        out = subprocess.run([str(exe), "missing.toml"], capture_output=True, text=True, env=env, cwd=tmp_path)
        assert out.returncode == 2 and "does not exist" in out.stderr
    code = ("import cellulus_amd; cellulus_amd.install_as_cellulus(); import cellulus; "
            "from cellulus.configs import ExperimentConfig; import cellulus.models.unet as a; "
            "import cellulus_amd.models.unet as b; from cellulus.train import train; from cellulus.infer import infer; "
            "from cellulus.utils.mean_shift import mean_shift_segmentation; from cellulus.utils.misc import size_filter; "
            "assert a is b and cellulus is cellulus_amd; print('alias ok')")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=tmp_path)
    assert out.returncode == 0 and "alias ok" in out.stdout, out.stderr


# ---------------------------------------------------------------- topology
def test_topology_matches_oracle_shapes():
    from cellulus_amd.models.plan import build_topology
    from oracle.unet_oracle import OracleUNetModel

    cases = [
        (dict(in_channels=1, out_channels=2, num_fmaps=4, fmap_inc_factor=3, features_in_last_layer=8,
              downsampling_factors=[[2, 2]], num_spatial_dims=2), (60, 44)),
        (dict(in_channels=1, out_channels=2, num_fmaps=4, fmap_inc_factor=2, features_in_last_layer=8,
              downsampling_factors=[[2, 2], [3, 3]], num_spatial_dims=2), (108, 120)),
        # crop_to_factor is NOT a no-op here: 3x upsampling of 17 -> 51, (51-4) % 3 != 0
        (dict(in_channels=1, out_channels=2, num_fmaps=4, fmap_inc_factor=2, features_in_last_layer=8,
              downsampling_factors=[[3, 3]], num_spatial_dims=2), (67, 67)),
        (dict(in_channels=2, out_channels=3, num_fmaps=4, fmap_inc_factor=2, features_in_last_layer=8,
              downsampling_factors=[[1, 2, 2]], num_spatial_dims=3), (20, 28, 32)),
    ]
    for cfg, spatial in cases:
        oracle = OracleUNetModel(**cfg)
        with torch.no_grad():
            y = oracle(torch.zeros(1, cfg["in_channels"], *spatial))
        topo = build_topology(spatial=spatial, **cfg)
        assert tuple(topo.out_shape[3 - cfg["num_spatial_dims"]:]) == tuple(y.shape[2:]), (cfg, spatial)
    # output = crop - 16 for the single 2x level (zarr_dataset.py:94)
    assert build_topology(1, 2, 4, 3, 8, [[2, 2]], 2, (256, 256)).out_shape == (1, 240, 240)
    with pytest.raises(RuntimeError, match="downsample"):
        build_topology(1, 2, 4, 3, 8, [[2, 2]], 2, (45, 44))
    with pytest.raises(ValueError, match="too small"):
        build_topology(1, 2, 4, 3, 8, [[2, 2]], 2, (12, 12))


def test_state_dict_names_match_reference_layout():
    from cellulus_amd.models import get_model
    from oracle.unet_oracle import OracleUNetModel

    cfg = dict(in_channels=1, out_channels=2, num_fmaps=4, fmap_inc_factor=3, features_in_last_layer=8,
               downsampling_factors=[[2, 2], [2, 2]], num_spatial_dims=2)
    a, b = get_model(**cfg).state_dict(), OracleUNetModel(**cfg).state_dict()
    assert list(a) == list(b)
    assert all(a[k].shape == b[k].shape for k in a)
    assert "backbone.r_conv.0.1.conv_pass.6.bias" in a and "head.2.weight" in a


# ------------------------------------------------------------ pair sampler
def test_pair_sampler_counts_and_geometry(tmp_path):
    from cellulus_amd.configs import DatasetConfig
    from cellulus_amd.datasets import get_dataset
    from cellulus_amd.utils import zarr_io

    f = zarr_io.open(tmp_path / "d.zarr")
    raw = np.random.default_rng(0).random((4, 1, 300, 280)).astype(np.float32)
    raw[0] = 0.0                                            # an all-zero sample must be rejected
    f["train/raw"] = raw
    f["train/raw"].attrs["axis_names"] = ["s", "c", "y", "x"]
    ds = get_dataset(DatasetConfig(container_path=tmp_path / "d.zarr", dataset_name="train/raw"),
                     crop_size=(256, 256), elastic_deform=False, control_point_spacing=64,
                     control_point_jitter=2.0, density=0.1, kappa=10.0, normalization_factor=1.0)
    assert ds.get_num_anchors() == 4840 and ds.get_num_references() == 31
    assert ds.get_num_samples() == 150040
    np.random.seed(0)
    it = iter(ds)
    for _ in range(3):
        crop, anchor, ref = next(it)
        assert crop.shape == (1, 256, 256) and crop.dtype == np.float32 and crop.max() > 0
        assert anchor.shape == ref.shape == (150040, 2)
        assert anchor.min() >= 10 and anchor.max() <= 230
        d = ref - anchor
        assert ((d ** 2).sum(1) < 100).all() and (np.abs(d).sum(1) > 0).all()
        assert (anchor[:31] == anchor[0]).all()             # np.repeat layout: 31 refs per anchor
    # elastic path produces finite crops of the right shape
    ds2 = get_dataset(DatasetConfig(container_path=tmp_path / "d.zarr", dataset_name="train/raw"),
                      crop_size=(128, 128), elastic_deform=True, control_point_spacing=64,
                      control_point_jitter=2.0, density=0.1, kappa=10.0, normalization_factor=None)
    crop, _, _ = next(iter(ds2))
    assert crop.shape == (1, 128, 128) and np.isfinite(crop).all()


# ---------------------------------------------------------- centre de-dup
def test_dedup_centers_equals_oracle_dict_sort_dedup():
    from cellulus_amd.utils.mean_shift import dedup_centers
    from oracle import infer_oracle as IO

    rng = np.random.default_rng(0)
    truth = rng.random((12, 2)) * 200
    centers = truth[rng.integers(0, 12, size=400)] + rng.normal(0, 0.01, size=(400, 2))
    centers[50:60] = centers[50]                       # exact duplicates (dict keys collapse)
    counts = rng.integers(0, 50, size=400).astype(np.int32)
    got = dedup_centers(centers, counts, 15.0)
    # oracle's Python dict + sorted + greedy loop on the same converged centres
    d = {}
    for c, n in zip(centers, counts):
        if n:
            d[tuple(c)] = int(n)
    items = sorted(d.items(), key=lambda t: (t[1], t[0]), reverse=True)
    sc = np.array([t[0] for t in items])
    uniq = np.ones(len(sc), bool)
    for i, c in enumerate(sc):
        if uniq[i]:
            uniq[((sc - c) ** 2).sum(1) <= 225.0] = 0
            uniq[i] = 1
    np.testing.assert_array_equal(got, sc[uniq])
    with pytest.raises(ValueError, match="bandwidth"):
        dedup_centers(centers, np.zeros(400, dtype=np.int32), 15.0)
    assert IO is not None
    # libclx's host function (sorts + hash grid) against the numpy form, 2-D and 3-D, signed zeros, one-cell and wide grids
    from cellulus_amd.utils.mean_shift import _dedup_centers_numpy
    for trial in range(40):
        nd = 2 + trial % 2
        n, k = int(rng.integers(1, 2500)), int(rng.integers(1, 150))
        modes = rng.uniform(-100, 500, size=(k, nd))
        c = modes[rng.integers(0, k, size=n)] + rng.normal(0, rng.choice([1e-3, 0.5, 5.0]), size=(n, nd))
        if n > 20:
            c[5:15] = c[5]
            c[16] = -0.0
            c[17] = 0.0
        cnt = rng.integers(0, 5, size=n).astype(np.int32)
        cnt[0] = max(cnt[0], 1)
        bw = float(rng.choice([3.0, 15.0, 60.0, 1e4]))
        np.testing.assert_array_equal(dedup_centers(c, cnt, bw), _dedup_centers_numpy(c, cnt, bw))


# -------------------------------------------------------------- evaluation
def test_pairwise_iou_equals_bruteforce_definition():
    """Host half of evaluate (joint histogram -> IoU / SEG / F1); the histogram is a HIP kernel
    (tests/test_gpu_infer.py) and comes from the oracle here."""
    from cellulus_amd.evaluate import compute_F1, iou_from_joint
    from oracle import infer_oracle as IO

    def compute_pairwise_IoU(prediction, groundtruth):
        return iou_from_joint(*IO.joint_histogram(prediction, groundtruth))

    rng = np.random.default_rng(0)
    pred = np.kron(rng.integers(0, 5, size=(6, 6)), np.ones((4, 4), dtype=np.int64)).astype(np.uint16)
    gt = np.roll(pred, 2, axis=1)
    iou, seg, n = compute_pairwise_IoU(pred, gt)
    pids = [i for i in np.unique(pred) if i != 0]
    gids = [i for i in np.unique(gt) if i != 0]
    ref = np.zeros((len(pids), len(gids)))
    iog = np.zeros_like(ref)
    for j, p in enumerate(pids):       # cellulus/evaluate.py:72-97
        for k, g in enumerate(gids):
            inter = ((pred == p) & (gt == g)).sum()
            ref[j, k] = inter / ((pred == p) | (gt == g)).sum()
            iog[j, k] = inter / (gt == g).sum()
    np.testing.assert_allclose(iou, ref)
    assert n == len(gids) and abs(seg - ref[iog > 0.5].sum()) < 1e-12
    f1, tp, fp, fn = compute_F1(iou)
    assert 0 <= f1 <= 1 and tp + fn == len(gids)
    assert compute_pairwise_IoU(pred, np.zeros_like(gt)) is None


def test_tile_offsets():
    from cellulus_amd.predict import tile_offsets

    assert tile_offsets(512, 240) == [0, 240, 272]
    assert tile_offsets(240, 240) == [0]
    assert tile_offsets(480, 240) == [0, 240]
    with pytest.raises(RuntimeError):
        tile_offsets(100, 240)


def test_logger_csv_layout(tmp_path, monkeypatch):
    from cellulus_amd.utils import get_logger

    monkeypatch.chdir(tmp_path)
    lg = get_logger(keys=["loss", "oce_loss"], title="loss")
    for i in range(3):
        lg.add("loss", 1.5 * i)
        lg.add("oce_loss", 0.5 * i)
        lg.write()
    lines = open("loss.csv").read().strip().split("\n")
    assert lines[0] == ",loss,oce_loss" and lines[3].split(",")[0] == "2" and len(lines) == 4
    assert float(lines[2].split(",")[1]) == 1.5


# ------------------------------------------------------------- DDP (gloo)

def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
_DDP_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from cellulus_amd import parallel
rank, world, _ = parallel.init_from_env(backend="gloo")
assert world == 2
# loss is a SUM over pairs => global gradient = SUM of per-rank gradients (no averaging)
w = torch.tensor([1.0, -2.0, 0.5])
x = torch.arange(12.0).view(4, 3) + 1
shard = x[rank * 2:(rank + 1) * 2]
g = (2 * (shard @ w)[:, None] * shard).sum(0)
flat = g.clone()
parallel.all_reduce_sum_(flat)
full = (2 * (x @ w)[:, None] * x).sum(0)
assert torch.allclose(flat, full), (flat, full)
p = torch.full((5,), float(rank))
parallel.broadcast_(p, src=0)
assert (p == 0).all()
lo, hi = parallel.shard_range(7)
assert (lo, hi) == ((0, 4) if rank == 0 else (4, 7))
sums = torch.tensor([float(rank + 1)], dtype=torch.float64)
parallel.all_reduce_sum_(sums)
assert sums.item() == 3.0
# gradient buckets: parameters complete from the tail of the flat buffer, out of order in places;
# only the completed SUFFIX is reduced, every element exactly once, result = SUM over ranks
sizes = [5, 1, 40, 3, 64, 7, 2]
flat = torch.arange(float(sum(sizes))) * (rank + 1)
views, off = [], 0
for n in sizes:
    views.append(flat[off:off + n]); off += n
extra = torch.tensor([10.0 * (rank + 1)], dtype=torch.float64)
gb = parallel.GradientBuckets(flat, views, min_bytes=40 * 4)
gb.add(extra)
gb.params_done(6); assert gb.issued == []                       # 2 elements < 40
gb.params_done(4); assert gb.issued == []                       # 5 is missing: suffix still [6:]
gb.params_done(5); assert gb.issued == [(49, 122)]              # 64 + 7 + 2 elements
gb.params_done(3, 2); assert gb.issued[-1] == (6, 49)
gb.params_done(1, 0); assert len(gb.issued) == 2                # 6 elements wait for finish()
gb.finish()
assert gb.issued == [(49, 122), (6, 49), (0, 6)] and gb.works == []
assert torch.equal(flat, torch.arange(float(sum(sizes))) * 3) and extra.item() == 30.0
gb0 = parallel.GradientBuckets(flat, views, min_bytes=1 << 30)   # nothing before finish
for i in reversed(range(len(sizes))):
    gb0.params_done(i)
assert gb0.issued == []
gb0.finish()
assert gb0.issued == [(0, 122)] and torch.equal(flat, torch.arange(float(sum(sizes))) * 6)
# a rank-0-only step that fails makes EVERY rank raise (nobody is left waiting at a barrier)
def boom():
    raise FileExistsError("dataset exists")
try:
    parallel.rank0_first(boom)
    raise SystemExit("rank0_first swallowed the error")
except FileExistsError:
    assert rank == 0
except RuntimeError as e:
    assert rank != 0 and "FileExistsError: dataset exists" in str(e)
ran = []
parallel.rank0_first(lambda: ran.append(rank))
assert ran == ([0] if rank == 0 else [])
dist.barrier()
print("rank", rank, "ok")
"""

_DDP8_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from cellulus_amd import parallel
rank, world, _ = parallel.init_from_env(backend="gloo")
assert world == 8 and parallel.world_size() == 8 and parallel.rank() == rank
# 5 independent samples on 8 ranks: three ranks get nothing, the others one each, no overlap
lo, hi = parallel.shard_range(5)
assert (lo, hi) == ((rank, rank + 1) if rank < 5 else (5, 5))
mine = torch.zeros(5); mine[lo:hi] = 1
parallel.all_reduce_sum_(mine)
assert torch.equal(mine, torch.ones(5))
# SUM, never a mean: eight ranks' gradients of a summed loss
g = torch.full((1000,), float(rank + 1))
parallel.all_reduce_sum_(g)
assert torch.equal(g, torch.full((1000,), 36.0))
# the bucket sequence depends on the plan only: every rank issues the same ranges in the same order
sizes = [3, 300, 17, 1024, 64, 5, 2048, 9]
flat = torch.ones(sum(sizes)) * (rank + 1)
views, off = [], 0
for n in sizes:
    views.append(flat[off:off + n]); off += n
gb = parallel.GradientBuckets(flat, views, min_bytes=1024 * 4)
for i in (7, 6, 4, 5, 3, 1, 2, 0):
    gb.params_done(i)
gb.finish()
assert torch.equal(flat, torch.full((sum(sizes),), 36.0))
issued = [None] * world
dist.all_gather_object(issued, gb.issued)
assert all(i == issued[0] for i in issued) and issued[0][0] == (1413, 3470), issued[0]
def boom():
    raise FileExistsError("dataset exists")
try:
    parallel.rank0_first(boom)
    raise SystemExit("rank0_first swallowed the error")
except FileExistsError:
    assert rank == 0
except RuntimeError as e:
    assert rank != 0 and "FileExistsError" in str(e)
dist.barrier()
print("rank", rank, "ok")
"""


def _run_ranks(script_text, world, tmp_path):
    script = tmp_path / f"ddp{world}.py"
    script.write_text(script_text)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
               OMP_NUM_THREADS="1")
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


def test_data_parallel_wiring_gloo_world8(tmp_path):
    """Eight ranks (the node size of BASELINE configs[2]): ranks without samples, SUM semantics, one
    bucket sequence on every rank, a failing rank-0-only step raising everywhere."""
    _run_ranks(_DDP8_SCRIPT, 8, tmp_path)


def test_multi_rank_train_defaults_to_device_pairs_and_caps_the_loaders(monkeypatch):
    """train()'s input-pipeline policy (host side only): one rank = the reference's loader; several ranks
    share the host — loader processes capped to the rank's share of the cores, pairs drawn on the device
    unless CLX_DEVICE_PAIRS=0."""
    from cellulus_amd.train import loader_policy

    monkeypatch.delenv("CLX_DEVICE_PAIRS", raising=False)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.setattr(os, "sched_getaffinity", lambda _pid: set(range(64)))
    one = loader_policy(1, 8)
    assert one["loader_procs"] == 8 and not one["device_pairs"] and one["host_cores_per_rank"] == 64
    eight = loader_policy(8, 8)
    assert eight["loader_procs"] == 7 and eight["device_pairs"] and eight["host_cores_per_rank"] == 8
    assert loader_policy(8, 4)["loader_procs"] == 4
    # two nodes of eight ranks: a rank's share is the host's cores over the ranks ON THAT HOST
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    sixteen = loader_policy(16, 8)
    assert sixteen["host_cores_per_rank"] == 8 and sixteen["loader_procs"] == 7 and sixteen["device_pairs"]
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    monkeypatch.setenv("CLX_DEVICE_PAIRS", "0")
    assert not loader_policy(8, 8)["device_pairs"]
    monkeypatch.setenv("CLX_DEVICE_PAIRS", "1")
    assert loader_policy(1, 8)["device_pairs"]


def test_data_parallel_wiring_gloo_world2(tmp_path):
    script = tmp_path / "ddp.py"
    script.write_text(_DDP_SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o
    assert parallel_shard_cover()


def parallel_shard_cover():
    from cellulus_amd import parallel

    for n in (0, 1, 7, 8, 64):
        for w in (1, 2, 3, 8):
            got = []
            for r in range(w):
                lo, hi = parallel.shard_range(n, r, w)
                got += list(range(lo, hi))
            if got != list(range(n)):
                return False
    return True


def test_peak_local_max_matches_skimage_golden():
    """detect.py:128-132 seeds: the restated peak_local_max equals scikit-image 0.18.3."""
    from cellulus_amd.detect import peak_local_max

    g = np.load(os.path.join(G, "g5_skimage.npz"))
    for k in ("peaks", "peaks3"):
        np.testing.assert_array_equal(peak_local_max(g[f"{k}/image"]), g[f"{k}/coords"])


@pytest.mark.parametrize("nd", [2, 3])
def test_subpixel_phase_weights_equal_upsample_then_conv(nd):
    """Algebra behind the sub-pixel rewrite (plan._phase_weights / _fold_phase_grads), on the CPU:
    conv3(nearest_upsample2(x), w) == depth_to_space(conv2(x, phase_weights(w))) and the fold is
    the adjoint of the phase summation."""
    from cellulus_amd.models.plan import UNetPlan

    class L:
        pass

    torch.manual_seed(0)
    f = (1, 2, 2) if nd == 2 else (2, 2, 2)
    layer = L()
    layer.cout = 5
    layer.kernel = (1, 3, 3) if nd == 2 else (3, 3, 3)
    C1, N = 3, 8
    sp = dict(fac=f, P=f[0] * f[1] * f[2], N=N, C1=C1, zk=tuple(2 if ff == 2 else k for ff, k in zip(f, layer.kernel)))
    plan = UNetPlan.__new__(UNetPlan)
    w = torch.randn((5, C1) + layer.kernel)
    weff = plan._phase_weights(layer, sp, w)
    low = torch.randn(2, C1, *((1, 7, 8) if nd == 2 else (5, 6, 7)))
    ref = torch.nn.functional.conv3d(torch.nn.functional.interpolate(low, scale_factor=f, mode="nearest"), w)
    z = torch.nn.functional.conv3d(low, weff)
    zs = z.shape[2:]
    z = z.reshape(2, f[0], f[1], f[2], N, *zs)[:, :, :, :, :5]
    out = z.permute(0, 4, 5, 1, 6, 2, 7, 3).reshape(2, 5, zs[0] * f[0], zs[1] * f[1], zs[2] * f[2])
    assert (out - ref).abs().max().item() < 1e-4
    g = torch.randn_like(weff)
    assert abs((weff * g).sum().item() - (w * plan._fold_phase_grads(layer, sp, g)).sum().item()) < 1e-3


def test_evaluate_matches_real_reference_golden():
    """g10: the REAL cellulus.evaluate.compute_pairwise_IoU / compute_F1 (O(#pred x #gt) mask loops)
    vs (i) the oracle's restatement of those loops and (ii) the product's host step that turns a
    joint histogram of id pairs into the same tables (the histogram itself is a HIP kernel —
    tests/test_gpu_infer.py — here it comes from the oracle): identical IoU, SEG, F1, TP, FP, FN."""
    from cellulus_amd.evaluate import compute_F1, iou_from_joint
    from oracle import infer_oracle as IO

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_evaluate.npz"))
    for i in range(3):
        pred, gt = g[f"{i}/pred"], g[f"{i}/gt"]
        iou_o, seg_o, n_o = IO.compute_pairwise_IoU(pred, gt)
        np.testing.assert_array_equal(iou_o, g[f"{i}/iou"])
        np.testing.assert_allclose([seg_o, n_o, *IO.compute_F1(iou_o)], g[f"{i}/scalars"], rtol=1e-15)
        iou, seg, n = iou_from_joint(*IO.joint_histogram(pred, gt))
        np.testing.assert_array_equal(iou, g[f"{i}/iou"])
        f1, tp, fp, fn = compute_F1(iou)
        np.testing.assert_allclose([seg, n, f1, tp, fp, fn], g[f"{i}/scalars"], rtol=1e-15)
    assert bool(g["none_for_empty_gt"])
    assert IO.compute_pairwise_IoU(g["0/pred"], np.zeros_like(g["0/gt"])) is None
    assert iou_from_joint(*IO.joint_histogram(g["0/pred"], np.zeros_like(g["0/gt"]))) is None


def test_evaluate_has_no_cpu_path():
    from cellulus_amd import _clx
    from cellulus_amd.evaluate import compute_pairwise_IoU

    if torch.cuda.is_available():
        pytest.skip("HIP device present")
    with pytest.raises(_clx.ClxError, match="no CPU path"):
        compute_pairwise_IoU(np.ones((4, 4), np.uint16), np.ones((4, 4), np.uint16))


@pytest.mark.parametrize("tag", ["2d", "3d", "2d_small"])
def test_pair_sampler_matches_real_reference_golden(tag):
    """g11: the REAL ZarrDataset.sample_coordinates (zarr_dataset.py:163-248) under np.random.seed
    vs this package's sampler: identical anchor / reference coordinates, identical counts."""
    from cellulus_amd.datasets.zarr_dataset import ZarrDataset

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_pair_sampler.npz"))
    nd, kappa, density, seed = g[f"{tag}/params"]
    ds = object.__new__(ZarrDataset)
    ds.num_spatial_dims, ds.kappa, ds.density = int(nd), float(kappa), float(density)
    ds.output_shape = tuple(int(v) for v in g[f"{tag}/output_shape"])
    ds.unbiased_shape = tuple(int(s - 2 * ds.kappa) for s in ds.output_shape)
    np.random.seed(int(seed))
    anchors, references = ds.sample_coordinates()
    np.testing.assert_array_equal(anchors, g[f"{tag}/anchors"])
    np.testing.assert_array_equal(references, g[f"{tag}/references"])
    np.testing.assert_array_equal([ds.get_num_anchors(), ds.get_num_references(), ds.get_num_samples()],
                                  g[f"{tag}/counts"])


# ------------------------------------------------------------------ Blosc input
def test_blosc_chunks_written_by_the_real_c_blosc_decode(tmp_path):
    """g12: chunks compressed by the real c-blosc (tests/golden/make_golden_blosc.py, conda's
    imagecodecs) — LZ4 / LZ4HC / zlib / zstd / BloscLZ, byte shuffle, bit shuffle (whole blocks and the
    unshuffled odd ones) and none, split and unsplit blocks, several blocks with a shorter last one, an
    incompressible (stored) chunk, a tiny one — decode to the original arrays, directly and through a zarr
    array whose compressor is zarr's default Blosc; numcodecs' plain Zstd / LZ4 / BZ2 compressors too."""
    import json

    from cellulus_amd.utils import zarr_io

    g = np.load(os.path.join(G, "g12_blosc.npz"))
    names = sorted({k.split("/")[0] for k in g.files})
    assert len(names) >= 17
    for name in names:
        arr, chunk = g[f"{name}/array"], g[f"{name}/chunk"].tobytes()
        assert zarr_io.blosc_decode(chunk) == arr.tobytes(), name
    # a zarr array as zarr-python lays it out: one chunk per sample, compressor = Blosc(lz4, 5, SHUFFLE)
    arr = g["u8_lz4_shuffle/array"]                       # (2, 1, 64, 80): the whole array is one chunk here
    path = tmp_path / "c.zarr" / "train" / "raw"
    os.makedirs(path)
    (tmp_path / "c.zarr" / ".zgroup").write_text('{"zarr_format": 2}')
    (tmp_path / "c.zarr" / "train" / ".zgroup").write_text('{"zarr_format": 2}')
    (path / ".zarray").write_text(json.dumps({
        "zarr_format": 2, "shape": list(arr.shape), "chunks": list(arr.shape), "dtype": arr.dtype.str,
        "compressor": {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0},
        "fill_value": 0, "order": "C", "filters": None}))
    (path / ".zattrs").write_text(json.dumps({"axis_names": ["s", "c", "y", "x"]}))
    (path / "0.0.0.0").write_bytes(g["u8_lz4_shuffle/chunk"].tobytes())
    ds = zarr_io.open(tmp_path / "c.zarr", "r")["train/raw"]
    np.testing.assert_array_equal(ds[...], arr)
    np.testing.assert_array_equal(ds[1, 0, 10:20, 5:9], arr[1, 0, 10:20, 5:9])
    # corrupt / unsupported input fails loudly
    bad = bytearray(g["f32_lz4_shuffle/chunk"].tobytes())
    bad[2] = (bad[2] & 0x1f) | (6 << 5)                   # a codec id c-blosc does not define
    with pytest.raises(zarr_io.ZarrError, match="codec id 6"):
        zarr_io.blosc_decode(bytes(bad))
    bad[2] = (bad[2] & 0x1f) | (4 << 5)                   # claims zstd: the LZ4 bytes are not a zstd frame
    with pytest.raises(Exception):
        zarr_io.blosc_decode(bytes(bad))
    # numcodecs' stand-alone compressors: Zstd (a frame carrying its content size), LZ4 (int32 size + block), BZ2
    import bz2

    import pyarrow

    payload = g["i64_labels_lz4/array"].tobytes()
    frame = pyarrow.Codec("zstd").compress(payload, asbytes=True)
    assert zarr_io._decode(frame, {"id": "zstd", "level": 1}) == payload
    block = pyarrow.Codec("lz4_raw").compress(payload, asbytes=True)
    assert zarr_io._decode(len(payload).to_bytes(4, "little") + block, {"id": "lz4", "acceleration": 1}) == payload
    assert zarr_io._decode(bz2.compress(payload), {"id": "bz2", "level": 1}) == payload
    with pytest.raises(zarr_io.ZarrError, match="lzma"):
        zarr_io._decode(b"", {"id": "lzma"})
    with pytest.raises(zarr_io.ZarrError):
        zarr_io.blosc_decode(g["f32_lz4_shuffle/chunk"].tobytes()[:40] + b"\x00" * 100)


def test_blosc_writer_round_trips_and_is_read_by_the_real_c_blosc(tmp_path):
    """clx_blosc_compress_lz4 (the writer behind zarr_io's default compressor — zarr-python's default, which the
    reference's stages use for their outputs): smooth / label / random / tiny / empty arrays of several item sizes,
    with and without byte shuffle, decode to the same bytes through zarr_io's reader (itself pinned by chunks from
    the real c-blosc) and — where conda's imagecodecs is present — through the real c-blosc; create_dataset writes
    zarr's default compressor into .zarray, compressor=None plain chunks."""
    import json
    import subprocess

    from cellulus_amd.utils import zarr_io

    rng = np.random.default_rng(0)
    yy, xx = np.meshgrid(np.arange(300, dtype=np.float32), np.arange(400, dtype=np.float32), indexing="ij")
    smooth = np.exp(-((yy - 150) ** 2 + (xx - 200) ** 2) / 5000).astype(np.float32)
    cases = {
        "f32": smooth, "f64": smooth.astype(np.float64), "u8": (smooth * 255).astype(np.uint8),
        "u16": (smooth * 60000).astype(np.uint16), "i64_labels": np.repeat(rng.integers(0, 9, size=4000), 173).astype(np.int64),
        "random_u8": rng.integers(0, 256, size=70001, dtype=np.uint8), "tiny": np.arange(5, dtype=np.int32),
        "empty": np.zeros(0, np.float32), "odd_f32": smooth.reshape(-1)[:99991].copy(),
        "big_f64": np.tile(smooth.astype(np.float64), (3, 1)),            # several 256-KB blocks + a shorter last one
    }
    comp = dict(zarr_io.DEFAULT_COMPRESSOR)
    for name, arr in cases.items():
        for shuffle in (1, 0):
            chunk = zarr_io._encode(arr.tobytes(), dict(comp, shuffle=shuffle), arr.dtype.itemsize)
            assert zarr_io._decode(chunk, comp) == arr.tobytes(), (name, shuffle)
            if name in ("f32", "f64", "u16", "i64_labels", "u8", "big_f64"):
                assert len(chunk) < (0.8 if shuffle else 0.95) * arr.nbytes, (name, shuffle, len(chunk))
            if name == "random_u8":
                assert chunk[2] & 0x02 and len(chunk) == arr.nbytes + 16            # stored verbatim
            (tmp_path / f"{name}_{shuffle}.blosc").write_bytes(chunk)
            np.save(tmp_path / f"{name}_{shuffle}.npy", arr)
    with pytest.raises(zarr_io.ZarrError, match="LZ4"):
        zarr_io._encode(b"abcd", dict(comp, cname="zstd"), 1)
    conda = "/opt/conda/bin/python3.9"
    if os.path.exists(conda):
        script = ("import imagecodecs, numpy as np, glob\n"
                  f"fs = sorted(glob.glob('{tmp_path}/*.blosc'))\n"
                  "n = 0\n"
                  "for f in fs:\n"
                  "    a = np.load(f[:-6] + '.npy')\n"
                  "    if a.size:\n"
                  "        assert imagecodecs.blosc_decode(open(f, 'rb').read()) == a.tobytes(), f\n"
                  "        n += 1\n"
                  "print('decoded', n)\n")
        res = subprocess.run([conda, "-c", script], capture_output=True, text=True)
        if "No module named" not in res.stderr:
            assert res.returncode == 0 and "decoded 18" in res.stdout, res.stderr[-2000:]
    f = zarr_io.open(tmp_path / "w.zarr")
    ds = f.create_dataset("a/b", shape=(2, 300, 400), dtype=np.float32)
    ds[0], ds[1] = smooth, 2 * smooth
    meta = json.loads((tmp_path / "w.zarr" / "a" / "b" / ".zarray").read_text())
    assert meta["compressor"] == zarr_io.DEFAULT_COMPRESSOR and meta["chunks"] == [1, 300, 400]
    assert os.path.getsize(tmp_path / "w.zarr" / "a" / "b" / "1.0.0") < 0.8 * smooth.nbytes
    np.testing.assert_array_equal(zarr_io.open(tmp_path / "w.zarr", "r")["a/b"][1], 2 * smooth)
    raw = f.create_dataset("plain", shape=(300, 400), dtype=np.float32, chunks=(300, 400), compressor=None)
    raw[...] = smooth
    assert (tmp_path / "w.zarr" / "plain" / "0.0").read_bytes() == smooth.tobytes()


def test_decoded_chunk_cache_serves_repeated_crops_and_sees_rewrites(tmp_path, monkeypatch):
    """A compressed array decodes a chunk once for repeated crops (least recently used chunks leave when
    CLX_ZARR_CACHE_MB is exceeded), its own writes and another handle's rewrite of the chunk file are seen."""
    from cellulus_amd.utils import zarr_io

    rng = np.random.default_rng(0)
    data = rng.random((4, 1, 64, 64), dtype=np.float32)
    f = zarr_io.open(tmp_path / "c.zarr")
    f["raw"] = data
    calls = []
    real = zarr_io._decode
    monkeypatch.setattr(zarr_io, "_decode", lambda raw, comp, n=None: (calls.append(1), real(raw, comp, n))[1])
    monkeypatch.setenv("CLX_ZARR_CACHE_MB", str(2.5 * 64 * 64 * 4 / (1 << 20)))       # room for two chunks
    a = zarr_io.open(tmp_path / "c.zarr", "r")["raw"]
    for _ in range(5):
        np.testing.assert_array_equal(a[1, 0, 10:30, 5:25], data[1, 0, 10:30, 5:25])
    assert len(calls) == 1
    a[2], a[1], a[3]
    assert len(calls) == 3                      # 1 was still cached; 2 and 3 were decoded
    a[2]
    assert len(calls) == 4                      # ... and 2, the least recently used, had to leave for 3
    b = zarr_io.open(tmp_path / "c.zarr")["raw"]
    b[1] = data[1] + 1.0                        # another handle (another rank) rewrites the chunk file
    np.testing.assert_array_equal(a[1], data[1] + 1.0)
    a2 = zarr_io.open(tmp_path / "c.zarr")["raw"]
    a2[0, 0, :8, :8] = 7.0                      # partial write through the handle that cached
    assert a2[0, 0, 3, 3] == 7.0 and a2[0, 0, 20, 20] == data[0, 0, 20, 20]


def test_chunk_headers_are_checked_against_the_array_before_allocating():
    """A corrupt or foreign chunk whose header announces another size than prod(chunks) * itemsize
    is rejected up front (no multi-GiB allocation from an untrusted field)."""
    from cellulus_amd.utils import zarr_io

    g = np.load(os.path.join(G, "g12_blosc.npz"))
    chunk = g["f32_lz4_shuffle/chunk"].tobytes()
    nbytes = int.from_bytes(chunk[4:8], "little")
    assert len(zarr_io._decode(chunk, {"id": "blosc"}, nbytes)) == nbytes
    with pytest.raises(zarr_io.ZarrError, match="announces"):
        zarr_io._decode(chunk, {"id": "blosc"}, nbytes + 4)
    huge = bytearray(chunk)
    huge[4:8] = (0xFFFFFFF0).to_bytes(4, "little")
    with pytest.raises(zarr_io.ZarrError, match="announces"):
        zarr_io._decode(bytes(huge), {"id": "blosc"}, nbytes)
    with pytest.raises(zarr_io.ZarrError):                       # block size larger than the chunk
        bad = bytearray(chunk)
        bad[8:12] = (nbytes * 2).to_bytes(4, "little")
        zarr_io.blosc_decode(bytes(bad))
    with pytest.raises(zarr_io.ZarrError, match="announces"):
        zarr_io._decode((1 << 31).to_bytes(4, "little") + b"\x00" * 8, {"id": "lz4"}, 64)
    with pytest.raises(zarr_io.ZarrError):
        zarr_io._decode(b"\x01", {"id": "lz4"}, 64)


def test_corrupted_blosc_chunks_raise_or_decode_but_never_crash():
    """The chunk decoders read files: 1500 random corruptions (byte flips anywhere, truncations) of the golden
    chunks — every codec and shuffle mode — must end in an exception or in some bytes, never in a crash of the
    host functions (clx_lz4_decompress / clx_blosclz_decompress / clx_unshuffle_bytes bound every copy)."""
    import random

    from cellulus_amd.utils import zarr_io

    g = np.load(os.path.join(G, "g12_blosc.npz"))
    names = sorted({k.split("/")[0] for k in g.files})
    rnd = random.Random(0)
    raised = 0
    for _ in range(1500):
        b = bytearray(g[rnd.choice(names) + "/chunk"].tobytes())
        for _k in range(rnd.randint(1, 4)):
            b[rnd.randrange(4, len(b))] = rnd.randrange(256)      # (bytes 0-3: versions / flags / typesize also hit)
        if rnd.random() < 0.2:
            b = b[:rnd.randrange(1, len(b))]
        try:
            zarr_io.blosc_decode(bytes(b))
        except Exception:
            raised += 1
    assert raised > 100


# ------------------------------------------------------- round-2 host pieces
def test_noise_prefetcher_draws_the_reference_sequence():
    """predict.NoisePrefetcher: the background thread makes exactly the torch.rand calls of the
    reference's predict(): FIRST the 2 * num_infer_iterations draws of the dry-run forward on a zero
    tile (predict.py:32-39 calls the model after set_infer, so unet.py:75-88 runs once before the
    scan), then one per noisy copy, tile after tile (unet.py:81) — same numbers, same final
    generator state — only earlier."""
    from cellulus_amd.predict import NoisePrefetcher

    tile = (1, 1, 20, 24)
    torch.manual_seed(7)
    for _ in range(6):                                # the dry run: drawn, never used
        torch.rand(*tile)
    ref = [torch.stack([torch.rand(*tile) for _ in range(6)]) for _tile in range(3)]
    after = torch.rand(4)
    torch.manual_seed(7)
    pre = NoisePrefetcher(num_tiles=3, copies=6, tile_shape=tile, depth=2, dry_run=6)
    got = [pre.next().clone() for _ in range(3)]
    pre.finish()
    for a, b in zip(got, ref):
        assert a.shape == (6,) + tile and torch.equal(a, b)
    assert torch.equal(torch.rand(4), after)          # the generator is where the reference leaves it
    with pytest.raises(AssertionError):
        pre.next()
    # a rank without samples (more ranks than samples) still makes the dry-run draws and stops
    torch.manual_seed(7)
    pre = NoisePrefetcher(num_tiles=0, copies=6, tile_shape=tile, dry_run=6)
    pre.finish()
    torch.manual_seed(7)
    for _ in range(6):
        torch.rand(*tile)
    state = torch.get_rng_state()
    torch.manual_seed(7)
    pre = NoisePrefetcher(num_tiles=0, copies=6, tile_shape=tile, dry_run=6)
    pre.finish()
    assert torch.equal(torch.get_rng_state(), state)


def test_oracle_scan_replays_the_reference_predict_call_sequence():
    """oracle.unet_oracle.predict_scan against a LITERAL transcription of what the reference's
    predict() makes the model do (cellulus/predict.py:21-39 + gp.torch.Predict per tile, the model
    being cellulus/models/unet.py:73-100): set_infer, model(zeros) — a full infer-mode forward —,
    then one model(tile) per scanned tile.  Both forms of the oracle's dry run (the forward itself /
    its draws alone) give the same embeddings and leave the generator in the same state."""
    import itertools

    from oracle.unet_oracle import OracleUNetModel, predict_scan

    torch.manual_seed(3)
    model = OracleUNetModel(in_channels=1, out_channels=2, num_fmaps=4, fmap_inc_factor=2,
                            features_in_last_layer=8, downsampling_factors=[(2, 2)], num_spatial_dims=2)
    raw = np.random.RandomState(0).rand(2, 1, 50, 64).astype(np.float32)
    crop, n_it, p = (40, 40), 3, 0.1

    torch.manual_seed(11)
    model.set_infer(p, n_it)
    with torch.no_grad():
        out_shape = model(torch.zeros(1, 1, *crop)).shape          # predict.py:32-39
    assert tuple(out_shape) == (1, 3, 24, 24)
    expect = np.zeros((2, 3, 50, 64))
    for s in range(2):
        padded = np.pad(raw[s], [(0, 0), (8, 8), (8, 8)], mode="reflect")
        for oy, ox in itertools.product([0, 24, 26], [0, 24, 40]):
            with torch.no_grad():
                e = model(torch.from_numpy(padded[:, oy:oy + 40, ox:ox + 40].copy())[None])[0].numpy()
            expect[s, :, oy:oy + 24, ox:ox + 24] = e
    state = torch.get_rng_state()

    for literal in (True, False):
        torch.manual_seed(11)
        got = predict_scan(model, raw, crop, p, n_it, 1.0, literal_dry_run=literal)
        np.testing.assert_array_equal(got, expect)
        assert torch.equal(torch.get_rng_state(), state)
    # without the dry run the first tile would have seen other numbers
    torch.manual_seed(11)
    with torch.no_grad():
        first = model(torch.from_numpy(np.pad(raw[0], [(0, 0), (8, 8), (8, 8)], mode="reflect")[:, :40, :40].copy())[None])
    assert np.abs(first[0].numpy() - expect[0, :, :24, :24]).max() > 0


def test_gaussian_weights_are_scipys():
    from scipy.ndimage import _filters

    from cellulus_amd.detect import gaussian_weights

    for sigma in (2.0, 0.7, 3.5):
        w, r = gaussian_weights(sigma)
        ref = _filters._gaussian_kernel1d(sigma, 0, int(4.0 * sigma + 0.5))
        assert r == (len(ref) - 1) // 2
        np.testing.assert_array_equal(w, ref[r:])
        np.testing.assert_array_equal(ref[:r + 1][::-1], w)      # symmetric: scipy takes the symmetric branch


def test_centre_grid_and_device_pair_table():
    from cellulus_amd.datasets.zarr_dataset import DevicePairSampler, ZarrDataset
    from cellulus_amd.utils.mean_shift import _center_grid

    rng = np.random.default_rng(0)
    for nd in (2, 3):
        centers = rng.uniform(-20, 300, size=(200, nd))
        order, cstart, origin, dims = _center_grid(centers, 15.0)
        cells = np.floor((centers - origin) / 15.0).astype(int)
        cid = cells[:, 0] + dims[0] * cells[:, 1] + (dims[0] * dims[1] * cells[:, 2] if nd == 3 else 0)
        assert cstart[-1] == 200 and sorted(order.tolist()) == list(range(200))
        for c in range(dims[0] * dims[1] * dims[2]):
            assert set(order[cstart[c]:cstart[c + 1]].tolist()) == set(np.flatnonzero(cid == c).tolist())
        ds = ZarrDataset.__new__(ZarrDataset)
        ds.num_spatial_dims, ds.kappa, ds.density = nd, 5.0, 0.1
        ds.output_shape = (40,) * nd
        ds.unbiased_shape = (30,) * nd
        s = DevicePairSampler(ds, torch.device("cpu"), seed=1)       # the table is host work; sample() needs HIP
        table = {tuple(o) for o in s.offsets.numpy().tolist()}
        np.random.seed(0)
        ref = {tuple(o) for o in ds.sample_offsets_within_radius(5.0, 20000).tolist()}
        assert ref <= table and len(table - ref) <= 2              # the reference's rejection loop fills the same set
        assert (0,) * nd not in table and all(sum(v * v for v in o) < 25 for o in table)
        assert s.lo == 5 and s.hi == [35] * nd


def test_ctypes_descriptor_layout_equals_the_c_struct(tmp_path):
    """cellulus_amd/_clx.py::ClxConvDesc against include/clx.h::clx_conv_desc as gcc lays it out: size and
    the offset of every field (a field inserted on one side only shifts pointers silently)."""
    from cellulus_amd import _clx

    fields = [name for name, _t in _clx.ClxConvDesc._fields_]
    prog = ['#include <stddef.h>', '#include <stdio.h>', f'#include "{os.path.join(ROOT, "include", "clx.h")}"',
            'int main(void) {', '  printf("%zu\\n", sizeof(clx_conv_desc));']
    prog += [f'  printf("%zu\\n", offsetof(clx_conv_desc, {name}));' for name in fields]
    prog += ['  printf("%zu\\n", sizeof(clx_src));', '  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    vals = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert vals[0] == ctypes.sizeof(_clx.ClxConvDesc)
    for name, off in zip(fields, vals[1:1 + len(fields)]):
        assert getattr(_clx.ClxConvDesc, name).offset == off, name
    assert vals[-1] == ctypes.sizeof(_clx.ClxSrc)


def test_fused_1x1_pairs_need_exclusive_tensors():
    """the fused backward of a 64-channel 1x1 pair overwrites the gradient of the pair's input and never writes
    the middle tensor's, so a pair is formed only if the middle tensor is read by the second layer alone and the input by
    the first layer alone.  The reference topology qualifies (two pairs: the top level's 1x1 layers and the head); a
    tensor that is also pooled, a skip connection or read by a second convolution keeps the layer-by-layer path."""
    import copy

    from cellulus_amd.models import plan as P

    topo = P.build_topology(1, 2, 64, 3, 64, [(2, 2)], 2, (64, 64))
    direct = {layer.name: 0 for layer in topo.convs}
    pairs = P.find_chain_pairs(topo, direct, 2)
    assert [(a.name, b.name) for a, b in pairs] == [
        ("backbone.l_conv.0.conv_pass.2", "backbone.l_conv.0.conv_pass.4"),
        ("backbone.r_conv.0.0.conv_pass.2", "backbone.r_conv.0.0.conv_pass.4"), ("head.0", "head.2")]
    readers = P.tensor_consumers(topo)
    assert readers["l0.3"] == 2 and readers["l0.0"] == readers["l0.1"] == 1      # l0.3: pooled AND the skip connection
    by_name = {layer.name: layer for layer in topo.convs}
    # (1) the MIDDLE tensor of the first pair gets a second reader (a pooling): that pair is dropped, the others stay
    t1 = copy.deepcopy(topo)
    t1.pools.append(P.PoolOp(src="l0.1", out="extra", channels=64, in_shape=by_name["backbone.l_conv.0.conv_pass.2"].out_shape,
                             factor=(1, 2, 2)))
    assert [(a.name, b.name) for a, b in P.find_chain_pairs(t1, direct, 2)] == [
        ("backbone.r_conv.0.0.conv_pass.2", "backbone.r_conv.0.0.conv_pass.4"), ("head.0", "head.2")]
    # (2) the pair's INPUT is also a source of another convolution (a skip connection would be): dropped as well
    t2 = copy.deepcopy(topo)
    t2.convs[-1].sources.append(P.Source("r0.0", 64))
    names = [(a.name, b.name) for a, b in P.find_chain_pairs(t2, direct, 2)]
    assert ("backbone.r_conv.0.0.conv_pass.2", "backbone.r_conv.0.0.conv_pass.4") not in names
    assert ("backbone.l_conv.0.conv_pass.2", "backbone.l_conv.0.conv_pass.4") in names
    # a Winograd layer is never half of a pair
    wino = dict(direct, **{"head.0": 2})
    assert ("head.0", "head.2") not in [(a.name, b.name) for a, b in P.find_chain_pairs(topo, wino, 2)]


def test_infer_chunk_sizes_and_shared_device_rule(monkeypatch):
    """UNetModel.infer_chunk: copies per forward by pixel count (eight 528^2 tiles' worth, a divisor of the copies; all of
    them for small tiles; max_infer_batch overrides).  parallel.ranks_sharing_device: 1 unless the test hook
    CLX_LOCAL_DEVICE pins the host's ranks to one device."""
    from cellulus_amd import parallel
    from cellulus_amd.models import get_model

    m = get_model(in_channels=1, out_channels=2, num_fmaps=8, fmap_inc_factor=2, features_in_last_layer=8,
                  downsampling_factors=[(2, 2)], num_spatial_dims=2)
    assert m.infer_chunk(32, (528, 528)) == 8
    assert m.infer_chunk(32, (272, 272)) == 16
    assert m.infer_chunk(32, (1040, 1040)) == 2
    assert m.infer_chunk(32, (4000, 4000)) == 1
    assert m.infer_chunk(4, (56, 56)) == 4
    assert m.infer_chunk(6, (400, 400)) == 6
    assert m.infer_chunk(32, (64, 64, 64)) == 8
    assert m.infer_chunk(30, (528, 528)) == 6            # a divisor of the copies
    m.max_infer_batch = 3
    assert m.infer_chunk(32, (528, 528)) == 3 and m.infer_chunk(2, (16, 16)) == 2
    for k in ("CLX_LOCAL_DEVICE", "LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    assert parallel.ranks_sharing_device() == 1
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert parallel.ranks_sharing_device() == 1          # eight ranks, eight devices
    monkeypatch.setenv("CLX_LOCAL_DEVICE", "0")
    assert parallel.ranks_sharing_device() == 8
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    assert parallel.ranks_sharing_device() == 4


# ------------------------------------------------- pair sampler: numpy's legacy stream restated in libclx
def test_native_pair_offsets_are_numpys_stream_value_for_value_and_leave_its_state():
    """clx_sample_offsets_mt19937 (host function: MT19937 + numpy's masked rejection + the reference's filter) against
    np.random itself: the same offsets, and the global generator in the same state afterwards — so a seeded run draws the
    reference's pairs whichever form ran (zarr_dataset.py:185-198)."""
    from cellulus_amd.datasets.zarr_dataset import ZarrDataset

    class Bare(ZarrDataset):
        def __init__(self, nd):
            self.num_spatial_dims = nd

    for nd in (2, 3):
        d = Bare(nd)
        for seed in (0, 3, 12345):
            for radius, n in ((10, 196850 if nd == 2 else 20000), (3 if nd == 2 else 5, 1000), (10.0, 5000),
                              (4 if nd == 2 else 6, 17), (40, 3000), (10, 1), (10, 311), (10, 312), (10, 313)):
                np.random.seed(seed)
                np.random.rand(seed % 7 + 1)                   # somewhere inside a block of the generator
                want = d._sample_offsets_numpy(radius, n)
                state_want = np.random.get_state()
                np.random.seed(seed)
                np.random.rand(seed % 7 + 1)
                got = d.sample_offsets_within_radius(radius, n)
                state_got = np.random.get_state()
                assert got.dtype == want.dtype and got.shape == want.shape
                np.testing.assert_array_equal(got, want)
                assert state_got[2] == state_want[2] and np.array_equal(state_got[1], state_want[1])
                assert np.random.randint(0, 1 << 30) == (np.random.set_state(state_want) or np.random.randint(0, 1 << 30))
    # a Gaussian draw cached in the generator survives the hand-over
    np.random.seed(5)
    np.random.standard_normal(1)
    before = np.random.get_state()
    assert before[3] == 1
    Bare(2).sample_offsets_within_radius(10, 100)
    after = np.random.get_state()
    assert after[3] == 1 and after[4] == before[4]


def test_planes_geometry_and_argument_validation_without_gpu():
    """The P3 plane format of the split-precision products (include/clx.h, csrc/sp_planes.h): 6 bytes per element, rows
    padded to a multiple of 64 and at least 128; the entry points refuse what the kernels do not cover before any launch."""
    from cellulus_amd import _clx

    lib = _clx.load()
    assert lib.clx_planes_bytes(1, 16) == 128 * 16 * 6
    assert lib.clx_planes_bytes(128, 64) == 128 * 64 * 6
    assert lib.clx_planes_bytes(129, 64) == 192 * 64 * 6
    assert lib.clx_planes_bytes(516128, 256) == (516128 + 63) // 64 * 64 * 256 * 6
    assert lib.clx_planes_bytes(100, 40) == 0 and lib.clx_planes_bytes(0, 64) == 0
    null = ctypes.c_void_p(0)
    p = ctypes.c_void_p(4096)
    with pytest.raises(_clx.ClxError, match="null"):
        _clx.call("clx_split_planes", null, 64, 10, 64, p, null)
    with pytest.raises(_clx.ClxError, match="multiple of 16"):
        _clx.call("clx_split_planes", p, 64, 10, 40, p, null)
    with pytest.raises(_clx.ClxError, match="N %% 128|N % 128"):
        _clx.call("clx_gemm_planes", p, p, 64, 96, 128, null, 0, p, 96, null)
    with pytest.raises(_clx.ClxError, match="K"):
        _clx.call("clx_gemm_planes", p, p, 64, 128, 64, null, 0, p, 128, null)
    with pytest.raises(_clx.ClxError, match="128"):
        _clx.call("clx_wgrad_planes", p, p, 1000, 128, 64, p, 64, null)
    d = _clx.ClxConvDesc()
    assert lib.clx_conv_sp_covers(ctypes.byref(d)) == 0
    # the switch of the Python layer
    from cellulus_amd.models import plan as P

    assert P.DEFAULT_PRECISION == "f32x3bf16"
