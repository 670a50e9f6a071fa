"""GPU: the train() driver end to end on a small synthetic zarr — loader, fused step,
logging, checkpoints (reference file names / dictionary keys), snapshots, resume, CLI."""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make_zarr(path):
    from cellulus_amd.utils import zarr_io

    rng = np.random.default_rng(0)
    f = zarr_io.open(path)
    raw = (rng.random((3, 1, 96, 96)) * 255).astype(np.uint8)
    f["train/raw"] = raw
    f["train/raw"].attrs["axis_names"] = ["s", "c", "y", "x"]


def _toml(container, extra_model="", max_iterations=3):
    return f"""
[model_config]
num_fmaps = 8
fmap_inc_factor = 2
features_in_last_layer = 16
{extra_model}

[train_config]
crop_size = [64, 64]
batch_size = 2
max_iterations = {max_iterations}
num_workers = 0
elastic_deform = false
kappa = 4.0
save_model_every = 2
save_best_model_every = 1
save_snapshot_every = 2

[train_config.train_data_config]
container_path = "{container}"
dataset_name = "train/raw"
"""


def test_train_driver_checkpoints_resume_and_cli(tmp_path, monkeypatch):
    import tomli
    from click.testing import CliRunner

    from cellulus_amd.cli import train as train_cli
    from cellulus_amd.configs import ExperimentConfig
    from cellulus_amd.train import train
    from cellulus_amd.utils import zarr_io

    monkeypatch.chdir(tmp_path)
    container = str(tmp_path / "data.zarr")
    _make_zarr(container)
    torch.manual_seed(0)
    np.random.seed(0)
    train(ExperimentConfig(**tomli.loads(_toml(container))))
    assert os.path.exists("models/000000.pth") and os.path.exists("models/000002.pth")
    assert os.path.exists("models/best_loss.pth")
    state = torch.load("models/000002.pth", weights_only=False)
    assert set(state) == {"iteration", "lowest_loss", "model_state_dict", "optim_state_dict", "logger_data"}
    assert state["iteration"] == 2 and len(state["logger_data"]["loss"]) == 3
    assert "backbone.l_conv.0.conv_pass.0.weight" in state["model_state_dict"]
    assert all(np.isfinite(v) for v in state["logger_data"]["loss"])
    lines = open("loss.csv").read().strip().split("\n")
    assert lines[0] == ",loss,oce_loss" and len(lines) == 4
    snap = zarr_io.open("snapshots.zarr", "r")
    assert snap["0/raw"].shape == (2, 1, 64, 64) and snap["2/prediction"].shape == (2, 2, 48, 48)
    assert snap["2/prediction"].attrs["offset"] == [8.0, 8.0]
    pred = snap["2/prediction"][...]
    assert np.abs(pred.reshape(2, 2, -1).mean(-1)).max() < 1e-4      # mean-subtracted offsets

    # the optimizer state loads into torch.optim.Adam (same state_dict layout) ...
    ref = torch.nn.ParameterList([torch.nn.Parameter(v.clone().float().cpu())
                                  for v in state["model_state_dict"].values()])
    opt = torch.optim.Adam(ref.parameters(), lr=4e-5, weight_decay=0.01)
    sd = state["optim_state_dict"]
    opt.load_state_dict({"state": {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                                   for k, v in sd["state"].items()}, "param_groups": sd["param_groups"]})
    # ... and training resumes from the checkpoint through the CLI
    open("resume.toml", "w").write(_toml(container, 'checkpoint = "models/000002.pth"', max_iterations=5))
    res = CliRunner().invoke(train_cli, ["resume.toml"])
    assert res.exit_code == 0, res.output + str(res.exception)
    assert os.path.exists("models/000004.pth")
    state2 = torch.load("models/000004.pth", weights_only=False)
    assert state2["iteration"] == 4 and len(state2["logger_data"]["loss"]) == 5
    assert int(state2["optim_state_dict"]["state"][0]["step"]) == 5


def test_train_refuses_cpu_device(tmp_path, monkeypatch):
    import tomli

    from cellulus_amd.configs import ExperimentConfig
    from cellulus_amd.train import train

    monkeypatch.chdir(tmp_path)
    container = str(tmp_path / "data.zarr")
    _make_zarr(container)
    cfg = tomli.loads(_toml(container))
    cfg["train_config"]["device"] = "cpu"
    with pytest.raises(RuntimeError, match="no CPU path"):
        train(ExperimentConfig(**cfg))
