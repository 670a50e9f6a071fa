"""`TrainConfig` — mirror of cellulus/configs/train_config.py:10-127 (same names, defaults, validators)."""

from typing import List

import attrs
from attrs.validators import instance_of

from .dataset_config import DatasetConfig
from .utils import to_config


@attrs.define
class TrainConfig:
    """Training hyper-parameters.

    crop_size ([252, 252]) / batch_size (8) / max_iterations (100000)
    initial_learning_rate (4e-5): Adam learning rate (no schedule).
    density (0.1): fraction of output pixels used as anchors; kappa (10.0): pair radius.
    temperature (10.0) / regularizer_weight (1e-5): OCE loss constants.
    save_model_every (1000) / save_best_model_every (100) / save_snapshot_every (1000)
    num_workers (8): loader processes.
    elastic_deform (True), control_point_spacing (64), control_point_jitter (2.0): augmentation.
    train_data_config / validate_data_config: DatasetConfig of the raw data.
    device ('cuda:0'): on ROCm 'cuda:N' is HIP device N. cellulus_amd has no CPU path.
    """

    train_data_config: DatasetConfig = attrs.field(default=None, converter=to_config(DatasetConfig))
    validate_data_config: DatasetConfig = attrs.field(default=None, converter=to_config(DatasetConfig))
    crop_size: List = attrs.field(default=[252, 252], validator=instance_of(List))
    batch_size: int = attrs.field(default=8, validator=instance_of(int))
    max_iterations: int = attrs.field(default=100_000, validator=instance_of(int))
    initial_learning_rate: float = attrs.field(default=4e-5, validator=instance_of(float))
    density: float = attrs.field(default=0.1, validator=instance_of(float))
    kappa: float = attrs.field(default=10.0, validator=instance_of(float))
    temperature: float = attrs.field(default=10.0, validator=instance_of(float))
    regularizer_weight: float = attrs.field(default=1e-5, validator=instance_of(float))
    save_model_every: int = attrs.field(default=1_000, validator=instance_of(int))
    save_best_model_every: int = attrs.field(default=100, validator=instance_of(int))
    save_snapshot_every: int = attrs.field(default=1_000, validator=instance_of(int))
    num_workers: int = attrs.field(default=8, validator=instance_of(int))
    elastic_deform: bool = attrs.field(default=True, validator=instance_of(bool))
    control_point_spacing: int = attrs.field(default=64, validator=instance_of(int))
    control_point_jitter: float = attrs.field(default=2.0, validator=instance_of(float))
    device: str = attrs.field(default="cuda:0", validator=instance_of(str))
