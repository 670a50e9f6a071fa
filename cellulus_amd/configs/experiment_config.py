"""`ExperimentConfig` — mirror of cellulus/configs/experiment_config.py:12-62."""

from datetime import datetime

import attrs
from attrs.validators import instance_of, optional

from .inference_config import InferenceConfig
from .model_config import ModelConfig
from .train_config import TrainConfig
from .utils import to_config


@attrs.define
class ExperimentConfig:
    """Top-level configuration parsed from train.toml / infer.toml.

    model_config (required), experiment_name (today's date), normalization_factor (None),
    object_size (30, must be an int), train_config, inference_config.
    """

    model_config: ModelConfig = attrs.field(converter=to_config(ModelConfig))
    experiment_name: str = attrs.field(
        default=datetime.today().strftime("%Y-%m-%d"), validator=instance_of(str))
    normalization_factor: float = attrs.field(default=None, validator=optional(instance_of(float)))
    object_size: int = attrs.field(default=30, validator=instance_of(int))
    train_config: TrainConfig = attrs.field(default=None, converter=to_config(TrainConfig))
    inference_config: InferenceConfig = attrs.field(default=None, converter=to_config(InferenceConfig))
