"""`ModelConfig` — mirror of cellulus/configs/model_config.py:10-59."""

from pathlib import Path
from typing import List

import attrs
from attrs.validators import instance_of

from .utils import to_path


@attrs.define
class ModelConfig:
    """U-Net shape and checkpoint.

    num_fmaps: feature maps at the top level.
    fmap_inc_factor: multiplier of the feature maps per level.
    features_in_last_layer (64): channels fed to the 1x1 head.
    downsampling_factors ([[2, 2]]): one factor tuple per down-sampling; sets the depth.
    checkpoint (None): path of a .pth to resume from (train) or to load (infer).
    initialize (True): Kaiming-normal initialisation of all conv weights before training.
    """

    num_fmaps: int = attrs.field(validator=instance_of(int))
    fmap_inc_factor: int = attrs.field(validator=instance_of(int))
    features_in_last_layer: int = attrs.field(default=64)
    downsampling_factors: List[List[int]] = attrs.field(default=[[2, 2]])
    checkpoint: Path = attrs.field(default=None, converter=to_path)
    initialize: bool = attrs.field(default=True, validator=instance_of(bool))
