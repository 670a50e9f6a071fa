"""`InferenceConfig` — mirror of cellulus/configs/inference_config.py:10-159."""

from typing import List

import attrs
from attrs.validators import in_, instance_of, optional

from .dataset_config import DatasetConfig
from .utils import to_config


@attrs.define
class InferenceConfig:
    """Inference settings.

    dataset_config: raw data; prediction/detection/segmentation/evaluation_dataset_config:
        where each stage reads (secondary_dataset_name) and writes (dataset_name).
    device ('cuda:0'), crop_size ([252, 252]): tile fed to the network.
    p_salt_pepper (0.01), num_infer_iterations (16): noise copies per tile (2 x iterations).
    threshold (None -> Otsu on the std channel), clustering ('meanshift' | 'greedy'),
    use_seeds (False), bandwidth (None -> 0.5 * object_size), num_bandwidths (1),
    reduction_probability (0.1), min_size (None -> from object_size),
    post_processing ('cell' | 'nucleus'), grow_distance (3), shrink_distance (6).
    """

    dataset_config: DatasetConfig = attrs.field(default=None, converter=to_config(DatasetConfig))
    prediction_dataset_config: DatasetConfig = attrs.field(default=None, converter=to_config(DatasetConfig))
    detection_dataset_config: DatasetConfig = attrs.field(default=None, converter=to_config(DatasetConfig))
    segmentation_dataset_config: DatasetConfig = attrs.field(default=None, converter=to_config(DatasetConfig))
    evaluation_dataset_config: DatasetConfig = attrs.field(default=None, converter=to_config(DatasetConfig))
    device: str = attrs.field(default="cuda:0", validator=instance_of(str))
    crop_size: List = attrs.field(default=[252, 252], validator=instance_of(List))
    p_salt_pepper = attrs.field(default=0.01, validator=instance_of(float))
    num_infer_iterations = attrs.field(default=16, validator=instance_of(int))
    threshold = attrs.field(default=None, validator=optional(instance_of(float)))
    clustering = attrs.field(default="meanshift", validator=in_(["meanshift", "greedy"]))
    use_seeds = attrs.field(default=False, validator=instance_of(bool))
    bandwidth = attrs.field(default=None, validator=optional(instance_of(float)))
    num_bandwidths = attrs.field(default=1, validator=instance_of(int))
    reduction_probability = attrs.field(default=0.1, validator=instance_of(float))
    min_size = attrs.field(default=None, validator=optional(instance_of(int)))
    post_processing = attrs.field(default="cell", validator=in_(["cell", "nucleus"]))
    grow_distance = attrs.field(default=3, validator=instance_of(int))
    shrink_distance = attrs.field(default=6, validator=instance_of(int))
