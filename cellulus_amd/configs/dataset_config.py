"""`DatasetConfig` — field-for-field mirror of cellulus/configs/dataset_config.py:7-41."""

from pathlib import Path

import attrs
from attrs.validators import instance_of, optional


@attrs.define
class DatasetConfig:
    """Where a zarr array lives.

    container_path: zarr container (directory store).
    dataset_name: array inside the container that a stage WRITES (or, for raw data, reads).
    secondary_dataset_name: array a stage READS (e.g. embeddings for detection,
        detection for segmentation, segmentation for evaluation).
    """

    container_path: Path = attrs.field(converter=Path)
    dataset_name: str = attrs.field(validator=instance_of(str))
    secondary_dataset_name: str = attrs.field(default=None, validator=optional(instance_of(str)))
