"""Dataset factory — same signature as cellulus/datasets/__init__.py:8-27."""

from typing import Tuple

from ..configs import DatasetConfig
from .meta_data import DatasetMetaData  # noqa: F401
from .zarr_dataset import ZarrDataset


def get_dataset(
    dataset_config: DatasetConfig,
    crop_size: Tuple[int, ...],
    elastic_deform: bool,
    control_point_spacing: int,
    control_point_jitter: float,
    density: float,
    kappa: int,
    normalization_factor: float,
) -> ZarrDataset:
    return ZarrDataset(
        dataset_config=dataset_config,
        crop_size=crop_size,
        elastic_deform=elastic_deform,
        control_point_spacing=control_point_spacing,
        control_point_jitter=control_point_jitter,
        density=density,
        kappa=kappa,
        normalization_factor=normalization_factor,
    )
