"""`DatasetMetaData` — same fields and errors as cellulus/datasets/meta_data.py:8-90,
reading the ``axis_names`` attribute through the in-tree zarr-v2 reader."""

from typing import Tuple

from ..configs import DatasetConfig
from ..utils import zarr_io

_HELP = (
    "\n\nThe raw dataset should have shape (s, c, [t,] [z,] y, x), where s = # of samples, "
    "c = # of channels, t = # of frames, and z/y/x are spatial extents. The dataset should "
    'have an "axis_names" attribute that contains the names of the used axes, e.g., '
    '["s", "c", "y", "x"] for a 2D dataset.'
)


def _invalid(message):
    raise RuntimeError(message + _HELP)


class DatasetMetaData:
    def __init__(self, shape, axis_names):
        self.num_dims = len(axis_names)
        self.num_spatial_dims: int = 0
        self.num_samples: int = 0
        self.num_channels: int = 0
        self.sample_dim = None
        self.channel_dim = None
        self.time_dim = None
        self.spatial_array: Tuple[int, ...] = ()
        for dim, axis_name in enumerate(axis_names):
            if axis_name == "s":
                self.sample_dim, self.num_samples = dim, shape[dim]
            elif axis_name == "c":
                self.channel_dim, self.num_channels = dim, shape[dim]
            elif axis_name == "t":
                self.num_spatial_dims += 1
                self.time_dim = dim
            elif axis_name in ("z", "y", "x"):
                self.num_spatial_dims += 1
                self.spatial_array += (shape[dim],)
        if self.sample_dim is None:
            _invalid("dataset does not have a sample dimension")
        if self.channel_dim is None:
            _invalid("dataset does not have a channel dimension")
        if self.num_dims != len(shape):
            _invalid(f"dataset has {len(shape)} dimensions, but attribute axis_names has "
                     f"{self.num_dims} entries")

    @staticmethod
    def from_dataset_config(dataset_config: DatasetConfig) -> "DatasetMetaData":
        container = zarr_io.open(dataset_config.container_path, "r")
        try:
            data = container[dataset_config.dataset_name]
        except KeyError:
            _invalid(f"Zarr container {dataset_config.container_path} does not contain "
                     f'"{dataset_config.dataset_name}" dataset')
        try:
            axis_names = data.attrs["axis_names"]
        except KeyError:
            _invalid(f'"{dataset_config.dataset_name}" dataset in {dataset_config.container_path} '
                     'does not contain "axis_names" attribute')
        try:
            return DatasetMetaData(data.shape, axis_names)
        except RuntimeError as e:
            raise RuntimeError(f'"{dataset_config.dataset_name}" dataset in '
                               f"{dataset_config.container_path} has invalid meta-data") from e
