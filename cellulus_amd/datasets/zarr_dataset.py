"""`ZarrDataset` — random crops + pair coordinates, the producer side of
``train_iteration`` (cellulus/datasets/zarr_dataset.py:12-251).

The reference builds a gunpowder pipeline (ZarrSource -> RandomLocation ->
Normalize [-> ElasticAugment]); gunpowder and zarr are absent here, so the
pipeline is restated on numpy/scipy: uniform random sample + spatial offset,
``float32(data) * factor`` normalisation, optional elastic deformation
(jittered control-point grid + rotation in [0, pi/2] + scale in [0.9, 1.1],
linear interpolation), rejection of all-zero crops (zarr_dataset.py:139-158).
The pair sampler (`sample_coordinates`, `sample_offsets_within_radius`) follows
zarr_dataset.py:177-251 call for call on the global numpy RNG, so seeding
``np.random`` reproduces the reference's coordinates.
"""

import math
import os
import random
from typing import Tuple

import numpy as np
from torch.utils.data import IterableDataset

from ..configs import DatasetConfig
from ..utils import zarr_io
from .meta_data import DatasetMetaData


def default_normalization_factor(dtype):
    """gunpowder.Normalize(factor=None): scale integer types to [0, 1]."""
    dtype = np.dtype(dtype)
    if dtype == np.uint8:
        return 1.0 / 255
    if dtype == np.uint16:
        return 1.0 / 65535
    if dtype.kind == "f":
        return 1.0
    raise RuntimeError(f"automatic normalization is not implemented for dtype {dtype}; "
                       "set normalization_factor")


_BLAS_LIMIT = None


def _single_threaded_blas():
    """Inside a loader process: numpy's BLAS on ONE thread.  The augmentation's small matrix products (a 2 x 2 rotation
    over 65 536 grid points, the spline matrices over the jitter grids) are enough for OpenBLAS to wake its whole
    pool — one thread per host core, per loader process — and eight such pools spinning beside each other made the
    crops' times jitter: `train()` lost 4 % against the step rate on some hosts and nothing on others (round 5)."""
    global _BLAS_LIMIT
    if _BLAS_LIMIT is None:
        try:
            from threadpoolctl import threadpool_limits

            _BLAS_LIMIT = threadpool_limits(limits=1)
        except Exception:            # (threadpoolctl absent: nothing to limit with)
            _BLAS_LIMIT = False


class ZarrDataset(IterableDataset):  # type: ignore
    def __init__(
        self,
        dataset_config: DatasetConfig,
        crop_size: Tuple[int, ...],
        elastic_deform: bool,
        control_point_spacing: int,
        control_point_jitter: float,
        density: float,
        kappa: float,
        normalization_factor: float,
    ):
        self.dataset_config = dataset_config
        self.crop_size = tuple(crop_size)
        self.elastic_deform = elastic_deform
        self.control_point_spacing = control_point_spacing
        self.control_point_jitter = control_point_jitter
        self.normalization_factor = normalization_factor
        self.__read_meta_data()
        assert len(crop_size) == self.num_spatial_dims, (
            f'"crop_size" must have the same dimension as the spatial(temporal) dimensions of the '
            f'"{self.dataset_config.dataset_name}" dataset which is {self.num_spatial_dims}, '
            f"but it is {crop_size}")
        self.density = density
        self.kappa = kappa
        self.output_shape = tuple(int(_ - 16) for _ in self.crop_size)
        self.unbiased_shape = tuple(int(_ - (2 * self.kappa)) for _ in self.output_shape)
        self._array = None
        # opt-in (CLX_DEVICE_PAIRS=1, set by train()): the crops come without coordinates and the pairs
        # are drawn on the device (DevicePairSampler) — same distribution, not the reference's stream
        self.skip_pairs = False

    def __iter__(self):
        import torch.utils.data

        if torch.utils.data.get_worker_info() is not None:
            _single_threaded_blas()
        return iter(self.__yield_sample())

    # ------------------------------------------------------------------ source
    def _open(self):
        if self._array is None:
            container = zarr_io.open(self.dataset_config.container_path, "r")
            self._array = container[self.dataset_config.dataset_name]
            for size, crop in zip(self._array.shape[2:], self.crop_size):
                if size < crop:
                    raise RuntimeError(f"crop_size {self.crop_size} exceeds the dataset's spatial "
                                       f"extent {self._array.shape[2:]}")
        return self._array

    def _random_crop(self):
        arr = self._open()
        s = random.randint(0, arr.shape[0] - 1)
        factor = self.normalization_factor
        if factor is None:
            factor = default_normalization_factor(arr.dtype)
        spatial = arr.shape[2:]
        if not self.elastic_deform:
            off = [random.randint(0, n - c) for n, c in zip(spatial, self.crop_size)]
            sl = (s, slice(None)) + tuple(slice(o, o + c) for o, c in zip(off, self.crop_size))
            return arr[sl].astype(np.float32) * np.float32(factor)
        return self._elastic_crop(arr, s, factor)

    def _elastic_ops(self):
        """What the elastic augmentation needs per crop and never changes: the crop's pixel grid relative to its centre,
        and — cubic-spline up-sampling being linear and separable — one (crop extent x control points) matrix per axis
        that maps a control-point grid to its up-sampled field, obtained from scipy's ``zoom`` itself on unit vectors
        (the same numbers as zooming every crop's jitter grid, which cost 8 ms per axis and crop)."""
        ops = getattr(self, "_elastic_cache", None)
        if ops is None:
            from scipy.ndimage import zoom

            nd = self.num_spatial_dims
            grid = np.stack(np.meshgrid(*[np.arange(c, dtype=np.float64) for c in self.crop_size], indexing="ij"))
            centre = (np.asarray(self.crop_size, dtype=np.float64) - 1) / 2
            rel0 = grid - centre.reshape((nd,) + (1,) * nd)
            cp_shape = [max(2, int(math.ceil(c / self.control_point_spacing)) + 1) for c in self.crop_size]
            mats = [zoom(np.eye(p), [c / p, 1.0], order=3, mode="nearest", grid_mode=False)[:c]
                    for c, p in zip(self.crop_size, cp_shape)]
            ops = self._elastic_cache = (rel0, cp_shape, mats)
        return ops

    def _elastic_crop(self, arr, s, factor):
        from scipy.ndimage import map_coordinates

        nd = self.num_spatial_dims
        rel0, cp_shape, mats = self._elastic_ops()
        spatial = np.asarray(arr.shape[2:], dtype=np.float64)
        angle = random.uniform(0, math.pi / 2)
        scale = random.uniform(0.9, 1.1)
        rot = np.eye(nd)
        c_, s_ = math.cos(angle), math.sin(angle)
        rot[-2:, -2:] = [[c_, -s_], [s_, c_]]          # rotate in the (y, x) plane
        rel = np.tensordot(rot, rel0, axes=1) * scale
        for d in range(nd):
            field = np.random.normal(0.0, self.control_point_jitter, size=cp_shape)
            for axis, m in enumerate(mats):            # up-sample axis by axis: field <- m applied along `axis`
                field = np.moveaxis(np.tensordot(m, field, axes=(1, axis)), 0, axis)
            rel[d] += field
        lo, hi = rel.reshape(nd, -1).min(axis=1), rel.reshape(nd, -1).max(axis=1)
        room = spatial - 1 - (hi - lo)
        mode = "constant" if np.all(room >= 0) else "reflect"
        origin = np.array([random.uniform(0, max(r, 0)) for r in room]) - lo
        coords = rel + origin.reshape((nd,) + (1,) * nd)
        data = arr[s].astype(np.float32) * np.float32(factor)
        out = np.stack([map_coordinates(ch, coords, order=1, mode=mode, cval=0.0) for ch in data])
        return out.astype(np.float32)

    def __yield_sample(self):
        """An infinite generator of crops."""
        while True:
            array_is_zero = True
            while array_is_zero:   # reject empty crops (zarr_dataset.py:139-158)
                sample_data = self._random_crop()
                if np.max(sample_data) <= 0.0:
                    continue
                array_is_zero = False
                if self.skip_pairs:
                    anchor_samples = reference_samples = np.zeros((0, self.num_spatial_dims), dtype=np.int64)
                else:
                    anchor_samples, reference_samples = self.sample_coordinates()
            yield sample_data, anchor_samples, reference_samples

    def __read_meta_data(self):
        meta_data = DatasetMetaData.from_dataset_config(self.dataset_config)
        self.num_dims = meta_data.num_dims
        self.num_spatial_dims = meta_data.num_spatial_dims
        self.num_channels = meta_data.num_channels
        self.num_samples = meta_data.num_samples
        self.sample_dim = meta_data.sample_dim
        self.channel_dim = meta_data.channel_dim
        self.time_dim = meta_data.time_dim

    def get_num_channels(self):
        return self.num_channels

    def get_num_spatial_dims(self):
        return self.num_spatial_dims

    # ------------------------------------------------------------ pair sampler
    def sample_offsets_within_radius(self, radius, number_offsets):
        """zarr_dataset.py:185-198.  Same offsets and the same state of numpy's global generator afterwards as the
        reference's code; drawn by libclx's host function (MT19937 + numpy's masked rejection restated in C: 20 -> 4 ms
        for a 256^2 crop's 196 850 pairs) when that generator is the legacy MT19937 and the radius an integer, by numpy
        (`_sample_offsets_numpy`) otherwise."""
        nd = self.num_spatial_dims
        if os.environ.get("CLX_NATIVE_PAIR_SAMPLER", "1") != "0" and float(radius).is_integer() and 1 <= radius < 16384:
            state = np.random.get_state(legacy=True)
            if state[0] == "MT19937":
                import ctypes

                from .. import _clx

                key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
                pos = ctypes.c_int(int(state[2]))
                offsets = np.empty((number_offsets, nd), dtype=np.int64)
                _clx.call("clx_sample_offsets_mt19937", key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos),
                          int(radius), nd, int(number_offsets), offsets.ctypes.data_as(ctypes.c_void_p), None)
                np.random.set_state(("MT19937", key, pos.value, state[3], state[4]))
                return offsets
        return self._sample_offsets_numpy(radius, number_offsets)

    def _sample_offsets_numpy(self, radius, number_offsets):
        # (the reference stacks the draws, filters twice with temporaries of the full size and slices: 19 of the 31 ms
        #  a 256^2 crop's pairs cost; the same draws in the same order, one combined mask, the survivors gathered once)
        nd = self.num_spatial_dims
        draws = [np.random.randint(-radius, radius + 1, size=nd * number_offsets) for _ in range(nd)]
        sq = draws[0] * draws[0]
        for d in draws[1:]:
            sq += d * d
        keep = np.flatnonzero((sq < radius ** 2) & (sq > 0))     # (integers: |o|_1 > 0 <=> |o|^2 > 0)
        if len(keep) < number_offsets:
            return self._sample_offsets_numpy(radius, number_offsets)
        keep = keep[:number_offsets]
        offsets = np.empty((number_offsets, nd), dtype=draws[0].dtype)
        for d in range(nd):
            offsets[:, d] = draws[d][keep]
        return offsets

    def sample_coordinates(self):
        num_anchors = self.get_num_anchors()
        num_references = self.get_num_references()
        columns = [np.random.randint(self.kappa, self.output_shape[d] - self.kappa + 1, size=num_anchors)
                   for d in range(self.num_spatial_dims)]
        anchor_coordinates = np.stack(columns, axis=1)
        anchor_samples = np.repeat(anchor_coordinates, num_references, axis=0)
        offset_in_pos_radius = self.sample_offsets_within_radius(self.kappa, len(anchor_samples))
        reference_samples = anchor_samples + offset_in_pos_radius
        return anchor_samples, reference_samples

    def get_num_anchors(self):
        return int(self.density * self.unbiased_shape[0] * self.unbiased_shape[1])

    def get_num_references(self):
        return int(self.density * self.kappa ** 2 * np.pi)

    def get_num_samples(self):
        return self.get_num_anchors() * self.get_num_references()


class DevicePairSampler:
    """Pairs with the distribution of ``ZarrDataset.sample_coordinates`` drawn by ``clx_sample_pairs``
    on the device: anchor column d uniform on the integers [kappa, output_shape[d] - kappa], every
    anchor repeated ``get_num_references()`` times, reference = anchor + an offset uniform over
    {o integer: |o|^2 < kappa^2, o != 0} (what the rejection loop of ``sample_offsets_within_radius``
    converges to).  Counter-based generator: (seed, step) reproduces a batch.  Opt-in: the numbers
    are NOT the reference's ``np.random`` stream."""

    def __init__(self, dataset, device, seed):
        import ctypes
        import itertools

        import torch

        self.device = device
        self.nd = dataset.num_spatial_dims
        kappa = float(dataset.kappa)
        self.lo = int(math.ceil(kappa))
        self.hi = [int(math.floor(s - kappa)) for s in dataset.output_shape]
        if any(h < self.lo for h in self.hi):
            raise ValueError(f"kappa={kappa} leaves no anchor position in an output of extent {dataset.output_shape}")
        r = range(int(math.ceil(-kappa)), int(math.floor(kappa)) + 1)
        table = [o for o in itertools.product(r, repeat=self.nd)
                 if sum(v * v for v in o) < kappa * kappa and sum(abs(v) for v in o) > 0]
        if not table:
            raise ValueError(f"kappa={kappa} admits no offset")
        self.offsets = torch.tensor(table, dtype=torch.int32, device=device).contiguous()
        self.num_anchors = dataset.get_num_anchors()
        self.num_refs = dataset.get_num_references()
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._hi_c = (ctypes.c_int * self.nd)(*self.hi)

    def sample(self, batch_size, step):
        import torch

        from .. import _clx

        P = self.num_anchors * self.num_refs
        anchor = torch.empty((batch_size, P, self.nd), dtype=torch.int64, device=self.device)
        reference = torch.empty_like(anchor)
        _clx.call("clx_sample_pairs", _clx.ptr(anchor), _clx.ptr(reference), _clx.ptr(self.offsets),
                  self.offsets.shape[0], batch_size, self.num_anchors, self.num_refs, self.nd, self.lo, self._hi_c,
                  self.seed, int(step) & 0xFFFFFFFFFFFFFFFF, _clx.stream_ptr(self.device))
        return anchor, reference
