"""Segmentation post-processing — drop-in for ``cellulus/segment.py:13-108``.

``post_processing="cell"``: grow by ``grow_distance`` / shrink by
``shrink_distance`` with exact integer squared distance transforms on the
device, then the size filter (connected components, drop small, relabel in
raster order) — all integer work, bit-exact.  ``"nucleus"``: the per-instance
intensity refinement keeps the reference's own library call
(``scipy.ndimage.binary_fill_holes``) on the host; only its final size filter
runs on the device (listed as not yet accelerated in DESIGN.md).
"""

import numpy as np
import torch
from tqdm import tqdm

from . import _clx, parallel
from .configs.inference_config import InferenceConfig
from .datasets.meta_data import DatasetMetaData
from .utils import zarr_io
from .utils.misc import label_on_device
from .utils.otsu import otsu_from_histogram


def grow_shrink_on_device(seg, grow_distance, shrink_distance):
    """seg: int32 device tensor, modified in place (segment.py:41-51)."""
    _clx.require_device(seg, "segmentation")
    assert seg.dtype == torch.int32 and seg.is_contiguous() and seg.ndim in (2, 3)
    Z, Y, X = (1,) * (3 - seg.ndim) + tuple(seg.shape)
    npix = Z * Y * X
    ws = torch.empty(9 * npix + 128, dtype=torch.uint8, device=seg.device)
    _clx.call("clx_grow_shrink", _clx.ptr(seg), Z, Y, X, int(grow_distance), int(shrink_distance),
              _clx.ptr(ws), _clx.stream_ptr(seg.device))
    return seg


def _host_otsu(values):
    """skimage.filters.threshold_otsu on a 1-D sample: integer data use one bin per value,
    float data 256 bins over [min, max]."""
    values = np.asarray(values)
    if values.size == 0:
        raise ValueError("empty instance")
    if np.all(values == values.flat[0]):
        return values.flat[0]
    if values.dtype.kind in "ui":
        lo, hi = int(values.min()), int(values.max())
        counts = np.bincount(values.astype(np.int64).ravel() - lo, minlength=hi - lo + 1)
        centers = np.arange(lo, hi + 1)
        edges = np.concatenate([centers - 0.5, [hi + 0.5]])
        return otsu_from_histogram(counts, edges)
    counts, edges = np.histogram(values, bins=256)
    return otsu_from_histogram(counts, edges)


def segment(inference_config: InferenceConfig) -> None:
    dataset_config = inference_config.dataset_config
    meta = DatasetMetaData.from_dataset_config(dataset_config)
    nd = meta.num_spatial_dims
    device = torch.device(inference_config.device)
    if parallel.world_size() > 1:
        device = torch.device("cuda", torch.cuda.current_device())

    f = zarr_io.open(inference_config.segmentation_dataset_config.container_path)
    ds = f[inference_config.segmentation_dataset_config.secondary_dataset_name]
    if parallel.rank() == 0:
        ds_new = f.create_dataset(
            inference_config.segmentation_dataset_config.dataset_name,
            shape=(meta.num_samples, inference_config.num_bandwidths, *meta.spatial_array),
            dtype=np.uint16)
        ds_new.attrs["axis_names"] = ["s", "c"] + ["t", "z", "y", "x"][-nd:]
        ds_new.attrs["resolution"] = (1,) * nd
        ds_new.attrs["offset"] = (0,) * nd
    if parallel.world_size() > 1:
        torch.distributed.barrier()
    ds_segmented = f[inference_config.segmentation_dataset_config.dataset_name]
    min_size = inference_config.min_size

    lo, hi = parallel.shard_range(meta.num_samples)
    for sample in tqdm(range(lo, hi), disable=parallel.rank() != 0):
        for bandwidth_factor in range(inference_config.num_bandwidths):
            segmentation = ds[sample, bandwidth_factor]
            if inference_config.post_processing == "cell":
                seg_d = torch.from_numpy(segmentation.astype(np.int32)).to(device)
                grow_shrink_on_device(seg_d, inference_config.grow_distance,
                                      inference_config.shrink_distance)
            else:  # "nucleus"
                refined = _nucleus_refine(segmentation, f[dataset_config.dataset_name][sample, 0], nd)
                seg_d = torch.from_numpy(refined.astype(np.int32)).to(device)
            if min_size == 0:          # size_filter returns its input unchanged (misc.py:12-13)
                out = seg_d
            else:
                out, _ = label_on_device(seg_d, min_size)
            ds_segmented[sample, bandwidth_factor, ...] = out.cpu().numpy()


def _nucleus_refine(segmentation, raw_image, nd):
    """segment.py:52-101 — per instance: Otsu on the raw intensities inside the mask, keep the
    brighter part, fill holes inside the bounding box.  Only refined pixels are written."""
    from scipy.ndimage import binary_fill_holes

    out = np.zeros_like(segmentation)
    ids = np.unique(segmentation)
    for id_ in ids[ids != 0]:
        m = segmentation == id_
        where = np.where(m)
        box = tuple(slice(int(w.min()), int(w.max()) + 1) for w in where)
        threshold = _host_otsu(raw_image[m])
        mask = m & (raw_image > threshold)
        mask[box] = binary_fill_holes(mask[box])
        out[mask] = id_
    return out
