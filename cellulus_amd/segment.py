"""Segmentation post-processing — drop-in for ``cellulus/segment.py:13-108``.

``post_processing="cell"``: grow by ``grow_distance`` / shrink by
``shrink_distance`` with exact integer squared distance transforms on the
device, then the size filter (connected components, drop small, relabel in
raster order) — all integer work, bit-exact.  ``"nucleus"``: bounding boxes,
the per-instance Otsu histograms (numpy.histogram's arithmetic in the raw
dtype) and ``binary_fill_holes`` run on the device for all instances at once
(``csrc/nucleus.hip``); only the 256-bin variance scan per instance is host
numpy.  Both are pinned against the real reference stage by
``tests/golden/g8_stages.npz``.
"""

import numpy as np
import torch
from tqdm import tqdm

from . import _clx, parallel
from .configs.inference_config import InferenceConfig
from .datasets.meta_data import DatasetMetaData
from .utils import zarr_io
from .utils.misc import label_on_device
from .utils.otsu import otsu_from_centers


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


_RAW_F32, _RAW_F64, _RAW_I32 = 0, 1, 2          # clx_raw_type


def _raw_to_device(raw_image, device):
    """Raw intensities keep their arithmetic type (the Otsu histogram is dtype dependent):
    float32 / float64 as they are, integer images as int32."""
    raw_image = np.ascontiguousarray(raw_image)
    if raw_image.dtype == np.float32:
        return torch.from_numpy(raw_image).to(device), _RAW_F32
    if raw_image.dtype == np.float64:
        return torch.from_numpy(raw_image).to(device), _RAW_F64
    if raw_image.dtype.kind in "ui":
        if raw_image.size and (int(raw_image.min()) < -2 ** 31 or int(raw_image.max()) >= 2 ** 31):
            raise ValueError("nucleus post-processing: integer raw values must fit in int32")
        return torch.from_numpy(raw_image.astype(np.int32)).to(device), _RAW_I32
    raise TypeError(f"nucleus post-processing: unsupported raw dtype {raw_image.dtype}")


def _decode_keys(keys, raw_type):
    """clx_inst_stats order-preserving keys -> values in the raw dtype."""
    keys = keys.astype(np.uint64)
    if raw_type == _RAW_I32:
        return (keys.astype(np.uint32) ^ np.uint32(0x80000000)).view(np.int32)
    if raw_type == _RAW_F32:
        k = keys.astype(np.uint32)
        neg = (k & np.uint32(0x80000000)) == 0
        return np.where(neg, ~k, k & np.uint32(0x7FFFFFFF)).astype(np.uint32).view(np.float32)
    neg = (keys & np.uint64(1 << 63)) == 0
    return np.where(neg, ~keys, keys & np.uint64((1 << 63) - 1)).astype(np.uint64).view(np.float64)


def nucleus_refine_on_device(seg, raw, raw_type):
    """segment.py:52-101 for one label image.  seg: int32 device tensor (2-D or 3-D), raw: device
    tensor of the same shape (float32 / float64 / int32 as `raw_type` says).  Returns a new int32
    tensor: per instance, pixels brighter than the instance's Otsu threshold with the holes
    inside its bounding box filled; later ids overwrite earlier ones."""
    _clx.require_device(seg, "segmentation")
    assert seg.dtype == torch.int32 and seg.is_contiguous() and seg.ndim in (2, 3)
    assert raw.shape == seg.shape and raw.is_contiguous() and raw.device == seg.device
    nd = seg.ndim
    Z, Y, X = (1,) * (3 - nd) + tuple(seg.shape)
    st = _clx.stream_ptr(seg.device)
    out = torch.zeros_like(seg)
    nid = int(seg.max().item()) + 1 if seg.numel() else 1
    if nid <= 1:
        return out
    bbox = torch.empty((nid, 6), dtype=torch.int32, device=seg.device)
    vkey = torch.empty((nid, 2), dtype=torch.int64, device=seg.device)
    _clx.call("clx_inst_stats", _clx.ptr(seg), _clx.ptr(raw), raw_type, Z, Y, X, nid,
              _clx.ptr(bbox), _clx.ptr(vkey), st)
    bbox_h = bbox.cpu().numpy()
    vals = _decode_keys(vkey.cpu().numpy().view(np.uint64), raw_type)       # (nid, 2) min, max
    present = np.flatnonzero(bbox_h[:, 3] >= 0)
    present = present[present != 0]
    # a constant instance has threshold == its value (skimage returns the first pixel), so
    # `raw > threshold` is empty and nothing is written for it
    ids = present[vals[present, 0] != vals[present, 1]].astype(np.int32)
    n = len(ids)
    if n == 0:
        return out
    lo, hi = vals[ids, 0], vals[ids, 1]
    slot = np.full(nid, -1, dtype=np.int32)
    slot[ids] = np.arange(n, dtype=np.int32)
    if raw_type == _RAW_I32:
        widths = (hi.astype(np.int64) - lo.astype(np.int64) + 1)
        nbins = int(widths.max())
        if n * nbins > (1 << 28):
            raise ValueError("nucleus post-processing: integer intensity range too wide "
                             f"({n} instances x {nbins} values)")
        edges_or_min = np.ascontiguousarray(lo.astype(np.int32))
    else:
        nbins = 256
        widths = np.full(n, nbins)
        # numpy.histogram's own edges, in the raw dtype (histograms.py: np.linspace(first, last, bins+1))
        edges_or_min = np.stack([np.linspace(a, b, nbins + 1, endpoint=True, dtype=vals.dtype)
                                 for a, b in zip(lo, hi)])
    counts = torch.zeros((n, nbins), dtype=torch.int32, device=seg.device)
    aux = torch.from_numpy(edges_or_min).to(seg.device)
    slot_d = torch.from_numpy(slot).to(seg.device)
    _clx.call("clx_inst_histogram", _clx.ptr(seg), _clx.ptr(raw), raw_type, seg.numel(), _clx.ptr(slot_d),
              nid, _clx.ptr(aux), nbins, _clx.ptr(counts), st)
    counts_h = counts.cpu().numpy()
    thr = np.empty(n, dtype=np.float64)
    for k in range(n):
        if raw_type == _RAW_I32:
            centers = np.arange(int(lo[k]), int(hi[k]) + 1)
            thr[k] = otsu_from_centers(counts_h[k, :int(widths[k])], centers)
        else:
            e = edges_or_min[k]
            thr[k] = otsu_from_centers(counts_h[k], (e[:-1] + e[1:]) / 2.0)
    box = bbox_h[ids].astype(np.int64)
    vol = (box[:, 3] - box[:, 0] + 1) * (box[:, 4] - box[:, 1] + 1) * (box[:, 5] - box[:, 2] + 1)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(vol, out=off[1:])
    scratch = torch.empty(int(off[-1]) + 16, dtype=torch.uint8, device=seg.device)
    ids_d = torch.from_numpy(ids).to(seg.device)
    thr_d = torch.from_numpy(thr).to(seg.device)
    off_d = torch.from_numpy(off).to(seg.device)
    _clx.call("clx_inst_refine", _clx.ptr(seg), _clx.ptr(raw), raw_type, nd, Z, Y, X, _clx.ptr(ids_d),
              _clx.ptr(bbox), _clx.ptr(thr_d), _clx.ptr(off_d), _clx.ptr(scratch), n, _clx.ptr(out), st)
    return out


def segment(inference_config: InferenceConfig) -> None:
    dataset_config = inference_config.dataset_config
    meta = DatasetMetaData.from_dataset_config(dataset_config)
    nd = meta.num_spatial_dims
    device = torch.device(inference_config.device)
    if parallel.world_size() > 1:
        device = torch.device("cuda", torch.cuda.current_device())

    f = zarr_io.open(inference_config.segmentation_dataset_config.container_path)
    ds = f[inference_config.segmentation_dataset_config.secondary_dataset_name]
    def create():
        ds_new = f.create_dataset(
            inference_config.segmentation_dataset_config.dataset_name,
            shape=(meta.num_samples, inference_config.num_bandwidths, *meta.spatial_array),
            dtype=np.uint16)
        ds_new.attrs["axis_names"] = ["s", "c"] + ["t", "z", "y", "x"][-nd:]
        ds_new.attrs["resolution"] = (1,) * nd
        ds_new.attrs["offset"] = (0,) * nd

    parallel.rank0_first(create)
    ds_segmented = f[inference_config.segmentation_dataset_config.dataset_name]

    lo, hi = parallel.shard_range(meta.num_samples)
    for sample in tqdm(range(lo, hi), disable=parallel.rank() != 0):
        raw_image = None
        if inference_config.post_processing != "cell":
            raw_image = f[dataset_config.dataset_name][sample, 0]
        for bandwidth_factor in range(inference_config.num_bandwidths):
            seg_d = torch.from_numpy(ds[sample, bandwidth_factor].astype(np.int32)).to(device)
            out = segment_sample(seg_d, raw_image, inference_config, device)
            ds_segmented[sample, bandwidth_factor, ...] = out.cpu().numpy()


def segment_sample(seg_d, raw_image, inference_config, device):
    """segment.py:41-108 for one (sample, bandwidth).  seg_d: int32 device tensor holding the
    detection labels (consumed); raw_image: host array of the sample's first raw channel (only the
    "nucleus" post-processing reads it).  Returns the int32 device tensor to store as uint16."""
    if inference_config.post_processing == "cell":
        grow_shrink_on_device(seg_d, inference_config.grow_distance, inference_config.shrink_distance)
    else:  # "nucleus"
        raw_d, raw_type = _raw_to_device(raw_image, device)
        seg_d = nucleus_refine_on_device(seg_d, raw_d, raw_type)
    if inference_config.min_size == 0:          # size_filter returns its input unchanged (misc.py:12-13)
        return seg_d
    out, _ = label_on_device(seg_d, inference_config.min_size)
    return out
