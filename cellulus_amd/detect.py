"""Detection stage — drop-in for ``cellulus/detect.py:14-192``.

Per sample: Otsu threshold of the std channel (device histogram), foreground
mask, centred embeddings (side output), then per bandwidth the mean-shift
clustering on the device.  Writes the same three zarr datasets as the
reference (``detection`` uint16, ``binary-segmentation`` uint16,
``centered-embeddings`` float64).  Samples are independent: ranks take blocks
of samples, no collective.
"""

import numpy as np
import torch
from tqdm import tqdm

from . import parallel
from .configs.inference_config import InferenceConfig
from .datasets.meta_data import DatasetMetaData
from .utils import zarr_io
from .utils.mean_shift import mean_shift_on_device
from .utils.otsu import threshold_otsu


def _create(f, name, shape, dtype, nd):
    ds = f.create_dataset(name, shape=shape, dtype=dtype)
    ds.attrs["axis_names"] = ["s", "c"] + ["t", "z", "y", "x"][-nd:]
    ds.attrs["resolution"] = (1,) * nd
    ds.attrs["offset"] = (0,) * nd
    return ds


def peak_local_max(image):
    """skimage.feature.peak_local_max(image) defaults (min_distance=1, exclude_border=1,
    threshold = image.min()) restated on scipy: coordinates of strict-threshold local maxima
    of a 3^D neighbourhood, sorted by decreasing intensity.  Pinned against scikit-image
    0.18.3 by tests/golden/g5_skimage.npz (tests/test_cpu_host.py)."""
    from scipy.ndimage import maximum_filter

    size = 3
    mx = maximum_filter(image, size=size, mode="nearest")
    mask = (image == mx) & (image > image.min())
    for ax in range(image.ndim):          # exclude_border=True -> min_distance pixels
        sl = [slice(None)] * image.ndim
        sl[ax] = slice(0, 1)
        mask[tuple(sl)] = False
        sl[ax] = slice(-1, None)
        mask[tuple(sl)] = False
    coords = np.stack(np.nonzero(mask), axis=1)
    order = np.argsort(-image[mask], kind="stable")
    return coords[order]


def gaussian_weights(sigma, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, radius) with radius = int(truncate * sigma
    + 0.5), centre first (the kernel is symmetric): the weights gaussian_filter correlates with."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    phi = phi / phi.sum()
    return np.ascontiguousarray(phi[radius:]), radius


def seeds_on_device(emb_d, nd):
    """detect.py:128-132 on the device: emb_d (ND, *spatial) float64 device tensor of centred
    embeddings -> seeds (n, ND) int64 host array in (x, y[, z]) order, sorted like
    np.flip(peak_local_max(-gaussian_filter(norm(emb, axis=0), sigma=2)), 1)."""
    from . import _clx

    spatial = tuple(emb_d.shape[1:])
    Z, Y, X = (1,) * (3 - nd) + spatial
    npix = Z * Y * X
    dev = emb_d.device
    st = _clx.stream_ptr(dev)
    emb_d = emb_d.contiguous()
    mag = torch.empty(npix, dtype=torch.float64, device=dev)
    smooth = torch.empty_like(mag)
    tmp = torch.empty_like(mag)
    w, radius = gaussian_weights(2.0)
    w_d = torch.from_numpy(w).to(dev)
    _clx.call("clx_offset_magnitude", _clx.ptr(emb_d), _clx.ptr(mag), nd, npix, st)
    _clx.call("clx_gaussian_filter_f64", _clx.ptr(mag), _clx.ptr(smooth), _clx.ptr(tmp), Z, Y, X,
              _clx.ptr(w_d), radius, st)
    _clx.call("clx_negate_f64", _clx.ptr(smooth), _clx.ptr(mag), npix, st)        # mag := -smooth
    if min(spatial) < 3:
        return np.zeros((0, nd), dtype=np.int64)             # every pixel is on the excluded border
    mm = torch.empty(_clx.MINMAX_DOUBLES, dtype=torch.float64, device=dev)       # [0..1] min, max (+ block partials)
    _clx.call("clx_minmax_f64", _clx.ptr(mag), npix, _clx.ptr(mm), st)
    capacity = max(1024, npix // 9 + 16)                     # a 3^ND maximum per 3^ND block at most ...
    peaks = torch.empty(capacity, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    _clx.call("clx_peak_local_max", _clx.ptr(mag), Z, Y, X, _clx.ptr(mm), _clx.ptr(peaks), capacity,
              _clx.ptr(count), st)
    n = int(count.item())
    if n > capacity:                                         # ... unless plateaus: rerun with room for all
        capacity = n
        peaks = torch.empty(capacity, dtype=torch.int32, device=dev)
        _clx.call("clx_peak_local_max", _clx.ptr(mag), Z, Y, X, _clx.ptr(mm), _clx.ptr(peaks), capacity,
                  _clx.ptr(count), st)
    idx = np.sort(peaks[:n].cpu().numpy().astype(np.int64))          # raster order = np.nonzero order
    vals = mag[torch.from_numpy(idx).to(dev)].cpu().numpy() if n else np.zeros(0)
    order = np.argsort(-vals, kind="stable")
    coords = np.stack(np.unravel_index(idx[order], spatial), axis=1).astype(np.int64).reshape(-1, nd)
    return np.flip(coords, 1)


def detect(inference_config: InferenceConfig) -> None:
    dataset_config = inference_config.dataset_config
    meta = DatasetMetaData.from_dataset_config(dataset_config)
    nd = meta.num_spatial_dims
    device = torch.device(inference_config.device)
    if parallel.world_size() > 1:
        device = torch.device("cuda", torch.cuda.current_device())

    f = zarr_io.open(inference_config.detection_dataset_config.container_path)
    ds = f[inference_config.detection_dataset_config.secondary_dataset_name]
    spatial = tuple(meta.spatial_array)
    def create():
        _create(f, inference_config.detection_dataset_config.dataset_name,
                (meta.num_samples, inference_config.num_bandwidths, *spatial), np.uint16, nd)
        _create(f, "binary-segmentation", (meta.num_samples, 1, *spatial), np.uint16, nd)
        _create(f, "centered-embeddings", (meta.num_samples, nd + 1, *spatial), float, nd)

    parallel.rank0_first(create)
    ds_detection = f[inference_config.detection_dataset_config.dataset_name]
    ds_binary_segmentation = f["binary-segmentation"]
    ds_object_centered_embeddings = f["centered-embeddings"]

    lo, hi = parallel.shard_range(meta.num_samples)
    for sample in tqdm(range(lo, hi), disable=parallel.rank() != 0):
        embeddings = ds[sample]                                  # (D+1, *spatial) float64

        def emit(kind, index, value, sample=sample):      # written as soon as it exists, as the
            if kind == "binary":                           # reference does (a later bandwidth may raise)
                ds_binary_segmentation[sample, 0, ...] = value
            elif kind == "centered":
                ds_object_centered_embeddings[sample] = value
            else:
                ds_detection[sample, index, ...] = _labels_to_host(value)

        detect_sample(embeddings, inference_config, nd, device, sample, emit=emit)


def _labels_to_host(labels):
    return labels.numpy() if labels.device.type == "cpu" else labels.cpu().numpy()


def detect_sample(embeddings, inference_config, nd, device, sample=0, emb_d=None, emit=None, std_minmax=None):
    """detect.py:82-192 for ONE sample.  embeddings: (D+1, *spatial) float64 host array (what the
    ``embeddings`` dataset holds); emb_d: the same values already on the device, if the caller has
    them (the fused driver does) — as float64, or as the float32 values they were widened from;
    std_minmax: (min, max) of the std channel as a 2-element device tensor, if the caller has it.  ``emit(kind, index, value)`` receives every output the moment
    it exists — "binary" mask, "centered" embeddings, "detection" label tensor of bandwidth
    ``index`` — in the order the reference writes them.  Returns the list of label tensors."""
    if emit is None:
        def emit(kind, index, value):
            return None
    if emb_d is None:
        emb_d = torch.from_numpy(np.ascontiguousarray(embeddings)).to(device)
    std_d = emb_d[-1].contiguous()
    if inference_config.threshold is None:
        threshold = threshold_otsu(std_d, minmax=std_minmax)
    else:
        threshold = inference_config.threshold
    print(f"For sample {sample}, binary threshold {threshold} was used.")
    binary_mask = embeddings[-1] < threshold

    # centred embeddings: subtract the mean of the non-zero masked offsets per channel
    embeddings_centered = embeddings.copy()
    masked = binary_mask[np.newaxis, ...] * embeddings[:nd]
    for k in range(nd):
        ck = masked[k]
        embeddings_centered[k] -= ck[ck != 0].mean()
    emit("binary", 0, binary_mask)
    emit("centered", 0, embeddings_centered.copy())     # (the seeds path below mutates the array)
    detections = []

    if inference_config.clustering == "greedy":        # detect.py:162-192
        from .utils.greedy_cluster import Cluster2d, Cluster3d

        if nd == 3:
            cluster = Cluster3d(width=embeddings.shape[-1], height=embeddings.shape[-2],
                                depth=embeddings.shape[-3], fg_mask=binary_mask, device=device)
        else:
            cluster = Cluster2d(width=embeddings.shape[-1], height=embeddings.shape[-2],
                                fg_mask=binary_mask, device=device)
        for bandwidth_factor in range(inference_config.num_bandwidths):
            segmentation = cluster.cluster(
                prediction=embeddings, bandwidth=inference_config.bandwidth / (2 ** bandwidth_factor),
                min_object_size=inference_config.min_size)
            emit("detection", bandwidth_factor, segmentation)
            detections.append(segmentation)
        return detections

    # use_seeds keeps the reference's aliasing (detect.py:116-118,142-144): the first
    # mean_shift_segmentation call adds the pixel coordinates to `embeddings_centered` IN PLACE
    # (its input is a view), later bandwidths re-read that mutated array — magnitudes, seeds
    # and inputs then carry offsets + coordinates, exactly as in the reference (where sklearn
    # then usually raises "No point was within bandwidth ... of any seed").
    centered_aliased = True
    for bandwidth_factor in range(inference_config.num_bandwidths):
        bandwidth = inference_config.bandwidth / (2 ** bandwidth_factor)
        if inference_config.use_seeds:
            src = torch.from_numpy(np.ascontiguousarray(embeddings_centered)).to(device)
            seeds = seeds_on_device(src[:nd], nd)
            mean_d, sd_d = src[:nd].contiguous().clone(), src[-1].contiguous()
        else:
            seeds = None
            # (a float32 hand-over is not modified by the clustering: no copy; float64: the reference's in-place
            # coordinate add lands in a copy that is dropped, detect.py:155-160)
            mean_d = emb_d[:nd].contiguous()
            mean_d, sd_d = (mean_d if mean_d.dtype == torch.float32 else mean_d.clone()), std_d
        labels, _ = mean_shift_on_device(
            mean_d, sd_d, bandwidth, inference_config.reduction_probability, threshold, seeds)
        if inference_config.use_seeds and centered_aliased:
            embeddings_centered[:nd] = mean_d.cpu().numpy()      # offsets + coordinates
            centered_aliased = False
        emit("detection", bandwidth_factor, labels)
        detections.append(labels)
    return detections
