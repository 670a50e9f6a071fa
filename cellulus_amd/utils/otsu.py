"""Otsu threshold with the histogram on the device — replaces
``skimage.filters.threshold_otsu`` as called in ``cellulus/detect.py:88-91``.

min/max and the 256-bin ``numpy.histogram`` (bit-exact index computation, see
``csrc/otsu.hip``) run on the device; the 256-element between-class-variance
scan is host numpy, written as skimage computes it."""

import numpy as np
import torch

from .. import _clx


def _aligned(x):
    """contiguous and 16-byte aligned (the kernels load 16 bytes per lane; a channel plane of an odd-sized image
    starts in the middle of such a group)"""
    x = x.contiguous()
    return x.clone() if x.data_ptr() % 16 else x


def minmax_on_device(x):
    """(min, max) of a float64 — or float32, read as float64 — device tensor, as host floats."""
    _clx.require_device(x, "image")
    assert x.dtype in (torch.float64, torch.float32)
    x = _aligned(x)
    mm = torch.empty(_clx.MINMAX_DOUBLES, dtype=torch.float64, device=x.device)      # results + block partials
    _clx.call("clx_minmax_f64" if x.dtype == torch.float64 else "clx_minmax_f32", _clx.ptr(x), x.numel(), _clx.ptr(mm),
              _clx.stream_ptr(x.device))
    lo, hi = mm[:2].cpu().tolist()
    return lo, hi


def histogram_on_device(x, nbins=256, minmax=None):
    """np.histogram(x, bins=nbins) for a float64 device tensor — or a float32 one whose values are read as float64 (the
    widening is exact) — -> (counts int64, bin_edges) on host.  minmax: (min, max) of x if the caller knows them."""
    _clx.require_device(x, "image")
    assert x.dtype in (torch.float64, torch.float32)
    x = _aligned(x)
    st = _clx.stream_ptr(x.device)
    lo, hi = minmax if minmax is not None else minmax_on_device(x)
    if lo == hi:                       # numpy widens a degenerate range by +-0.5
        lo, hi = lo - 0.5, hi + 0.5
    edges = np.linspace(lo, hi, nbins + 1, endpoint=True, dtype=np.float64)
    edges_d = torch.from_numpy(edges).to(x.device)
    counts = torch.empty(nbins, dtype=torch.int64, device=x.device)
    _clx.zero_many(counts)
    _clx.call("clx_histogram_f64" if x.dtype == torch.float64 else "clx_histogram_f32", _clx.ptr(x), x.numel(),
              _clx.ptr(edges_d), nbins, _clx.ptr(counts), st)
    return counts.cpu().numpy(), edges


def otsu_from_histogram(counts, bin_edges):
    return otsu_from_centers(counts, (bin_edges[:-1] + bin_edges[1:]) / 2.0)


def otsu_from_centers(counts, bin_centers):
    """skimage filters/thresholding.py:333-350 on a (counts, bin_centers) histogram."""
    counts = counts.astype(float)
    weight1 = np.cumsum(counts)
    weight2 = np.cumsum(counts[::-1])[::-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        mean1 = np.cumsum(counts * bin_centers) / weight1
        mean2 = (np.cumsum((counts * bin_centers)[::-1]) / weight2[::-1])[::-1]
    variance12 = weight1[:-1] * weight2[1:] * (mean1[:-1] - mean2[1:]) ** 2
    idx = np.argmax(variance12)
    return bin_centers[idx]


def threshold_otsu(image, nbins=256, minmax=None):
    """image: float64 device tensor, float32 device tensor (its values are read as float64: what the staged path
    reads back from the float64 `embeddings` dataset) or numpy array (uploaded as float64).  Constant images return
    their value (skimage.filters.threshold_otsu: ``if np.all(image == first_pixel): return first_pixel``).
    minmax: the image's (min, max) as host floats or a 2-element device tensor, if the caller already has them
    (clx_noise_stats_minmax emits them with the std plane): the image is then read once."""
    if not torch.is_tensor(image):
        if not torch.cuda.is_available():
            raise _clx.ClxError("threshold_otsu needs a HIP device; cellulus_amd has no CPU path")
        image = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float64)).cuda()
    if image.dtype not in (torch.float64, torch.float32):
        image = image.to(torch.float64)
    if minmax is None:
        lo, hi = minmax_on_device(image)
    elif torch.is_tensor(minmax):
        lo, hi = (float(v) for v in minmax[:2].cpu().tolist())   # float32 -> Python float: exact
    else:
        lo, hi = float(minmax[0]), float(minmax[1])
    if not (np.isfinite(lo) and np.isfinite(hi)):
        # numpy's histogram (skimage.filters.threshold_otsu -> np.histogram) refuses such an image too; the two range
        # paths treat a NaN differently (fmin / fmax drop it, the bit-pattern maximum of clx_noise_stats_minmax exposes
        # it), so neither may go on silently
        raise ValueError(f"threshold_otsu: the image's range [{lo}, {hi}] is not finite")
    if lo == hi:                 # min == max: every pixel equals the first one
        return lo
    counts, edges = histogram_on_device(image, nbins, minmax=(lo, hi))
    return float(otsu_from_histogram(counts, edges))
