"""Otsu threshold with the histogram on the device — replaces
``skimage.filters.threshold_otsu`` as called in ``cellulus/detect.py:88-91``.

min/max and the 256-bin ``numpy.histogram`` (bit-exact index computation, see
``csrc/otsu.hip``) run on the device; the 256-element between-class-variance
scan is host numpy, written as skimage computes it."""

import numpy as np
import torch

from .. import _clx


def histogram_on_device(x, nbins=256):
    """np.histogram(x, bins=nbins) for a float64 device tensor -> (counts int64, bin_edges) on host."""
    _clx.require_device(x, "image")
    assert x.dtype == torch.float64
    x = x.contiguous()
    st = _clx.stream_ptr(x.device)
    mm = torch.empty(2, dtype=torch.float64, device=x.device)
    _clx.call("clx_minmax_f64", _clx.ptr(x), x.numel(), _clx.ptr(mm), st)
    lo, hi = mm.cpu().tolist()
    if lo == hi:                       # numpy widens a degenerate range by +-0.5
        lo, hi = lo - 0.5, hi + 0.5
    edges = np.linspace(lo, hi, nbins + 1, endpoint=True, dtype=np.float64)
    edges_d = torch.from_numpy(edges).to(x.device)
    counts = torch.zeros(nbins, dtype=torch.int64, device=x.device)
    _clx.call("clx_histogram_f64", _clx.ptr(x), x.numel(), _clx.ptr(edges_d), nbins, _clx.ptr(counts), st)
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


def threshold_otsu(image, nbins=256):
    """image: float64 device tensor or numpy array (uploaded). Constant images return their value."""
    if not torch.is_tensor(image):
        if not torch.cuda.is_available():
            raise _clx.ClxError("threshold_otsu needs a HIP device; cellulus_amd has no CPU path")
        image = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float64)).cuda()
    image = image.to(torch.float64)
    first = image.reshape(-1)[0]
    if bool((image == first).all()):
        return float(first)
    counts, edges = histogram_on_device(image, nbins)
    return float(otsu_from_histogram(counts, edges))
