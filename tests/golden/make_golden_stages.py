"""Golden vectors from the REAL reference stages `cellulus.detect.detect` and
`cellulus.segment.segment` (cell and nucleus post-processing), run end to end:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 \
        /opt/conda/bin/python3.9 tests/golden/make_golden_stages.py

The conda interpreter is the only one here with scikit-image (0.18.3) and it has
scikit-learn 0.24.2 + scipy 1.7.1; what it lacks is handled like this:
  * `zarr` (absent everywhere): an in-memory stand-in below — containers are dicts of
    numpy-backed arrays with `.attrs`; reads return copies, writes assign, exactly the
    subset of the zarr API the two stages touch.  It holds DATA only.
  * `attrs` (needs >= 21.3 for `import attrs`): the pure-python `attr`/`attrs` packages of
    the 3.10 interpreter are imported with that site-packages directory put on sys.path
    only for those two imports, after numpy/scipy/skimage/sklearn are bound from conda.
  * `torch` (utils/mean_shift.py uses it to move data only: from_numpy / arange / permute /
    view / contiguous / numpy — mapped onto the same numpy buffer below; utils/greedy_cluster.py
    imports it but clustering="meanshift" never calls it), `gunpowder` (imported by datasets/zarr_dataset.py, unused by detect/segment),
    `matplotlib` (plotting only): empty module stubs.
Inputs are synthetic (generator below, seeded); outputs are whatever the reference wrote.
"""

import os
import sys
import types

import numpy as np
import scipy.ndimage  # noqa: F401  (bind conda's builds before touching sys.path)
import skimage  # noqa: F401
import sklearn.cluster  # noqa: F401
import tqdm  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------- stand-ins
class _Array:
    def __init__(self, data):
        self._a = data
        self.attrs = {}

    shape = property(lambda self: self._a.shape)
    dtype = property(lambda self: self._a.dtype)
    ndim = property(lambda self: self._a.ndim)

    def __getitem__(self, key):
        return np.array(self._a[key], copy=True)

    def __setitem__(self, key, value):
        self._a[key] = value


class _Group(dict):
    def create_dataset(self, name, shape, dtype, **_kw):
        self[name] = _Array(np.zeros(shape, dtype=dtype))
        return self[name]


_CONTAINERS = {}
zarr_stub = types.ModuleType("zarr")
zarr_stub.open = lambda path, mode="a", **_kw: _CONTAINERS.setdefault(str(path), _Group())
sys.modules["zarr"] = zarr_stub
for name in ("matplotlib", "matplotlib.pyplot", "torch", "torch.utils", "torch.utils.data", "gunpowder"):
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
if not hasattr(sys.modules["torch.utils.data"], "IterableDataset"):
    sys.modules["torch.utils.data"].IterableDataset = object     # base class of the unused ZarrDataset


class _T(np.ndarray):
    """utils/mean_shift.py only MOVES data through torch (from_numpy/arange/permute/view/
    contiguous/numpy); these are the same moves on the numpy buffer (shared memory, so the
    in-place coordinate add reaches the caller's array exactly as with torch.from_numpy)."""

    def permute(self, *dims):
        return self.transpose(*dims)

    def view(self, *shape):           # noqa: A003  (torch's view == reshape of a contiguous array)
        return self.reshape(*shape)

    def contiguous(self):
        return _as_t(np.ascontiguousarray(self))

    def numpy(self):
        return np.asarray(self)


def _as_t(a):
    return np.ndarray.view(a, _T)


if not hasattr(sys.modules["torch"], "from_numpy"):
    sys.modules["torch"].from_numpy = _as_t
    sys.modules["torch"].arange = np.arange
sys.path.insert(0, "/usr/local/lib/python3.10/dist-packages")
import attr  # noqa: E402,F401
import attrs  # noqa: E402,F401
sys.path.pop(0)

from cellulus.configs.inference_config import InferenceConfig  # noqa: E402
from cellulus.detect import detect  # noqa: E402
from cellulus.segment import segment  # noqa: E402


# ------------------------------------------------------------ synthetic data
def synthetic(shape, spacing, radius, seed, noise=0.25):
    """Jittered grid of discs/balls; embedding = (centre - pixel) + N(0, noise) in (x, y[, z])
    channel order, std small inside / large outside (+ ripple so Otsu sees a real histogram);
    raw = bright nucleus core inside each object on a dim cytoplasm, with a dark hole in
    some cores (binary_fill_holes has work to do)."""
    rng = np.random.RandomState(seed)
    nd = len(shape)
    grid = np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"))
    mean = rng.normal(0, noise, size=(nd,) + shape)
    std = 1.0 + 0.05 * rng.rand(*shape)
    raw = 0.05 * rng.rand(*shape)
    starts = [np.arange(spacing // 2, s - radius, spacing) for s in shape]
    k = 0
    for c in np.stack(np.meshgrid(*starts, indexing="ij"), -1).reshape(-1, nd):
        c = c + rng.randint(-3, 4, size=nd)
        d2 = sum((grid[i] - c[i]) ** 2 for i in range(nd))
        inside = d2 <= radius ** 2
        for i in range(nd):                       # channel 0 = x = last axis
            mean[i][inside] += (c[nd - 1 - i] - grid[nd - 1 - i])[inside]
        std[inside] = 0.01 + 0.02 * rng.rand(int(inside.sum()))
        raw[inside] = 0.3 + 0.05 * rng.rand(int(inside.sum()))
        core = d2 <= (0.6 * radius) ** 2
        raw[core] = 0.8 + 0.1 * rng.rand(int(core.sum()))
        if k % 2 == 0:
            raw[d2 <= 2] = 0.32
        k += 1
    emb = np.concatenate([mean, std[None]], 0)
    return np.round(emb * 4096) / 4096, raw       # dyadic values: exact in f64, small on disk


def run_case(out, tag, shape, spacing, radius, raw_dtype, use_seeds, num_bandwidths, bandwidth,
             min_size, rp, seed):
    nd = len(shape)
    path = f"{tag}.zarr"
    f = zarr_stub.open(path)
    embs, raws = [], []
    for s in range(2):
        e, r = synthetic(shape, spacing, radius, seed + s)
        embs.append(e)
        raws.append(r)
    raw = np.stack(raws)[:, None]
    if np.issubdtype(raw_dtype, np.integer):
        raw = np.round(raw * 200).astype(raw_dtype)
    else:
        raw = raw.astype(raw_dtype)
    f["raw"] = _Array(raw)
    f["raw"].attrs["axis_names"] = ["s", "c"] + ["z", "y", "x"][-nd:]
    f["embeddings"] = _Array(np.stack(embs))
    out[f"{tag}/raw"] = raw
    out[f"{tag}/embeddings"] = np.stack(embs)
    out[f"{tag}/params"] = np.array([bandwidth, min_size, rp, seed, num_bandwidths, int(use_seeds)], dtype=np.float64)
    base = dict(
        dataset_config=dict(container_path=path, dataset_name="raw"),
        detection_dataset_config=dict(container_path=path, dataset_name="detection",
                                      secondary_dataset_name="embeddings"),
        segmentation_dataset_config=dict(container_path=path, dataset_name="segmentation",
                                         secondary_dataset_name="detection"),
        use_seeds=use_seeds, num_bandwidths=num_bandwidths, bandwidth=bandwidth, min_size=min_size,
        reduction_probability=rp, device="cpu")
    np.random.seed(seed)
    try:
        detect(InferenceConfig(**base, post_processing="cell"))
    except ValueError as e:
        # use_seeds with num_bandwidths > 1: detect.py:142-144 re-reads the centred embeddings
        # AFTER mean_shift_segmentation added pixel coordinates to them in place, so the second
        # bandwidth clusters offsets + 2*coordinates against pixel-space seeds and sklearn
        # finds no point near any seed.  The error is the reference's behaviour; record it.
        out[f"{tag}/detect_error"] = np.array(f"{type(e).__name__}: {e}")
        out[f"{tag}/detection_partial"] = f["detection"][...]
        print(tag, "detect raised", type(e).__name__)
        return
    for name in ("detection", "binary-segmentation"):
        out[f"{tag}/{name}"] = f[name][...]
    # centred embeddings = embeddings - per-channel constant: a strided sample pins them
    out[f"{tag}/centered-embeddings_s4"] = f["centered-embeddings"][...][..., ::4, ::4]
    for pp in ("cell", "nucleus"):
        f.pop("segmentation", None)
        segment(InferenceConfig(**base, post_processing=pp, grow_distance=3, shrink_distance=6))
        out[f"{tag}/segmentation_{pp}"] = f["segmentation"][...]
    print(tag, "objects per sample/bandwidth:",
          [[int(out[f"{tag}/segmentation_cell"][s, b].max()) for b in range(num_bandwidths)] for s in range(2)],
          "nucleus:",
          [[int(out[f"{tag}/segmentation_nucleus"][s, b].max()) for b in range(num_bandwidths)] for s in range(2)])


def main():
    out = {}
    run_case(out, "2d_f32", (96, 112), 32, 10, np.float32, False, 2, 10.0, 25, 0.5, 11)
    run_case(out, "2d_u8_seeds", (80, 96), 32, 10, np.uint8, True, 1, 10.0, 25, 0.5, 21)
    run_case(out, "2d_seeds_bw2", (48, 64), 32, 10, np.float32, True, 2, 10.0, 25, 0.5, 51)
    run_case(out, "2d_f64", (64, 64), 32, 10, np.float64, False, 1, 10.0, 25, 1.0, 31)
    run_case(out, "3d_u16", (16, 36, 36), 18, 6, np.uint16, False, 1, 7.0, 60, 0.3, 41)
    np.savez_compressed(os.path.join(HERE, "g8_stages.npz"), **out)
    import sklearn
    print("written g8_stages.npz with scikit-image", skimage.__version__, "scikit-learn", sklearn.__version__)


if __name__ == "__main__":
    main()
