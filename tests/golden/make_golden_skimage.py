"""Golden vectors that need scikit-image (only installed under /opt/conda python3.9:
scikit-image 0.18.3, scipy 1.7.1):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 \
        /opt/conda/bin/python3.9 tests/golden/make_golden_skimage.py

Runs the REAL reference `cellulus.utils.misc.size_filter` (matplotlib, which that
module imports for plotting only, is stubbed), `skimage.measure.label`,
`skimage.filters.threshold_otsu` and `skimage.feature.peak_local_max` as
`cellulus/detect.py:88-132` and `cellulus/segment.py:103-108` call them.
Inputs: the label maps of g4_mean_shift.npz plus seeded random label images.
"""

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

for name in ("matplotlib", "matplotlib.pyplot"):
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)

from scipy.ndimage import gaussian_filter  # noqa: E402
from skimage import measure  # noqa: E402
from skimage.feature import peak_local_max  # noqa: E402
from skimage.filters import threshold_otsu  # noqa: E402

from cellulus.utils.misc import size_filter  # noqa: E402


def main():
    g4 = np.load(os.path.join(HERE, "g4_mean_shift.npz"))
    out = {}
    rng = np.random.RandomState(0)
    cases = {}
    for key in ("2d_rp1", "2d_rp02", "3d_rp05"):
        cases[key] = g4[f"{key}/labels"].astype(np.int32)
    # random multi-valued label images: touching regions of equal / different value, specks
    blocks = rng.randint(0, 4, size=(12, 14))
    cases["rand2d"] = np.kron(blocks, np.ones((5, 4), dtype=np.int32)).astype(np.int32)
    cases["rand2d"][rng.rand(*cases["rand2d"].shape) < 0.08] = 0
    cases["noise2d"] = (rng.rand(64, 48) < 0.45).astype(np.int32) * rng.randint(1, 3, size=(64, 48))
    cases["noise3d"] = (rng.rand(12, 20, 16) < 0.3).astype(np.int32) * rng.randint(1, 3, size=(12, 20, 16))
    cases["zeros2d"] = np.zeros((8, 9), dtype=np.int32)
    cases["full2d"] = np.ones((8, 9), dtype=np.int32) * 5
    for name, seg in cases.items():
        out[f"{name}/seg"] = seg
        out[f"{name}/label"] = measure.label(seg).astype(np.int32)
        for ms in (1, 4, 30):
            s = seg.copy()
            out[f"{name}/size_filter_{ms}"] = size_filter(s, ms).astype(np.int32)
            out[f"{name}/seg_after_{ms}"] = s
    # Otsu on std-like images (float64) and on the synthetic std channels
    for i, key in enumerate(("2d_rp1", "3d_rp05")):
        img = g4[f"{key}/std"] + rng.normal(0, 0.05, size=g4[f"{key}/std"].shape)
        out[f"otsu{i}/image"] = img
        out[f"otsu{i}/threshold"] = np.array(threshold_otsu(img))
    img = rng.gamma(2.0, 1.0, size=(70, 50))
    out["otsu2/image"] = img
    out["otsu2/threshold"] = np.array(threshold_otsu(img))
    # peak_local_max on a smoothed negative magnitude map (detect.py:128-132)
    mag = gaussian_filter(rng.rand(60, 70), sigma=2)
    out["peaks/image"] = -mag
    out["peaks/coords"] = peak_local_max(-mag)
    mag3 = gaussian_filter(rng.rand(16, 20, 24), sigma=2)
    out["peaks3/image"] = -mag3
    out["peaks3/coords"] = peak_local_max(-mag3)
    np.savez_compressed(os.path.join(HERE, "g5_skimage.npz"), **out)
    import skimage
    print("written g5_skimage.npz with scikit-image", skimage.__version__)


if __name__ == "__main__":
    main()
