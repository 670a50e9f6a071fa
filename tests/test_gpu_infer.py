"""GPU parity of the inference-side hot path: mean-shift, connected components +
size filter, Otsu, distance-transform grow/shrink, and the whole
predict -> detect -> segment pipeline over zarr — against the CPU oracle and the
golden vectors captured from the real reference (tests/golden)."""

import os

import numpy as np
import pytest
import torch

from cellulus_amd.utils import mean_shift as MS
from cellulus_amd.utils.misc import label_on_device, size_filter
from cellulus_amd.utils.otsu import histogram_on_device, threshold_otsu
from oracle import infer_oracle as IO

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------------ mean shift
MS_CASES = ["2d_rp1", "2d_rp05", "2d_rp02", "2d_seeds", "3d_rp05", "2d_empty"]


@pytest.mark.parametrize("case", MS_CASES)
def test_mean_shift_matches_reference_golden(case, device):
    g = np.load(os.path.join(G, "g4_mean_shift.npz"))
    bw, rp, thr, seed = g[f"{case}/params"]
    mean = g[f"{case}/mean"].copy()
    seeds = g[f"{case}/seeds"] if f"{case}/seeds" in g.files else None
    np.random.seed(int(seed))
    labels = MS.mean_shift_segmentation(mean, g[f"{case}/std"], float(bw), 10, float(rp), float(thr), seeds,
                                        device=device)
    ref = g[f"{case}/labels"]
    assert labels.dtype == np.int32 and labels.shape == ref.shape
    np.testing.assert_array_equal(mean, g[f"{case}/mean_after"])     # input mutated like the reference
    np.testing.assert_array_equal(labels > 0, ref > 0)
    # the instance PARTITION is bit-exact (ids are canonicalised by the final CC labelling)
    np.testing.assert_array_equal(IO.label(labels), IO.label(ref))
    np.testing.assert_array_equal(labels, ref)


@pytest.mark.parametrize("shape,rp", [((160, 192), 0.3), ((20, 48, 40), 0.4)])
def test_mean_shift_matches_oracle_on_fresh_inputs(shape, rp, device):
    mean, std = IO.synthetic_embeddings(shape, spacing=40 if len(shape) == 2 else 20,
                                        radius=11 if len(shape) == 2 else 6, seed=5)
    bw = 13.0 if len(shape) == 2 else 7.0
    np.random.seed(3)
    ref = IO.mean_shift_segmentation(mean.copy(), std, bw, 10, rp, 0.5, None)
    np.random.seed(3)
    got = MS.mean_shift_segmentation(mean.copy(), std, bw, 10, rp, 0.5, None, device=device)
    np.testing.assert_array_equal(IO.label(got), IO.label(ref))
    assert got.max() == ref.max() >= 4


def test_mean_shift_kernels_match_oracle_numbers(device):
    """centres/counts of the iteration kernel vs the C restatement, point by point."""
    from cellulus_amd import _clx

    rng = np.random.default_rng(0)
    fit = np.concatenate([rng.normal(c, 2.0, size=(300, 2)) for c in ((10, 10), (60, 20), (30, 70))])
    seeds = fit[::7].copy()
    lib = IO._clib()
    ns = len(seeds)
    c_ref = np.empty((ns, 2)); n_ref = np.empty(ns, np.int32); i_ref = np.empty(ns, np.int32)
    lib.ms_oracle_iterate(IO._dptr(fit), len(fit), IO._dptr(seeds), ns, 2, 6.0, 300, IO._dptr(c_ref),
                          IO._iptr(n_ref), IO._iptr(i_ref))
    fit_d = torch.from_numpy(fit).to(device); seeds_d = torch.from_numpy(seeds).to(device)
    c = torch.empty((ns, 2), dtype=torch.float64, device=device)
    n = torch.empty(ns, dtype=torch.int32, device=device)
    it = torch.empty(ns, dtype=torch.int32, device=device)
    _clx.call("clx_ms_iterate", _clx.ptr(fit_d), len(fit), _clx.ptr(seeds_d), ns, 2, 6.0, 300,
              _clx.ptr(c), _clx.ptr(n), _clx.ptr(it), _clx.stream_ptr(device))
    np.testing.assert_array_equal(n.cpu().numpy(), n_ref)
    np.testing.assert_array_equal(it.cpu().numpy(), i_ref)
    np.testing.assert_allclose(c.cpu().numpy(), c_ref, rtol=0, atol=1e-11)   # summation order only


def test_mean_shift_degenerate_inputs(device):
    mean, std = IO.synthetic_embeddings((40, 40), spacing=40, radius=8, seed=1)
    # nothing below the threshold -> all background, no RNG draw, no error (mean_shift.py:83-84)
    out = MS.mean_shift_segmentation(mean.copy(), np.ones_like(std), 10.0, 10, 0.5, 0.5, None, device=device)
    assert out.shape == (40, 40) and out.max() == 0
    # reduction keeps no point -> sklearn raises on an empty fit set
    with pytest.raises(ValueError):
        MS.mean_shift_segmentation(mean.copy(), std, 10.0, 10, 0.0, 0.5, None, device=device)
    # seeds far away from every point -> "No point was within bandwidth"
    with pytest.raises(ValueError, match="bandwidth"):
        MS.mean_shift_segmentation(mean.copy(), std, 3.0, 10, 1.0, 0.5, np.array([[1000, 1000]]), device=device)
    with pytest.raises(TypeError):
        MS.mean_shift_segmentation(mean.astype(np.float32), std, 3.0, 10, 1.0, 0.5, None, device=device)


# ------------------------------------------------------- CC / size filter / Otsu
SK_CASES = ["2d_rp1", "2d_rp02", "3d_rp05", "rand2d", "noise2d", "noise3d", "zeros2d", "full2d"]


@pytest.mark.parametrize("case", SK_CASES)
def test_label_and_size_filter_bit_exact_vs_skimage_golden(case, device):
    g = np.load(os.path.join(G, "g5_skimage.npz"))
    seg = g[f"{case}/seg"].astype(np.int32)
    lab, ncomp = label_on_device(torch.from_numpy(seg).to(device), 1)
    np.testing.assert_array_equal(lab.cpu().numpy(), g[f"{case}/label"])
    assert int(ncomp.item()) == g[f"{case}/label"].max()
    for ms in (1, 4, 30):
        s = seg.copy()
        out = size_filter(s, ms, device=device)
        assert out.dtype == np.int64
        np.testing.assert_array_equal(out, g[f"{case}/size_filter_{ms}"])
        np.testing.assert_array_equal(s, g[f"{case}/seg_after_{ms}"])      # input zeroed like the reference
    assert size_filter(seg, 0, device=device) is seg                          # misc.py:12-13


@pytest.mark.parametrize("shape", [(512, 512), (37, 61), (48, 64, 56), (1, 1), (1, 300), (3000, 2), (200, 260)])
def test_size_filter_matches_oracle_random(shape, device):
    rng = np.random.default_rng(sum(shape))
    seg = (rng.random(shape) < 0.55).astype(np.int32) * rng.integers(1, 4, size=shape).astype(np.int32)
    for ms in (1, 9, 70):
        np.testing.assert_array_equal(size_filter(seg.copy(), ms, device=device), IO.size_filter(seg.copy(), ms))
    # idempotence: filtering a filtered map changes nothing but (possibly) nothing at all
    once = size_filter(seg.copy(), 9, device=device)
    np.testing.assert_array_equal(size_filter(once.copy(), 9, device=device), once)


@pytest.mark.parametrize("shape", [(70, 130), (130, 70), (5, 129, 67), (64, 64), (3, 3, 200)])
def test_size_filter_run_based_union_find_structured_images(shape, device):
    """the run-based union-find (one link per pair of touching runs) on images made of runs: blocks,
    diagonal and anti-diagonal one-pixel lines (8-/26-connected only through corners), runs that
    cross the 64-pixel segments a wavefront owns, equal labels on different objects"""
    rng = np.random.default_rng(3)
    idx = np.indices(shape)
    images = [np.kron(rng.integers(0, 3, size=tuple(-(-n // 5) for n in shape)),
                      np.ones((5,) * len(shape), int))[tuple(slice(0, n) for n in shape)]]
    images.append(((idx.sum(0) % 3) == 0) * 1)                       # anti-diagonal lines
    images.append(((idx[-1] - idx[-2]) % 4 == 0) * 2)                 # diagonal lines
    images.append(((idx[-2] % 2 == 0) * 5))                           # full-width runs, every other row
    images.append(((idx[-1] + 2 * idx[-2] + (idx[0] if len(shape) == 3 else 0)) % 5 < 2) * (1 + idx[-1] // 50))
    images.append(np.ones(shape, int))
    for seg in images:
        seg = seg.astype(np.int32)
        for ms in (1, 6):
            np.testing.assert_array_equal(size_filter(seg.copy(), ms, device=device), IO.size_filter(seg.copy(), ms))


def test_long_snake_component(device):
    """A single serpentine component: worst case for union-find chains."""
    seg = np.zeros((64, 65), dtype=np.int32)
    seg[::2, :] = 7
    for r in range(1, 63, 2):
        seg[r, 64 if (r // 2) % 2 == 0 else 0] = 7
    out = size_filter(seg.copy(), 5, device=device)
    np.testing.assert_array_equal(out, IO.size_filter(seg.copy(), 5))
    assert out.max() == 1


def test_otsu_bit_exact_vs_skimage_golden(device):
    g = np.load(os.path.join(G, "g5_skimage.npz"))
    for i in range(3):
        img = g[f"otsu{i}/image"]
        got = threshold_otsu(torch.from_numpy(img).to(device))
        assert got == float(g[f"otsu{i}/threshold"])
        counts, edges = histogram_on_device(torch.from_numpy(img).to(device))
        ref_counts, ref_edges = np.histogram(img.ravel(), bins=256)
        np.testing.assert_array_equal(counts, ref_counts)
        np.testing.assert_array_equal(edges, ref_edges)
    assert threshold_otsu(torch.full((5, 5), 2.5, dtype=torch.float64, device=device)) == 2.5


# ------------------------------------------------------------------ EDT
@pytest.mark.parametrize("shape", [(64, 80), (300, 17), (20, 24, 28)])
def test_edt_sq_bit_exact(shape, device):
    from cellulus_amd import _clx

    rng = np.random.default_rng(1)
    for density in (0.02, 0.5, 1.0):      # 1.0: no zero anywhere -> scipy's phantom zero
        m = (rng.random(shape) < density) if density < 1.0 else np.ones(shape, bool)
        ref = IO.edt_sq(m)
        Z, Y, X = (1,) * (3 - len(shape)) + tuple(shape)
        md = torch.from_numpy(m.astype(np.uint8)).to(device)
        out = torch.empty(shape, dtype=torch.int32, device=device)
        ws = torch.empty(m.size * 4 + 64, dtype=torch.uint8, device=device)
        _clx.call("clx_edt_sq", _clx.ptr(md), _clx.ptr(out), Z, Y, X, 0, _clx.ptr(ws), _clx.stream_ptr(device))
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
        # capped search: exact below cap^2, >= cap^2 elsewhere
        cap = 4
        _clx.call("clx_edt_sq", _clx.ptr(md), _clx.ptr(out), Z, Y, X, cap, _clx.ptr(ws), _clx.stream_ptr(device))
        got = out.cpu().numpy()
        if density < 1.0:
            np.testing.assert_array_equal(got[ref < cap * cap], ref[ref < cap * cap])
            assert (got[ref >= cap * cap] >= cap * cap).all()
        else:
            np.testing.assert_array_equal(got, ref)     # no zero at all: phantom distances


@pytest.mark.parametrize("shape", [(96, 96), (24, 40, 40)])
def test_grow_shrink_bit_exact(shape, device):
    from cellulus_amd.segment import grow_shrink_on_device

    key = "2d_rp1" if len(shape) == 2 else "3d_rp05"
    seg = np.load(os.path.join(G, "g4_mean_shift.npz"))[f"{key}/labels"]
    for grow, shrink in ((3, 6), (1, 1), (0, 2), (5, 2)):
        ref = IO.grow_shrink(seg, grow, shrink)
        got = grow_shrink_on_device(torch.from_numpy(seg.copy()).to(device), grow, shrink).cpu().numpy()
        np.testing.assert_array_equal(got, ref)
    # everything foreground after growing: scipy's phantom-zero artefact is reproduced too
    full = np.ones(shape, dtype=np.int32)
    np.testing.assert_array_equal(
        grow_shrink_on_device(torch.from_numpy(full.copy()).to(device), 3, 6).cpu().numpy(),
        IO.grow_shrink(full, 3, 6))


@pytest.mark.parametrize("shape", [(97, 131), (33, 64), (1, 5), (70, 300), (9, 21, 45), (24, 8, 33), (3, 70, 70)])
def test_grow_shrink_tile_kernel_random_images(shape, device):
    """the one-kernel grow/shrink (tile + halo in LDS) against the scipy-EDT oracle: ragged extents,
    sparse / dense / empty / full label images, bounds from 0 to the fall-back of the generic path"""
    from cellulus_amd.segment import grow_shrink_on_device

    rng = np.random.default_rng(7)
    images = [np.zeros(shape, np.int32), np.ones(shape, np.int32)]
    for density in (0.003, 0.05, 0.6):
        images.append(((rng.random(shape) < density) * rng.integers(1, 9, size=shape)).astype(np.int32))
    blocks = np.kron(rng.integers(0, 3, size=tuple(-(-s // 7) for s in shape)), np.ones((7,) * len(shape), int))
    images.append(blocks[tuple(slice(0, s) for s in shape)].astype(np.int32))
    for seg in images:
        for grow, shrink in ((3, 6), (1, 1), (0, 0), (0, 3), (2, 0), (6, 3), (-1, 4), (9, 9), (14, 14)):
            ref = IO.grow_shrink(seg, grow, shrink)
            got = grow_shrink_on_device(torch.from_numpy(seg.copy()).to(device), grow, shrink).cpu().numpy()
            np.testing.assert_array_equal(got, ref, err_msg=f"shape {shape} grow {grow} shrink {shrink}")


# -------------------------------------------------------- end-to-end pipeline
def _write_raw(path, nd, seed=0, shape=None):
    from cellulus_amd.utils import zarr_io

    rng = np.random.default_rng(seed)
    shape = shape or ((2, 1, 72, 80) if nd == 2 else (1, 1, 40, 44, 40))
    raw = rng.random(shape).astype(np.float32)
    f = zarr_io.open(path)
    f["test/raw"] = raw
    f["test/raw"].attrs["axis_names"] = ["s", "c", "y", "x"] if nd == 2 else ["s", "c", "z", "y", "x"]
    return raw


@pytest.mark.parametrize("nd", [2, 3])
def test_infer_pipeline_matches_oracle(nd, device, tmp_path, monkeypatch):
    """infer(): predict -> detect -> segment over zarr vs the oracle stage by stage."""
    from cellulus_amd.configs import ExperimentConfig
    from cellulus_amd.infer import infer
    from cellulus_amd.utils import zarr_io
    from oracle.unet_oracle import OracleUNetModel

    monkeypatch.chdir(tmp_path)
    container = str(tmp_path / "data.zarr")
    raw = _write_raw(container, nd)
    mcfg = dict(num_fmaps=8, fmap_inc_factor=2, features_in_last_layer=16,
                downsampling_factors=[[2] * nd])
    torch.manual_seed(0)
    oracle = OracleUNetModel(in_channels=1, out_channels=nd, num_spatial_dims=nd, **mcfg)
    # Kaiming weights (train.py:65-68): activations keep their scale through the ten layers, so the
    # embeddings DEPEND on which pixels the salt/pepper noise hits — with torch's default
    # initialisation the noise moves them by 1e-5 and the comparison below would not notice a
    # different random sequence
    for layer in oracle.modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    os.makedirs("models", exist_ok=True)
    torch.save({"model_state_dict": oracle.state_dict()}, "models/best_loss.pth")
    crop = [56, 56] if nd == 2 else [36, 36, 36]
    n_it = 2
    cfg = ExperimentConfig(
        model_config=dict(checkpoint="models/best_loss.pth", **mcfg),
        object_size=12, normalization_factor=1.0,
        inference_config=dict(
            dataset_config=dict(container_path=container, dataset_name="test/raw"),
            prediction_dataset_config=dict(container_path=container, dataset_name="embeddings"),
            detection_dataset_config=dict(container_path=container, dataset_name="detection",
                                          secondary_dataset_name="embeddings"),
            segmentation_dataset_config=dict(container_path=container, dataset_name="segmentation",
                                             secondary_dataset_name="detection"),
            crop_size=crop, num_infer_iterations=n_it, p_salt_pepper=0.05,
            reduction_probability=0.5, min_size=6, grow_distance=2, shrink_distance=3,
            device="cuda:0"))
    torch.manual_seed(42)
    np.random.seed(42)
    infer(cfg)
    f = zarr_io.open(container, "r")
    spatial = raw.shape[2:]
    emb = f["embeddings"][...]
    assert emb.dtype == np.float64 and emb.shape == (raw.shape[0], nd + 1) + spatial
    assert f["embeddings"].attrs["axis_names"] == ["s", "c"] + ["z", "y", "x"][-nd:]

    # ---- oracle predict (cellulus/predict.py:21-135 restated, INCLUDING the dry-run forward on a
    # zero tile that the reference runs in infer mode before the scan): same torch.rand sequence.
    # infer() builds its model first (infer.py:41-50: the default initialisation of every layer
    # draws from the same generator before the checkpoint overwrites it) — so does this replay.
    after_infer = torch.rand(5)
    from oracle.unet_oracle import predict_scan
    torch.manual_seed(42)
    OracleUNetModel(in_channels=1, out_channels=nd, num_spatial_dims=nd, **mcfg)
    ref_emb = predict_scan(oracle, raw, crop, 0.05, n_it, 1.0, literal_dry_run=True)
    err = np.abs(emb - ref_emb).max()
    assert err < 1e-4, err
    # ... and the generator is where the reference leaves it
    assert torch.equal(torch.rand(5), after_infer)
    # the comparison is sensitive to the sequence: the same scan WITHOUT the dry run's draws (the
    # round-2 behaviour) is far outside the bar
    torch.manual_seed(42)
    OracleUNetModel(in_channels=1, out_channels=nd, num_spatial_dims=nd, **mcfg)
    oracle.set_infer(0.05, n_it)
    padded = np.pad(raw[0], [(0, 0)] + [(8, 8)] * nd, mode="reflect")
    with torch.no_grad():
        first = oracle(torch.from_numpy(padded[(slice(None),) + tuple(slice(0, c) for c in crop)].copy())[None])[0].numpy()
    out_sl = (0, slice(None)) + tuple(slice(0, c - 16) for c in crop)
    assert np.abs(first - ref_emb[out_sl]).max() > 100 * max(err, 1e-6)

    # ---- detect + segment, stage by stage on the pipeline's own zarr data
    np.random.seed(42)
    det = f["detection"][...]
    seg = f["segmentation"][...]
    assert det.dtype == seg.dtype == np.uint16
    bw, min_size = 0.5 * 12, 6
    for s in range(raw.shape[0]):
        thr = IO.threshold_otsu(emb[s, -1])
        np.testing.assert_array_equal(f["binary-segmentation"][s, 0], (emb[s, -1] < thr).astype(np.uint16))
        ref_det = IO.mean_shift_segmentation(emb[s][np.newaxis, :nd].copy(), emb[s, -1], bw, min_size, 0.5, thr, None)
        np.testing.assert_array_equal(IO.label(det[s, 0]), IO.label(ref_det))
        ref_seg = IO.size_filter(IO.grow_shrink(det[s, 0].astype(np.int32), 2, 3), min_size)
        np.testing.assert_array_equal(seg[s, 0], ref_seg.astype(np.uint16))
        centred = f["centered-embeddings"][s]
        m = (emb[s, -1] < thr)[None] * emb[s, :nd]
        for k in range(nd):
            assert np.allclose(centred[k], emb[s, k] - m[k][m[k] != 0].mean(), atol=0, rtol=0)


@pytest.mark.parametrize("nd", [2, 3])
def test_mean_shift_grid_kernel_equals_bruteforce(nd, device):
    """The cell-bucketed iteration visits exactly the members the brute-force sweep finds."""
    import ctypes

    from cellulus_amd import _clx

    shape = (200, 240) if nd == 2 else (24, 64, 72)
    mean, std = IO.synthetic_embeddings(shape, spacing=40 if nd == 2 else 24, radius=12 if nd == 2 else 8, seed=9)
    m = mean.copy()
    for c in range(nd):
        ax = nd - 1 - c
        sh = [1] * nd
        sh[ax] = shape[ax]
        m[0, c] += np.arange(shape[ax]).reshape(sh)
    pts = np.ascontiguousarray(np.moveaxis(m[0], 0, -1)[std < 0.5].reshape(-1, nd))
    assert len(pts) > MS.GRID_MIN_POINTS
    bw = 13.0 if nd == 2 else 9.0
    fit = torch.from_numpy(pts).to(device)
    ns = len(pts)
    st = _clx.stream_ptr(device)

    def outputs():
        return (torch.empty((ns, nd), dtype=torch.float64, device=device),
                torch.empty(ns, dtype=torch.int32, device=device),
                torch.empty(ns, dtype=torch.int32, device=device))

    c0, n0, i0 = outputs()
    _clx.call("clx_ms_iterate", _clx.ptr(fit), ns, _clx.ptr(fit), ns, nd, bw, 300,
              _clx.ptr(c0), _clx.ptr(n0), _clx.ptr(i0), st)
    fs, cell_start, origin, cell, (nx, ny, nz) = MS._bucket(fit, bw)
    c1, n1, i1 = outputs()
    _clx.call("clx_ms_iterate_grid", _clx.ptr(fs), ns, _clx.ptr(cell_start),
              (ctypes.c_double * nd)(*origin.tolist()), cell, nx, ny, nz, _clx.ptr(fit), ns, nd, bw, 300,
              _clx.ptr(c1), _clx.ptr(n1), _clx.ptr(i1), st)
    np.testing.assert_array_equal(n1.cpu().numpy(), n0.cpu().numpy())
    np.testing.assert_array_equal(i1.cpu().numpy(), i0.cpu().numpy())
    np.testing.assert_allclose(c1.cpu().numpy(), c0.cpu().numpy(), rtol=0, atol=1e-10)
    # a seed far outside the grid finds nothing, like the brute-force kernel
    far = torch.full((1, nd), 1e6, dtype=torch.float64, device=device)
    c2, n2, i2 = (torch.empty((1, nd), dtype=torch.float64, device=device),
                  torch.empty(1, dtype=torch.int32, device=device), torch.empty(1, dtype=torch.int32, device=device))
    _clx.call("clx_ms_iterate_grid", _clx.ptr(fs), ns, _clx.ptr(cell_start),
              (ctypes.c_double * nd)(*origin.tolist()), cell, nx, ny, nz, _clx.ptr(far), 1, nd, bw, 300,
              _clx.ptr(c2), _clx.ptr(n2), _clx.ptr(i2), st)
    assert int(n2.item()) == 0 and int(i2.item()) == 0


@pytest.mark.parametrize("nd", [2, 3])
def test_bucket_kernel_is_a_stable_sort_by_cell(nd, device):
    """clx_ms_bucket == stable sort of the points by uniform-grid cell id (x fastest) + exclusive
    scan of the cell histogram, for clustered points (thousands per cell) and scattered ones."""
    from cellulus_amd.utils import mean_shift as MS

    rng = np.random.default_rng(nd)
    clustered = np.concatenate([rng.normal(loc=c, scale=2.0, size=(3000, nd)) for c in rng.uniform(0, 300, size=(12, nd))])
    scattered = rng.uniform(-50, 400, size=(5000, nd))
    pts = np.concatenate([clustered, scattered, clustered[:7]])[rng.permutation(41007)]
    bw = 11.0
    fs, cell_start, origin, cell, (nx, ny, nz) = MS._bucket(torch.from_numpy(pts).to(device), bw)
    coords = np.floor((pts - pts.min(axis=0)) / cell).astype(np.int64)
    assert (nx, ny) == (coords[:, 0].max() + 1, coords[:, 1].max() + 1) and nz == (coords[:, 2].max() + 1 if nd == 3 else 1)
    cid = coords[:, 0] + nx * coords[:, 1] + (nx * ny * coords[:, 2] if nd == 3 else 0)
    order = np.argsort(cid, kind="stable")
    np.testing.assert_array_equal(fs.cpu().numpy(), pts[order])
    ref_start = np.concatenate([[0], np.cumsum(np.bincount(cid, minlength=nx * ny * nz))])
    np.testing.assert_array_equal(cell_start.cpu().numpy(), ref_start)
    np.testing.assert_array_equal(origin, pts.min(axis=0))
    # a degenerate prediction: (almost) every embedding lands in ONE cell — the cell is ranked by one thread per
    # point instead of one wavefront (round-2 advisor finding), the order is still the stable one
    heap = np.concatenate([rng.uniform(100.0, 100.0 + 0.9 * bw, size=(30000, nd)), rng.uniform(0, 300, size=(50, nd))])
    heap = heap[rng.permutation(len(heap))]
    fs, cell_start, _origin, cell, (nx, ny, nz) = MS._bucket(torch.from_numpy(heap).to(device), bw)
    coords = np.floor((heap - heap.min(axis=0)) / cell).astype(np.int64)
    cid = coords[:, 0] + nx * coords[:, 1] + (nx * ny * coords[:, 2] if nd == 3 else 0)
    assert np.bincount(cid).max() > 5000              # far beyond what one wavefront ranks
    np.testing.assert_array_equal(fs.cpu().numpy(), heap[np.argsort(cid, kind="stable")])


@pytest.mark.parametrize("nd", [2, 3])
def test_grid_assignment_equals_the_plain_nearest_centre_loop(nd, device):
    """clx_ms_assign_grid == clx_ms_assign_cells == clx_ms_assign (first minimum over all centres): pixels near their centre,
    pixels far from every centre (fall back to the full loop), exact ties between two centres
    (the smaller index wins), centres sharing a cell."""
    import ctypes

    from cellulus_amd import _clx
    from cellulus_amd.utils import mean_shift as MS

    rng = np.random.default_rng(10 + nd)
    K, bw = 150, 9.0
    centers = rng.uniform(0, 400, size=(K, nd))
    centers[5] = centers[4] + 0.5                      # two centres in one cell
    centers[7] = centers[6]
    centers[7, 0] += 4.0                                # tie partners: (6, 7) around their midpoint
    near = centers[rng.integers(0, K, size=20000)] + rng.normal(0, 2.0, size=(20000, nd))
    far = rng.uniform(-200, 800, size=(3000, nd))
    mid = np.repeat((0.5 * (centers[6] + centers[7]))[None], 16, axis=0)
    mid[:, 1] += np.arange(16) * 0.25                    # equidistant to centres 6 and 7, exactly
    pts = np.concatenate([near, far, mid])
    n = len(pts)
    X = torch.from_numpy(pts).to(device)
    index = torch.arange(n, dtype=torch.int32, device=device)
    cc = torch.from_numpy(centers).to(device)
    st = _clx.stream_ptr(device)
    ref = torch.zeros(n, dtype=torch.int32, device=device)
    _clx.call("clx_ms_assign", _clx.ptr(X), _clx.ptr(index), n, _clx.ptr(cc), K, nd, _clx.ptr(ref), st)
    order, cstart, origin, (gx, gy, gz) = MS._center_grid(centers, bw)
    got = torch.zeros(n, dtype=torch.int32, device=device)
    order_d, cstart_d = torch.from_numpy(order).to(device), torch.from_numpy(cstart).to(device)
    _clx.call("clx_ms_assign_grid", _clx.ptr(X), _clx.ptr(index), n, _clx.ptr(cc), K, nd,
              _clx.ptr(order_d), _clx.ptr(cstart_d),
              (ctypes.c_double * nd)(*origin.tolist()), bw, gx, gy, gz, _clx.ptr(got), st)
    np.testing.assert_array_equal(got.cpu().numpy(), ref.cpu().numpy())
    # ... and the form the product calls: centres handed over in cell order (clx_ms_assign_cells)
    got2 = torch.zeros(n, dtype=torch.int32, device=device)
    cc_sorted = torch.from_numpy(np.ascontiguousarray(centers[order])).to(device)
    _clx.call("clx_ms_assign_cells", _clx.ptr(X), _clx.ptr(index), n, _clx.ptr(cc_sorted), K, nd,
              _clx.ptr(order_d), _clx.ptr(cstart_d),
              (ctypes.c_double * nd)(*origin.tolist()), bw, gx, gy, gz, _clx.ptr(got2), st)
    np.testing.assert_array_equal(got2.cpu().numpy(), ref.cpu().numpy())
    d2 = ((pts[:, None, :] - centers[None]) ** 2).sum(-1)
    np.testing.assert_array_equal(ref.cpu().numpy()[:-16], d2.argmin(1)[:-16] + 1)
    assert set(ref.cpu().numpy()[-16:].tolist()) <= {7, 8}


@pytest.mark.parametrize("shape", [(64, 80), (97, 33), (5, 300), (20, 24, 28), (3, 40, 9)])
def test_seeds_on_device_equal_scipy_and_skimage_semantics(shape, device):
    """use_seeds = true (detect.py:128-132): np.linalg.norm -> scipy gaussian_filter(sigma=2) ->
    peak_local_max of the negated map, on the device in float64 with the libraries' operation order:
    the smoothed map bit for bit, the seed list element for element (order included)."""
    from scipy.ndimage import gaussian_filter

    from cellulus_amd import _clx
    from cellulus_amd.detect import gaussian_weights, peak_local_max, seeds_on_device

    nd = len(shape)
    rng = np.random.default_rng(sum(shape))
    emb = rng.normal(size=(nd,) + shape) * 5.0
    emb[(slice(None),) + tuple(s // 2 for s in shape)] = 0.0        # an exact zero of the magnitude
    mag = np.linalg.norm(emb, axis=0)
    smooth = gaussian_filter(mag, sigma=2)
    ref = np.flip(peak_local_max(-smooth), 1)
    emb_d = torch.from_numpy(emb).to(device)
    got = seeds_on_device(emb_d, nd)
    np.testing.assert_array_equal(got, ref)
    # the intermediate maps, bit for bit
    Z, Y, X = (1,) * (3 - nd) + shape
    npix = mag.size
    st = _clx.stream_ptr(device)
    m_d = torch.empty(npix, dtype=torch.float64, device=device)
    s_d, t_d = torch.empty_like(m_d), torch.empty_like(m_d)
    w, radius = gaussian_weights(2.0)
    assert radius == 8
    w_d = torch.from_numpy(w).to(device)
    _clx.call("clx_offset_magnitude", _clx.ptr(emb_d), _clx.ptr(m_d), nd, npix, st)
    _clx.call("clx_gaussian_filter_f64", _clx.ptr(m_d), _clx.ptr(s_d), _clx.ptr(t_d), Z, Y, X, _clx.ptr(w_d), radius, st)
    np.testing.assert_array_equal(m_d.cpu().numpy().reshape(shape), mag)
    np.testing.assert_array_equal(s_d.cpu().numpy().reshape(shape), smooth)
    # a constant map has no peak (nothing exceeds the minimum); plateaus keep every pixel of the plateau
    flat = seeds_on_device(torch.ones((nd,) + shape, dtype=torch.float64, device=device), nd)
    assert flat.shape == (0, nd)


# ------------------------------------------------------------------ greedy clustering
@pytest.mark.parametrize("case", ["2d", "3d"])
def test_greedy_cluster_matches_reference_golden(case, device):
    from cellulus_amd.utils.greedy_cluster import Cluster2d, Cluster3d

    g = np.load(os.path.join(G, "g6_greedy.npz"))
    pred, fg = g[f"{case}/pred"], g[f"{case}/fg"]
    bw, ms = g[f"{case}/params"]
    if case == "2d":
        c = Cluster2d(width=pred.shape[2], height=pred.shape[1], fg_mask=fg, device=device)
    else:
        c = Cluster3d(width=pred.shape[3], height=pred.shape[2], depth=pred.shape[1], fg_mask=fg, device=device)
    seg = c.cluster(prediction=pred, bandwidth=float(bw), min_object_size=int(ms))
    assert seg.dtype == torch.int16 and not seg.is_cuda
    np.testing.assert_array_equal(seg.numpy(), g[f"{case}/seg"])


def test_greedy_cluster_matches_oracle_fresh_and_degenerate(device):
    from cellulus_amd.utils.greedy_cluster import Cluster2d

    mean, std = IO.synthetic_embeddings((160, 144), spacing=40, radius=11, seed=12)
    rng = np.random.RandomState(3)
    std = std + rng.uniform(0, 0.05, size=std.shape)
    pred = np.concatenate([mean[0], std[None]], 0)
    fg = std < 0.5
    for bw, ms in ((7.0, 30), (3.0, 5), (20.0, 30)):
        ref = IO.greedy_cluster(pred, fg, bw, ms)
        got = Cluster2d(width=144, height=160, fg_mask=fg, device=device).cluster(pred, bw, ms).numpy()
        np.testing.assert_array_equal(got, ref)
    # empty foreground -> all zeros, no launch
    got = Cluster2d(width=144, height=160, fg_mask=np.zeros_like(fg), device=device).cluster(pred, 7.0, 30)
    assert got.shape == (160, 144) and int(got.max()) == 0
    # seed threshold above every score -> nothing clustered
    got = Cluster2d(width=144, height=160, fg_mask=fg, device=device).cluster(pred, 7.0, 30, seed_thresh=2.0)
    assert int(got.max()) == 0


STAGE_CASES = ["2d_f32", "2d_u8_seeds", "2d_f64", "3d_u16"]


def _stage_container(tmp_path, g, case):
    from cellulus_amd.utils import zarr_io

    container = str(tmp_path / f"{case}.zarr")
    f = zarr_io.open(container)
    raw = g[f"{case}/raw"]
    f["raw"] = raw
    f["raw"].attrs["axis_names"] = ["s", "c"] + ["z", "y", "x"][-(raw.ndim - 2):]
    f["embeddings"] = g[f"{case}/embeddings"]
    return container


def _stage_config(container, g, case, **kw):
    from cellulus_amd.configs import InferenceConfig

    bw, ms, rp, _seed, nb, use_seeds = g[f"{case}/params"]
    return InferenceConfig(
        dataset_config=dict(container_path=container, dataset_name="raw"),
        detection_dataset_config=dict(container_path=container, dataset_name="detection",
                                      secondary_dataset_name="embeddings"),
        segmentation_dataset_config=dict(container_path=container, dataset_name="segmentation",
                                         secondary_dataset_name=kw.pop("segment_from", "detection")),
        use_seeds=bool(use_seeds), num_bandwidths=int(nb), bandwidth=float(bw), min_size=int(ms),
        reduction_probability=float(rp), grow_distance=3, shrink_distance=6, device="cuda:0", **kw)


@pytest.mark.parametrize("case", STAGE_CASES)
def test_detect_and_segment_stages_match_real_reference(device, tmp_path, monkeypatch, case):
    """g8: outputs of the REAL cellulus detect() / segment() (tests/golden/make_golden_stages.py)
    vs the HIP stages on the same zarr inputs: Otsu mask, centred embeddings, mean-shift
    detection (2-D/3-D, with seeds, two bandwidths), then cell and nucleus post-processing
    (raw f32 / f64 / u8 / u16) + size filter.  Integer outputs bit-exact."""
    from cellulus_amd.detect import detect
    from cellulus_amd.segment import segment
    from cellulus_amd.utils import zarr_io

    g = np.load(os.path.join(G, "g8_stages.npz"))
    monkeypatch.chdir(tmp_path)
    container = _stage_container(tmp_path, g, case)
    np.random.seed(int(g[f"{case}/params"][3]))
    detect(_stage_config(container, g, case))
    f = zarr_io.open(container, "r")
    np.testing.assert_array_equal(f["binary-segmentation"][...], g[f"{case}/binary-segmentation"])
    np.testing.assert_array_equal(f["centered-embeddings"][...][..., ::4, ::4], g[f"{case}/centered-embeddings_s4"])
    det, ref = f["detection"][...], g[f"{case}/detection"]
    assert det.dtype == np.uint16 and det.shape == ref.shape
    for s in range(det.shape[0]):
        for b in range(det.shape[1]):          # cluster numbering = sort of near-tied centres
            np.testing.assert_array_equal(det[s, b] > 0, ref[s, b] > 0)
            np.testing.assert_array_equal(IO.label(det[s, b].astype(np.int32)), IO.label(ref[s, b].astype(np.int32)))
    np.testing.assert_array_equal(det, ref)    # ... and on these inputs the numbering agrees too
    for pp in ("cell", "nucleus"):
        # post-process the REFERENCE's detection so that this half does not depend on the first
        w = zarr_io.open(container)
        w["detection_ref"] = ref
        if "segmentation" in w:            # create_dataset refuses to replace a dataset, as zarr does
            del w["segmentation"]
        segment(_stage_config(container, g, case, post_processing=pp, segment_from="detection_ref"))
        seg = zarr_io.open(container, "r")["segmentation"][...]
        assert seg.dtype == np.uint16
        np.testing.assert_array_equal(seg, g[f"{case}/segmentation_{pp}"])


def test_detect_seeds_with_two_bandwidths_raises_like_reference(device, tmp_path, monkeypatch):
    """The reference re-reads the centred embeddings after the first bandwidth added pixel
    coordinates to them in place (detect.py:116-118,142-144); the second bandwidth then finds no
    point near any seed and sklearn raises ValueError.  Same inputs, same error here."""
    from cellulus_amd.detect import detect

    g = np.load(os.path.join(G, "g8_stages.npz"))
    case = "2d_seeds_bw2"
    assert str(g[f"{case}/detect_error"]).startswith("ValueError: No point was within bandwidth=5.0")
    monkeypatch.chdir(tmp_path)
    container = _stage_container(tmp_path, g, case)
    np.random.seed(int(g[f"{case}/params"][3]))
    with pytest.raises(ValueError, match="No point was within bandwidth=5.0"):
        detect(_stage_config(container, g, case))
    from cellulus_amd.utils import zarr_io
    np.testing.assert_array_equal(zarr_io.open(container, "r")["detection"][...], g[f"{case}/detection_partial"])


def test_nucleus_refine_on_device_matches_oracle_random(device):
    """Denser parity sweep for csrc/nucleus.hip: random label images with nested / touching
    instances, holes, constant instances; raw in every supported dtype, 2-D and 3-D."""
    from cellulus_amd.segment import _raw_to_device, nucleus_refine_on_device

    rng = np.random.default_rng(5)
    for shape in ((70, 90), (12, 40, 36)):
        for dtype in (np.float32, np.float64, np.uint8, np.uint16, np.int16):
            blocks = rng.integers(0, 9, size=tuple(-(-s // 10) for s in shape))
            seg = np.kron(blocks, np.ones((10,) * len(shape), dtype=np.int64))[tuple(slice(0, s) for s in shape)]
            seg[rng.random(shape) < 0.05] = 0
            seg = seg.astype(np.int32)
            raw = rng.random(shape)
            raw[rng.random(shape) < 0.3] *= 0.2                    # dark specks -> holes
            raw[seg == 3] = 0.5                                    # a constant instance
            if np.issubdtype(dtype, np.integer):
                raw = (raw * 200).astype(dtype) - (50 if dtype == np.int16 else 0)
            else:
                raw = raw.astype(dtype)
            ref = IO.nucleus_refine(seg, raw)
            raw_d, rt = _raw_to_device(raw, device)
            out = nucleus_refine_on_device(torch.from_numpy(seg).to(device), raw_d, rt)
            np.testing.assert_array_equal(out.cpu().numpy(), ref, err_msg=f"{shape} {dtype}")


def test_fused_infer_writes_the_same_datasets_as_the_staged_path(device, tmp_path, monkeypatch):
    """infer() hands embeddings and label maps from stage to stage in device memory (default) or
    goes through the zarr datasets stage by stage (CLX_FUSED_INFER=0, the reference's order): every
    dataset and attribute on disk is identical, for both post-processing modes."""
    from cellulus_amd.configs import ExperimentConfig
    from cellulus_amd.infer import infer
    from cellulus_amd.utils import zarr_io
    from oracle.unet_oracle import OracleUNetModel

    monkeypatch.chdir(tmp_path)
    mcfg = dict(num_fmaps=8, fmap_inc_factor=2, features_in_last_layer=16, downsampling_factors=[[2, 2]])
    torch.manual_seed(0)
    oracle = OracleUNetModel(in_channels=1, out_channels=2, num_spatial_dims=2, **mcfg)
    os.makedirs("models", exist_ok=True)
    torch.save({"model_state_dict": oracle.state_dict()}, "models/best_loss.pth")
    results = {}
    # (72, 80): the last tile of an axis is shifted back inside the image (tiles overlap: the Otsu range of the std channel
    # comes from its own pass); (80, 120): 2 x 3 tiles of 40 partition the image (the range is folded over the tiles' mean /
    # std launches, clx_noise_stats_minmax)
    # ... and the same partition with 33 noise iterations: 66 predictions per pixel are more than the statistics kernel
    # folds a range over (64) — the reference bounds num_infer_iterations nowhere (inference_config.py), the fused path
    # must then take the range from its own pass like the staged one
    for post, shape, n_it in (("cell", (2, 1, 72, 80), 2), ("nucleus", (2, 1, 72, 80), 2), ("cell", (2, 1, 80, 120), 2),
                              ("cell", (1, 1, 80, 120), 33)):
        tag = f"{post}_{shape[2]}x{shape[3]}_{n_it}"
        for fused in ("1", "0"):
            container = str(tmp_path / f"data_{tag}_{fused}.zarr")
            _write_raw(container, 2, shape=shape)
            cfg = ExperimentConfig(
                model_config=dict(checkpoint="models/best_loss.pth", **mcfg),
                object_size=12, normalization_factor=1.0,
                inference_config=dict(
                    dataset_config=dict(container_path=container, dataset_name="test/raw"),
                    prediction_dataset_config=dict(container_path=container, dataset_name="embeddings"),
                    detection_dataset_config=dict(container_path=container, dataset_name="detection",
                                                  secondary_dataset_name="embeddings"),
                    segmentation_dataset_config=dict(container_path=container, dataset_name="segmentation",
                                                     secondary_dataset_name="detection"),
                    crop_size=[56, 56], num_infer_iterations=n_it, p_salt_pepper=0.05, num_bandwidths=2,
                    reduction_probability=0.5, min_size=6, grow_distance=2, shrink_distance=3,
                    post_processing=post, device="cuda:0"))
            monkeypatch.setenv("CLX_FUSED_INFER", fused)
            torch.manual_seed(42)
            np.random.seed(42)
            infer(cfg)
            f = zarr_io.open(container, "r")
            results[tag, fused] = {name: (f[name][...], dict(f[name].attrs)) for name in
                                   ("embeddings", "detection", "binary-segmentation", "centered-embeddings",
                                    "segmentation")}
        for name, (data, attrs) in results[tag, "1"].items():
            ref_data, ref_attrs = results[tag, "0"][name]
            assert data.dtype == ref_data.dtype and data.shape == ref_data.shape, name
            np.testing.assert_array_equal(data, ref_data, err_msg=f"{tag}/{name}")
            assert {k: list(v) if isinstance(v, (list, tuple)) else v for k, v in attrs.items()} == \
                   {k: list(v) if isinstance(v, (list, tuple)) else v for k, v in ref_attrs.items()}, name
        assert results[tag, "1"]["segmentation"][0].max() > 0


# ------------------------------------------------------------------ evaluate (joint histogram on the device)
def test_evaluate_on_device_matches_real_reference_golden(device):
    """g10: IoU table, SEG, F1, TP, FP, FN of the REAL cellulus.evaluate functions (bit for bit: every
    entry is a quotient of the same two integers)."""
    from cellulus_amd.evaluate import compute_F1, compute_pairwise_IoU

    g = np.load(os.path.join(G, "g10_evaluate.npz"))
    for i in range(3):
        iou, seg, n = compute_pairwise_IoU(g[f"{i}/pred"], g[f"{i}/gt"])
        np.testing.assert_array_equal(iou, g[f"{i}/iou"])
        f1, tp, fp, fn = compute_F1(iou)
        np.testing.assert_allclose([seg, n, f1, tp, fp, fn], g[f"{i}/scalars"], rtol=1e-15)
    assert compute_pairwise_IoU(g["0/pred"], np.zeros_like(g["0/gt"])) is None


@pytest.mark.parametrize("shape", [(512, 512), (37, 53), (24, 40, 56), (1, 7)])
def test_joint_histogram_matches_oracle_random(shape, device):
    """Blocky random label maps with sparse ids up to 65535 (uint16 storage), odd extents (tails of
    the 8-pixel runs), background present or absent: ids and counts equal np.unique / np.add.at."""
    from cellulus_amd.evaluate import iou_from_joint, joint_histogram_on_device

    rng = np.random.default_rng(sum(shape))
    for trial in range(3):
        def blocky(nids, block):
            ids = np.concatenate([[0] if trial != 1 else [], rng.choice(np.arange(1, 65536), size=nids, replace=False)])
            coarse = tuple(-(-s // block) for s in shape)
            m = ids[rng.integers(0, len(ids), size=coarse)]
            for ax in range(len(shape)):
                m = np.repeat(m, block, axis=ax)
            return np.ascontiguousarray(m[tuple(slice(0, s) for s in shape)]).astype(np.uint16)

        pred, gt = blocky(40, 5), blocky(25, 7)
        p_ids, g_ids, joint = joint_histogram_on_device(pred, gt)
        rp, rg, rj = IO.joint_histogram(pred, gt)
        np.testing.assert_array_equal(p_ids, rp)
        np.testing.assert_array_equal(g_ids, rg)
        np.testing.assert_array_equal(joint, rj)
        assert joint.sum() == pred.size
        got, ref = iou_from_joint(p_ids, g_ids, joint), IO.compute_pairwise_IoU(pred, gt) if pred.size < 5000 else None
        if ref is not None and got is not None:
            np.testing.assert_array_equal(got[0], ref[0])
            assert got[1] == ref[1] and got[2] == ref[2]


def test_joint_histogram_takes_device_tensors_and_rejects_wide_ids(device):
    from cellulus_amd.evaluate import joint_histogram_on_device

    a = torch.randint(0, 9, (64, 64), device=device, dtype=torch.int32)
    p_ids, g_ids, joint = joint_histogram_on_device(a, a)
    assert np.array_equal(p_ids, g_ids) and np.array_equal(joint, np.diag(np.diag(joint)))
    np.testing.assert_array_equal(np.diag(joint), np.bincount(a.cpu().numpy().ravel())[p_ids])
    a[3, 3] = 70000
    with pytest.raises(ValueError, match="65536"):
        joint_histogram_on_device(a, a)


def test_evaluate_over_zarr_writes_the_reference_report(device, tmp_path, monkeypatch):
    """evaluate(): results_bandwidth-<b>.txt (evaluate.py:36-69) for two bandwidths over zarr label maps;
    every number against the oracle's mask-loop restatement, samples without ground truth skipped."""
    from cellulus_amd.configs import InferenceConfig
    from cellulus_amd.evaluate import evaluate
    from cellulus_amd.utils import zarr_io

    monkeypatch.chdir(tmp_path)
    container = str(tmp_path / "eval.zarr")
    rng = np.random.default_rng(3)
    S, shape = 3, (48, 40)

    def blocky(nids, block):
        m = rng.integers(0, nids + 1, size=tuple(-(-s // block) for s in shape))
        return np.kron(m, np.ones((block, block), dtype=np.int64))[:shape[0], :shape[1]]

    gt = np.stack([blocky(5, 8), np.zeros(shape, np.int64), blocky(4, 6)])[:, None].astype(np.uint16)
    seg = np.stack([np.stack([np.roll(gt[s, 0], b + 1, axis=1) for b in range(2)]) for s in range(S)]).astype(np.uint16)
    f = zarr_io.open(container)
    f["raw"] = rng.random((S, 1) + shape).astype(np.float32)
    f["raw"].attrs["axis_names"] = ["s", "c", "y", "x"]
    f["gt"] = gt
    f["segmentation"] = seg
    cfg = InferenceConfig(
        dataset_config=dict(container_path=container, dataset_name="raw"),
        evaluation_dataset_config=dict(container_path=container, dataset_name="gt",
                                       secondary_dataset_name="segmentation"),
        crop_size=[32, 32], num_bandwidths=2, device="cuda:0")
    evaluate(cfg)
    for b in range(2):
        lines = open(f"results_bandwidth-{b}.txt").read().splitlines()
        rows = [ln for ln in lines if ln[:1].isdigit()]
        assert [int(r.split(",")[0]) for r in rows] == [0, 2]              # sample 1 has no ground truth
        tps = fps = fns = 0
        seg_sum = n_ids = 0
        for r, s in zip(rows, (0, 2)):
            iou, seg_img, n = IO.compute_pairwise_IoU(seg[s, b], gt[s, 0])
            f1, tp, fp, fn = IO.compute_F1(iou)
            assert r == f"{s}, {f1:.05f}, {seg_img / n:.05f}, {tp}, {fp}, {fn}"
            tps, fps, fns, seg_sum, n_ids = tps + tp, fps + fp, fns + fn, seg_sum + seg_img, n_ids + n
        assert lines[-2] == f"F1 for complete dataset is {2 * tps / (2 * tps + fps + fns):.05f} "
        assert lines[-1] == f"SEG for complete dataset is {seg_sum / n_ids:.05f} "


# --------------------------------------------- float32 hand-over of the fused predict -> detect path
@pytest.mark.parametrize("shape", [(64, 80), (61, 67), (129, 1031), (12, 20, 24), (7, 9, 11)])
def test_float32_handover_kernels_equal_the_float64_ones_on_the_widened_values(shape, device):
    """clx_ms_prepare_f32 / clx_minmax_f32 / clx_histogram_f32 read the network's float32 output and widen in
    registers: points, raster indices, foreground count, min / max and bin counts are BIT-identical to the float64
    entry points on the widened tensors (what the staged path reads back from the `embeddings` dataset), for aligned
    and unaligned extents, and the float32 embedding is left untouched."""
    from cellulus_amd import _clx

    nd = len(shape)
    rng = np.random.default_rng(5)
    emb32 = torch.from_numpy((rng.normal(0, 6, size=(nd,) + shape)).astype(np.float32)).to(device)
    std32 = torch.from_numpy(rng.random(shape, dtype=np.float32)).to(device)
    Z, Y, X = (1,) * (3 - nd) + shape
    npix = Z * Y * X
    st = _clx.stream_ptr(device)
    lib = _clx.load()
    thr = 0.3

    def prepare(name, emb, std):
        ws = torch.full((int(lib.clx_ms_prepare_workspace(npix)),), 0xA5, dtype=torch.uint8, device=device)   # plain scratch
        pts = torch.full((npix, nd), float("nan"), dtype=torch.float64, device=device)
        idx = torch.full((npix,), -1, dtype=torch.int32, device=device)
        nfg = torch.zeros(1, dtype=torch.int32, device=device)
        for _ in range(2):                                   # (the workspace carries no state from call to call)
            e = emb.clone()
            _clx.call(name, _clx.ptr(e), _clx.ptr(std), thr, nd, Z, Y, X, _clx.ptr(pts), _clx.ptr(idx), _clx.ptr(nfg),
                      _clx.ptr(ws), st)
        n = int(nfg.item())
        return pts[:n].cpu().numpy(), idx[:n].cpu().numpy(), e

    p64, i64, e64 = prepare("clx_ms_prepare", emb32.double(), std32.double())
    p32, i32, e32 = prepare("clx_ms_prepare_f32", emb32, std32)
    assert len(i64) == int((std32.double() < thr).sum().item()) > 0
    np.testing.assert_array_equal(i32, i64)
    np.testing.assert_array_equal(p32, p64)
    assert torch.equal(e32, emb32) and not torch.equal(e64, emb32.double())     # only the float64 form adds in place
    # Otsu primitives
    for x in (std32.reshape(-1), std32.reshape(-1)[: npix - 3]):
        x = x.clone()
        mm32 = torch.empty(_clx.MINMAX_DOUBLES, dtype=torch.float64, device=device)
        mm64 = torch.empty(_clx.MINMAX_DOUBLES, dtype=torch.float64, device=device)
        _clx.call("clx_minmax_f32", _clx.ptr(x), x.numel(), _clx.ptr(mm32), st)
        _clx.call("clx_minmax_f64", _clx.ptr(x.double()), x.numel(), _clx.ptr(mm64), st)
        assert torch.equal(mm32[:2], mm64[:2]) and mm32[0].item() == float(x.min().item()) and mm32[1].item() == float(x.max().item())
        c32, e32_ = histogram_on_device(x)
        c64, e64_ = histogram_on_device(x.double())
        np.testing.assert_array_equal(c32, c64)
        np.testing.assert_array_equal(e32_, e64_)
        ref_counts, _ = np.histogram(x.cpu().numpy().astype(np.float64), bins=256)
        np.testing.assert_array_equal(c32, ref_counts)
    assert threshold_otsu(std32) == threshold_otsu(std32.double())
    mm = (float(std32.min().item()), float(std32.max().item()))
    assert threshold_otsu(std32, minmax=mm) == threshold_otsu(std32.double())
    assert threshold_otsu(torch.full((5, 7), 1.25, dtype=torch.float32, device=device)) == 1.25


def test_noise_stats_emits_the_std_range(device):
    """clx_noise_stats_minmax: the same (mean, std) planes as clx_noise_stats, plus the running minimum / maximum of the
    std plane, folded over several calls (the tiles of one image) when init is 0."""
    from cellulus_amd import _clx

    st = _clx.stream_ptr(device)
    torch.manual_seed(3)
    T, C = 32, 2
    mm = torch.full((_clx.NOISE_MINMAX_FLOATS,), float("nan"), dtype=torch.float32, device=device)
    lo, hi = float("inf"), 0.0
    for k, n in enumerate((4096, 1000, 333 * 7)):
        preds = torch.randn(T, C, n, device=device) * (k + 1)
        if k == 1:
            preds[:, :, 17] = 0.25                      # a pixel whose std is exactly 0
        ref = torch.empty(C + 1, n, device=device)
        out = torch.empty(C + 1, n, device=device)
        _clx.call("clx_noise_stats", _clx.ptr(preds), _clx.ptr(ref), T, C, n, st)
        _clx.call("clx_noise_stats_minmax", _clx.ptr(preds), _clx.ptr(out), T, C, n, _clx.ptr(mm), 1 if k == 0 else 0, st)
        assert torch.equal(out, ref)
        lo, hi = min(lo, float(ref[C].min().item())), max(hi, float(ref[C].max().item()))
        assert mm[:2].cpu().tolist() == [lo, hi]
    assert lo == 0.0
    # reset
    _clx.call("clx_noise_stats_minmax", _clx.ptr(preds), _clx.ptr(out), T, C, n, _clx.ptr(mm), 1, st)
    assert mm[:2].cpu().tolist() == [float(ref[C].min().item()), float(ref[C].max().item())]


@pytest.mark.parametrize("K,extent,bw", [(3000, 400.0, 25.0), (40, 30.0, 25.0), (1, 10.0, 5.0), (700, 2000.0, 3.0)])
def test_assign_cells_candidate_sequence_dense_cells_and_tiny_grids(K, extent, bw, device):
    """The 2-D form of clx_ms_assign_cells walks the 3 x 3 block's candidates as one sequence, two per trip: blocks holding
    many centres (odd and even counts), grids of one or two cells per axis (clamped rows), blocks that hold none (the
    doubling search takes over) — all equal to the plain loop over every centre."""
    import ctypes

    from cellulus_amd import _clx
    from cellulus_amd.utils import mean_shift as MS

    rng = np.random.default_rng(K)
    centers = rng.uniform(0, extent, size=(K, 2))
    pts = np.concatenate([rng.uniform(-0.2 * extent, 1.2 * extent, size=(30000, 2)),
                          centers[rng.integers(0, K, size=5000)] + rng.normal(0, 0.3 * bw, size=(5000, 2))])
    n = len(pts)
    X = torch.from_numpy(pts).to(device)
    index = torch.from_numpy(rng.permutation(n).astype(np.int32)).to(device)
    st = _clx.stream_ptr(device)
    ref = torch.zeros(n, dtype=torch.int32, device=device)
    cc = torch.from_numpy(centers).to(device)
    _clx.call("clx_ms_assign", _clx.ptr(X), _clx.ptr(index), n, _clx.ptr(cc), K, 2, _clx.ptr(ref), st)
    order, cstart, origin, (gx, gy, gz) = MS._center_grid(centers, bw)
    got = torch.zeros(n, dtype=torch.int32, device=device)
    order_d, cstart_d = torch.from_numpy(order).to(device), torch.from_numpy(cstart).to(device)
    cc_sorted = torch.from_numpy(np.ascontiguousarray(centers[order])).to(device)
    _clx.call("clx_ms_assign_cells", _clx.ptr(X), _clx.ptr(index), n, _clx.ptr(cc_sorted), K, 2, _clx.ptr(order_d),
              _clx.ptr(cstart_d), (ctypes.c_double * 2)(*origin.tolist()), bw, gx, gy, gz, _clx.ptr(got), st)
    assert torch.equal(got, ref)
    d2 = ((pts[:, None, :] - centers[None]) ** 2).sum(-1)
    want = np.zeros(n, dtype=np.int32)
    want[index.cpu().numpy()] = d2.argmin(1) + 1
    np.testing.assert_array_equal(ref.cpu().numpy(), want)


@pytest.mark.parametrize("shape", [(1, 64, 64), (2, 33, 47), (1, 7, 9, 11)])
def test_noise_inject_equals_the_torch_expression(shape, device):
    """clx_noise_inject == torch.where(rnd <= p, [0.5] * n + [1.0] * n, raw) — the comparison in float32 like torch's
    (p = 0.1 rounds UP in float32: a draw of exactly float32(0.1) counts as noise)."""
    from cellulus_amd import _clx

    torch.manual_seed(5)
    n_it = 3
    T = 2 * n_it
    raw = torch.rand((1,) + shape, device=device)
    rnd = torch.rand((T,) + shape, device=device)
    for p in (0.01, 0.1, 0.5):
        rnd.view(-1)[::7] = float(np.float32(p))
        vals = torch.tensor([0.5] * n_it + [1.0] * n_it, device=device).view((T,) + (1,) * len(shape))
        want = torch.where(rnd <= p, vals, raw.expand_as(rnd))
        got = torch.empty_like(rnd)
        _clx.call("clx_noise_inject", _clx.ptr(rnd), _clx.ptr(raw), _clx.ptr(got), T, n_it, raw.numel(), p,
                  _clx.stream_ptr(device))
        assert torch.equal(got, want)
        assert (got != raw.expand_as(rnd)).any()


def test_zero_many_and_row_gather(device):
    from cellulus_amd import _clx

    sizes = [1, 3, 4, 5, 1024, 4099, 1 << 20, 7, 2, 65537, 12]          # more than eight buffers: two launches
    bufs = [torch.full((s,), 7.0, dtype=torch.float32, device=device) for s in sizes]
    guard = [torch.full((s + 8,), 3.0, dtype=torch.float32, device=device) for s in sizes]
    views = [g[4:4 + s] for g, s in zip(guard, sizes)]                     # 16-byte aligned views inside a guard band
    i64 = torch.full((257,), 9, dtype=torch.int64, device=device)
    _clx.zero_many(*bufs, None, i64, *views)
    for b in bufs + views + [i64]:
        assert int(torch.count_nonzero(b).item()) == 0
    for g, s in zip(guard, sizes):
        assert g[:4].eq(3.0).all() and g[4 + s:].eq(3.0).all()
    rng = np.random.default_rng(0)
    for nd in (2, 3):
        src = torch.from_numpy(rng.normal(size=(5000, nd))).to(device)
        rows = torch.from_numpy(np.flatnonzero(rng.random(5000) < 0.1).astype(np.int32)).to(device)
        dst = torch.empty((rows.shape[0], nd), dtype=torch.float64, device=device)
        _clx.call("clx_gather_rows_f64", _clx.ptr(src), _clx.ptr(rows), rows.shape[0], nd, _clx.ptr(dst),
                  _clx.stream_ptr(device))
        assert torch.equal(dst, src[rows.long()])


@pytest.mark.parametrize("shape", [(64, 64), (97, 33), (5, 301), (130, 1024), (1000, 1030), (9, 20, 31), (16, 64, 64)])
@pytest.mark.parametrize("f32", [False, True])
def test_dense_assignment_equals_scatter_into_a_zeroed_map(shape, f32, device):
    """clx_ms_assign_dense (whole label map from the compaction's tiles, read back from the prepare workspace) ==
    zero fill + clx_ms_assign_cells through the raster index: odd image sizes (ragged last tile, unaligned rows), tiles
    without foreground, tiles that are all foreground, both prepare forms, with and without the raster index."""
    import ctypes

    from cellulus_amd import _clx
    from cellulus_amd.utils import mean_shift as MS

    nd = len(shape)
    rng = np.random.default_rng(sum(shape) + int(f32))
    npix = int(np.prod(shape))
    Z, Y, X = (1,) * (3 - nd) + tuple(shape)
    emb_np = rng.normal(0, 3.0, size=(nd,) + tuple(shape))
    std_np = rng.uniform(0, 1, size=shape)
    std_np.reshape(-1)[: min(npix, 2500)] = 0.9            # leading tiles without foreground ...
    std_np.reshape(-1)[npix // 2: npix // 2 + min(npix // 4, 3000)] = 0.1        # ... and a stretch that is all foreground
    dt = torch.float32 if f32 else torch.float64
    emb = torch.from_numpy(emb_np).to(device=device, dtype=dt)
    std = torch.from_numpy(std_np).to(device=device, dtype=dt)
    lib = _clx.load()
    st = _clx.stream_ptr(device)
    ws = torch.empty(int(lib.clx_ms_prepare_workspace(npix)), dtype=torch.uint8, device=device)
    pts = torch.empty((npix, nd), dtype=torch.float64, device=device)
    index = torch.empty(npix, dtype=torch.int32, device=device)
    nfg_d = torch.zeros(1, dtype=torch.int32, device=device)
    prepare = "clx_ms_prepare_f32" if f32 else "clx_ms_prepare"
    _clx.call(prepare, _clx.ptr(emb.clone()), _clx.ptr(std), 0.5, nd, Z, Y, X, _clx.ptr(pts), _clx.ptr(index),
              _clx.ptr(nfg_d), _clx.ptr(ws), st)
    nfg = int(nfg_d.item())
    assert 0 < nfg < npix
    K, bw = 60, 4.0
    centers = np.stack([rng.uniform(0, s, size=K) for s in shape[::-1]], axis=1)
    order, cstart, origin, (gx, gy, gz) = MS._center_grid(centers, bw)
    order_d, cstart_d = torch.from_numpy(order).to(device), torch.from_numpy(cstart).to(device)
    cc_sorted = torch.from_numpy(np.ascontiguousarray(centers[order])).to(device)
    origin_c = (ctypes.c_double * nd)(*origin.tolist())
    want = torch.zeros(shape, dtype=torch.int32, device=device)
    _clx.call("clx_ms_assign_cells", _clx.ptr(pts), _clx.ptr(index), nfg, _clx.ptr(cc_sorted), K, nd, _clx.ptr(order_d),
              _clx.ptr(cstart_d), origin_c, bw, gx, gy, gz, _clx.ptr(want), st)
    got = torch.full(shape, -7, dtype=torch.int32, device=device)
    _clx.call("clx_ms_assign_dense", _clx.ptr(pts), _clx.ptr(cc_sorted), K, nd, _clx.ptr(order_d), _clx.ptr(cstart_d),
              origin_c, bw, gx, gy, gz, _clx.ptr(ws), 1 if f32 else 0, Z, Y, X, _clx.ptr(got), st)
    assert torch.equal(got, want)
    assert int((want > 0).sum().item()) == nfg
    # the compaction without the raster index leaves the same points and the same workspace
    pts2 = torch.empty_like(pts)
    _clx.call(prepare, _clx.ptr(emb.clone()), _clx.ptr(std), 0.5, nd, Z, Y, X, _clx.ptr(pts2), None,
              _clx.ptr(nfg_d), _clx.ptr(ws), st)
    assert int(nfg_d.item()) == nfg and torch.equal(pts2[:nfg], pts[:nfg])
    got.fill_(-7)
    _clx.call("clx_ms_assign_dense", _clx.ptr(pts2), _clx.ptr(cc_sorted), K, nd, _clx.ptr(order_d), _clx.ptr(cstart_d),
              origin_c, bw, gx, gy, gz, _clx.ptr(ws), 1 if f32 else 0, Z, Y, X, _clx.ptr(got), st)
    assert torch.equal(got, want)


@pytest.mark.parametrize("n,nd", [(1, 2), (255, 2), (4500, 2), (70000, 2), (300000, 3), (17, 3)])
def test_rows_extent_equals_numpy_min_max(n, nd, device):
    from cellulus_amd import _clx

    rng = np.random.default_rng(n)
    src_np = rng.normal(0, 100, size=(n, nd))
    src = torch.from_numpy(src_np).to(device)
    ext = torch.full((_clx.ROWS_EXTENT_DOUBLES,), float("nan"), dtype=torch.float64, device=device)
    _clx.call("clx_rows_extent_f64", _clx.ptr(src), n, nd, _clx.ptr(ext), _clx.stream_ptr(device))
    got = ext[:2 * nd].cpu().numpy().reshape(2, nd)
    np.testing.assert_array_equal(got, np.stack([src_np.min(axis=0), src_np.max(axis=0)]))


@pytest.mark.parametrize("lo,hi", [(0.0, 1.0), (-3.5, 7.25), (1e6, 1e6 + 1e-3), (-1e-9, 1e-9), (5.0, 5.0 + 2.0 ** -30)])
def test_histogram_bins_equal_numpy_on_and_next_to_the_edges(lo, hi, device):
    """clx_histogram_f64 / _f32 guess a value's bin by one multiplication and look at the edges only near a boundary:
    values exactly on every edge, one ulp to either side of it, and a range that is tiny against its offset (where an edge's
    own rounding is many bins' worth of 1e-6) must land where np.histogram puts them."""
    rng = np.random.default_rng(7)
    edges = np.linspace(lo, hi, 257)
    vals = np.concatenate([edges, np.nextafter(edges, -np.inf), np.nextafter(edges, np.inf),
                           rng.uniform(lo, hi, size=200001), np.full(33, lo), np.full(33, hi)])
    vals = np.clip(vals, lo, hi)
    rng.shuffle(vals)
    want, want_edges = np.histogram(vals, bins=256)
    got, got_edges = histogram_on_device(torch.from_numpy(vals).to(device))
    np.testing.assert_array_equal(got_edges, want_edges)
    np.testing.assert_array_equal(got, want)
    v32 = vals.astype(np.float32)
    if np.unique(v32).size > 300:                       # (the float32 form: the widened floats, their own range)
        want32, _ = np.histogram(v32.astype(np.float64), bins=256)
        got32, _ = histogram_on_device(torch.from_numpy(v32).to(device))
        np.testing.assert_array_equal(got32, want32)
