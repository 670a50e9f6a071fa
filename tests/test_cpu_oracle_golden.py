"""CPU tier: the oracle (oracle/) is pinned against golden vectors produced by the
REAL reference code (tests/golden/make_golden*.py).  No GPU, no product code."""

import os

import numpy as np
import pytest
import torch

from oracle import infer_oracle as IO
from oracle import unet_oracle as UO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def test_oce_loss_oracle_matches_reference():
    g = _load("g1_oce_loss.npz")
    for nd in (2, 3):
        a = torch.from_numpy(g[f"a{nd}"]).requires_grad_(True)
        r = torch.from_numpy(g[f"r{nd}"])
        loss, oce, reg = UO.oce_loss(a, r, 10.0, 1e-5)
        loss.backward()
        np.testing.assert_allclose([loss.item(), oce.item(), reg.item()], g[f"sums{nd}"], rtol=1e-6)
        np.testing.assert_allclose(a.grad.numpy(), g[f"grad{nd}"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(g["kat_sums"], [1.2476590872, 1.2475949526, 6.41421e-5], rtol=1e-5)


def test_gather_oracle_matches_reference():
    g = _load("g2_gather.npz")
    for nd in (2, 3):
        sel = UO.select_and_add_coordinates(torch.from_numpy(g[f"offsets{nd}"]), torch.from_numpy(g[f"coords{nd}"]))
        np.testing.assert_array_equal(sel.numpy(), g[f"sel{nd}"])


@pytest.mark.parametrize("nd", [2, 3])
def test_unet_oracle_matches_reference_wrapper(nd):
    """Reference UNetModel (real head + infer loop, stubbed backbone) == OracleUNetModel."""
    g = _load("g3_unet.npz")
    cfg = dict(in_channels=1, out_channels=nd, num_fmaps=4, fmap_inc_factor=2, features_in_last_layer=8,
               downsampling_factors=[(2,) * nd], num_spatial_dims=nd)
    model = UO.OracleUNetModel(**cfg)
    sd = {k[len(f"w{nd}/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"w{nd}/")}
    model.load_state_dict(sd, strict=True)
    raw = torch.from_numpy(g[f"raw{nd}"])
    with torch.no_grad():
        out = model(raw)
    assert out.shape == tuple(g[f"train{nd}"].shape)
    assert tuple(out.shape[2:]) == tuple(s - 16 for s in raw.shape[2:])   # zarr_dataset.py:94
    np.testing.assert_allclose(out.numpy(), g[f"train{nd}"], atol=1e-6)
    model.set_infer(0.05, 2)
    torch.manual_seed(7)
    with torch.no_grad():
        inf = model(raw)
    np.testing.assert_allclose(inf.numpy(), g[f"infer{nd}"], atol=1e-6)


MS_CASES = ["2d_rp1", "2d_rp05", "2d_rp02", "2d_seeds", "3d_rp05", "2d_empty"]


@pytest.mark.parametrize("case", MS_CASES)
def test_mean_shift_oracle_matches_reference(case):
    g = _load("g4_mean_shift.npz")
    bw, rp, thr, seed = g[f"{case}/params"]
    mean = g[f"{case}/mean"].copy()
    seeds = g[f"{case}/seeds"] if f"{case}/seeds" in g.files else None
    np.random.seed(int(seed))
    labels = IO.mean_shift_segmentation(mean, g[f"{case}/std"], bw, 10, rp, thr, seeds)
    ref = g[f"{case}/labels"]
    assert labels.dtype == ref.dtype == np.int32
    np.testing.assert_array_equal(mean, g[f"{case}/mean_after"])      # in-place coordinate add
    # cluster numbering follows sklearn's sort of near-tied centres; the PARTITION is the contract
    np.testing.assert_array_equal(labels > 0, ref > 0)
    np.testing.assert_array_equal(IO.label(labels), IO.label(ref))
    # ... and on these inputs even the numbering agrees
    np.testing.assert_array_equal(labels, ref)


SK_CASES = ["2d_rp1", "2d_rp02", "3d_rp05", "rand2d", "noise2d", "noise3d", "zeros2d", "full2d"]


@pytest.mark.parametrize("case", SK_CASES)
def test_label_and_size_filter_oracle_match_skimage(case):
    g = _load("g5_skimage.npz")
    seg = g[f"{case}/seg"]
    np.testing.assert_array_equal(IO.label(seg), g[f"{case}/label"])
    for ms in (1, 4, 30):
        np.testing.assert_array_equal(IO.size_filter(seg.copy(), ms), g[f"{case}/size_filter_{ms}"])


def test_otsu_oracle_matches_skimage():
    g = _load("g5_skimage.npz")
    for i in range(3):
        assert IO.threshold_otsu(g[f"otsu{i}/image"]) == float(g[f"otsu{i}/threshold"])


def test_edt_sq_matches_scipy_semantics():
    rng = np.random.default_rng(0)
    m = rng.random((20, 30)) < 0.9
    d = IO.edt_sq(m)
    assert d[~m].max() == 0 and d[m].min() >= 1
    # no zero anywhere: scipy reports the distance to a phantom zero at index -1 of axis 0
    ones = np.ones((4, 5), dtype=bool)
    yy, xx = np.mgrid[0:4, 0:5]
    np.testing.assert_array_equal(IO.edt_sq(ones), (yy + 1) ** 2 + xx ** 2)
    ones3 = np.ones((3, 4, 5), dtype=bool)
    zz, yy, xx = np.mgrid[0:3, 0:4, 0:5]
    np.testing.assert_array_equal(IO.edt_sq(ones3), (zz + 1) ** 2 + yy ** 2 + xx ** 2)


@pytest.mark.parametrize("case", ["2d", "3d"])
def test_greedy_cluster_oracle_matches_reference(case):
    """numpy restatement == the real Cluster2d / Cluster3d of cellulus/utils/greedy_cluster.py."""
    g = _load("g6_greedy.npz")
    bw, ms = g[f"{case}/params"]
    out = IO.greedy_cluster(g[f"{case}/pred"], g[f"{case}/fg"], float(bw), int(ms))
    assert out.dtype == np.int16 and out.max() >= 4
    np.testing.assert_array_equal(out, g[f"{case}/seg"])


STAGE_CASES = ["2d_f32", "2d_u8_seeds", "2d_f64", "3d_u16"]


@pytest.mark.parametrize("case", STAGE_CASES)
def test_stage_oracle_matches_reference_detect_and_segment(case):
    """g8: the REAL cellulus.detect.detect / cellulus.segment.segment (cell + nucleus), run end to
    end over an in-memory zarr stand-in (tests/golden/make_golden_stages.py), vs the oracle."""
    g = _load("g8_stages.npz")
    bw, ms, rp, seed, nb, use_seeds = g[f"{case}/params"]
    emb, raw = g[f"{case}/embeddings"], g[f"{case}/raw"]
    np.random.seed(int(seed))
    for s in range(emb.shape[0]):
        mask, centred, labels = IO.detect_sample(emb[s], bw, int(nb), int(ms), rp, None, bool(use_seeds))
        np.testing.assert_array_equal(mask.astype(np.uint16), g[f"{case}/binary-segmentation"][s, 0])
        np.testing.assert_array_equal(centred[..., ::4, ::4], g[f"{case}/centered-embeddings_s4"][s])
        for b in range(int(nb)):
            det = g[f"{case}/detection"][s, b]
            np.testing.assert_array_equal(labels[b].astype(np.uint16), det)
            for pp in ("cell", "nucleus"):
                seg = IO.segment_sample(det.astype(np.int32), raw[s, 0], pp, 3, 6, int(ms))
                np.testing.assert_array_equal(seg, g[f"{case}/segmentation_{pp}"][s, b])
    # the nucleus refinement really changes something (holes filled, dim rim dropped)
    assert not np.array_equal(g[f"{case}/segmentation_nucleus"], g[f"{case}/segmentation_cell"])


def test_stage_oracle_reproduces_reference_error_with_seeds_and_two_bandwidths():
    """use_seeds + num_bandwidths=2: the reference's in-place coordinate add leaks into the second
    bandwidth (detect.py:116-118,142-144) and sklearn raises; the oracle keeps that behaviour."""
    g = _load("g8_stages.npz")
    case = "2d_seeds_bw2"
    bw, ms, rp, seed, nb, use_seeds = g[f"{case}/params"]
    assert str(g[f"{case}/detect_error"]).startswith("ValueError: No point was within bandwidth=5.0")
    np.random.seed(int(seed))
    with pytest.raises(ValueError, match="No point was within bandwidth=5.0"):
        IO.detect_sample(g[f"{case}/embeddings"][0], bw, int(nb), int(ms), rp, None, bool(use_seeds))


@pytest.mark.parametrize("nd", [2, 3])
def test_train_step_oracle_matches_reference_train_iteration(nd):
    """g9: the REAL cellulus.train.train_iteration + get_model + get_loss + Adam(weight_decay=0.01)
    (train.py:80-82,160-180), four iterations, vs the oracle's train_step."""
    g = _load("g9_train_iteration.npz")
    model = UO.OracleUNetModel(in_channels=1, out_channels=nd, num_fmaps=4, fmap_inc_factor=2,
                               features_in_last_layer=8, downsampling_factors=[(2,) * nd], num_spatial_dims=nd)
    pre = f"w{nd}/init/"
    model.load_state_dict({k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}, strict=True)
    opt = torch.optim.Adam(model.parameters(), lr=float(g["lr"]), weight_decay=0.01)
    for it in range(4):
        b = [torch.from_numpy(g[f"b{nd}/{it}/{k}"]) for k in ("raw", "anchor", "reference")]
        loss, oce, offsets = UO.train_step(model, opt, *b, 10.0, 1e-5)
        np.testing.assert_allclose([loss, oce], g[f"losses{nd}"][it], rtol=1e-6)
        np.testing.assert_allclose(offsets.detach().numpy(), g[f"b{nd}/{it}/offsets"], atol=1e-6)
    post = f"w{nd}/final/"
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g[post + k], atol=1e-6, err_msg=k)


@pytest.mark.parametrize("nd", [2, 3])
def test_gemm_form_of_the_oracle_convolutions_equals_nn_conv(nd):
    """oracle.unet_oracle.gemm_convolutions (one dgemm per filter tap, used so that the float64
    oracle is fast enough at BASELINE sizes) against nn.Conv2d / nn.Conv3d in float64: values and
    every parameter gradient to rounding."""
    import torch

    from oracle.unet_oracle import OracleUNetModel, gemm_convolutions

    cfg = dict(in_channels=2, out_channels=nd, num_fmaps=6, fmap_inc_factor=3, features_in_last_layer=8,
               downsampling_factors=[[2] * nd], num_spatial_dims=nd)
    torch.manual_seed(0)
    m = OracleUNetModel(**cfg).double()
    x = torch.rand(2, 2, *((44, 52) if nd == 2 else (28, 24, 32)), dtype=torch.float64)
    y = m(x)
    g = torch.randn_like(y)
    y.backward(g)
    ref = [p.grad.clone() for p in m.parameters()]
    for p in m.parameters():
        p.grad = None
    with gemm_convolutions(m):
        y2 = m(x)
        y2.backward(g)
    assert (y - y2).abs().max().item() < 1e-13
    for p, r in zip(m.parameters(), ref):
        assert ((p.grad - r).abs().max() / r.abs().max()).item() < 1e-12
    assert torch.equal(m(x), y)            # the patch is gone after the with-block
