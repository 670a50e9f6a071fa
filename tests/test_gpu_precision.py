"""Both arithmetics of the plain products stay on the record: the default (CLX_PRECISION=f32x3bf16: an exact three-way
bfloat16 split of the float32 operands, six products per float32 product on the bf16 matrix cores, csrc/gemm_sp.hip)
runs everywhere else in this suite; here the float32-MFMA kernels of rounds 1-5 (CLX_PRECISION=f32) go through the
same bars — the full-size trained-scale parity against the float32 and float64 oracles, the Winograd layers against
float64, the backward pass on forced decisions, and the sparse inference paths bit for bit.

Replaces the same reference arithmetic: nn.Conv{2,3}d in float32 (cellulus/models/unet.py:24-63, cellulus/train.py:178).
"""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _float32_mfma(monkeypatch):
    monkeypatch.setenv("CLX_PRECISION", "f32")


def test_switch_selects_the_kernels(device, monkeypatch):
    """the plan of a 128-channel network carries weight planes in the default precision and none with CLX_PRECISION=f32;
    an unknown name is an error; the reproducible mode selects float32"""
    import torch

    from cellulus_amd.models import get_model
    from cellulus_amd.models import plan as P

    cfg = dict(in_channels=1, out_channels=2, num_fmaps=128, fmap_inc_factor=1, features_in_last_layer=64,
               downsampling_factors=[[2, 2]], num_spatial_dims=2)
    x = torch.rand(1, 1, 64, 64, device=device)
    with torch.no_grad():
        model = get_model(**cfg).to(device)
        a = model(x).clone()
        pl = next(iter(model._plans.values()))
        assert pl.precision == 0 and not pl._wplanes
        monkeypatch.delenv("CLX_PRECISION")
        assert P.precision_name() == "f32x3bf16" == P.DEFAULT_PRECISION
        model._plans = {}
        b = model(x)
        pl = next(iter(model._plans.values()))
        assert pl.precision == 1 and pl._wplanes
    # two float32 results of the same network: a few 1e-7 of the output range apart
    assert (a - b).abs().max().item() < 1e-5 * max(1.0, a.abs().max().item())
    monkeypatch.setenv("CLX_PRECISION", "bf16")
    with pytest.raises(ValueError):
        P.precision_name()
    monkeypatch.setenv("CLX_PRECISION", "f32x3bf16")
    monkeypatch.setenv("CLX_DETERMINISTIC", "1")
    assert P.precision_code() == 0


def test_full_size_trained_scale_parity_in_float32_mfma(device, monkeypatch):
    import test_gpu_fullsize_oracle as T

    T.test_full_size_forward_and_gradients_match_the_oracle("cfg2-trained-scale", T.CFG2, (256, 256), 0.0, device, monkeypatch)


@pytest.mark.parametrize("name", ["2d_wide", "2d_chain64"])
def test_backward_in_float32_mfma(name, device):
    import test_gpu_unet as T

    T.test_forward_matches_oracle(name, device)
    T.test_backward_matches_oracle(name, device)


@pytest.mark.parametrize("name", ["2d_wide", "2d_96"])
def test_sparse_inference_paths_in_float32_mfma(name, device, monkeypatch):
    import test_gpu_unet as T

    T.test_noisy_copies_through_changed_rows_equal_the_dense_forward_bit_for_bit(name, device, monkeypatch)
