"""GPU tier: NUMERICAL parity with the oracle at BASELINE.json's real sizes.

* cfg-2 (2-D 256^2, num_fmaps=256, fmap_inc_factor=3) and cfg-4 (3-D 64^3, num_fmaps=64) at batch 1,
  on the DEFAULT launch plan (Winograd F(4x4) + sub-pixel upsample convolution + V cache): forward
  against the float32 CPU oracle and the float64 one (< 1e-4 absolute, the north-star tolerance for
  embeddings); every parameter gradient against the float64 oracle evaluated on the same ReLU /
  max-pool decisions (relative L2 < 1e-4), and against the free-running float64 oracle no farther
  than the reference's own float32 CPU arithmetic is (~1e-3 at these sizes: a few hundred of the
  1.3e8 gate decisions differ between any float32 forward pass and the float64 one).
* cfg-1 (2-D 256^2 synthetic zarr, num_fmaps=16, one level, 50 iterations of train()): the driver's
  own batches replayed through the oracle's train step — loss trajectory and final weights.
* cfg-5: one 512^2 sample through predict() at 256 feature maps against the oracle's tiled scan.

The float64 oracle runs its convolutions as one dgemm per filter tap (oracle.unet_oracle.
gemm_convolutions; held against nn.ConvNd in tests/test_cpu_oracle_golden.py) — torch's native
float64 convolution would take minutes per crop at these sizes.
"""

import os

import numpy as np
import pytest
import torch

from cellulus_amd.models import get_model
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu

CFG2 = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64,
            downsampling_factors=[[2, 2]], num_spatial_dims=2)
CFG4 = dict(in_channels=1, out_channels=3, num_fmaps=64, fmap_inc_factor=3, features_in_last_layer=64,
            downsampling_factors=[[2, 2, 2]], num_spatial_dims=3)


def _kaiming(model):
    for _n, layer in model.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")     # train.py:65-68


def _blobs(crop, seed):
    """the benchmark's synthetic crop (bench.py::synthetic_raw)"""
    rs = np.random.RandomState(seed)
    nd = len(crop)
    grids = np.meshgrid(*[np.arange(c, dtype=np.float32) for c in crop], indexing="ij")
    img = np.zeros(crop, dtype=np.float32)
    for c in np.stack(np.meshgrid(*[np.arange(24, c, 48) for c in crop], indexing="ij"), -1).reshape(-1, nd):
        c = c + rs.randint(-6, 7, size=nd)
        img += np.exp(-sum((g - ci) ** 2 for g, ci in zip(grids, c)) / (2 * 6.0 ** 2))
    img += rs.normal(0, 0.02, size=crop).astype(np.float32)
    return torch.from_numpy(np.clip(img, 0, 1)[None, None])


def _planar(plan, name, nd):
    """a stored activation of the launch plan (pixel-major, padded channels) as (B, C, *spatial) on the CPU"""
    shape, c = plan.topo.shapes[name]
    t = plan.buf[name].view((plan.B,) + tuple(shape) + (-1,))[..., :c].permute(0, 4, 1, 2, 3).contiguous().cpu()
    return t[:, :, 0] if nd == 2 else t


@pytest.mark.parametrize("name,cfg,crop,head_scale", [
    ("cfg2", CFG2, (256, 256), 1.0), ("cfg4", CFG4, (64, 64, 64), 1.0), ("cfg2-trained-scale", CFG2, (256, 256), 0.0)])
def test_full_size_forward_and_gradients_match_the_oracle(name, cfg, crop, head_scale, device, monkeypatch):
    """Forward: < 1e-4 against the float32 AND the float64 oracle.  head_scale 0 = "trained scale": a Kaiming network
    emits offsets of range ~0.3, a trained one offsets of O(object_size / 2) pixels — the head's last layer is scaled
    so that the output range is 15 (the offsets of object_size 30), where the same RELATIVE error is 50x the absolute one.
    There the bars are the north-star tolerance itself, ABSOLUTE — |HIP - float64| < 1e-4, for the training plan and for
    the inference plan (fused Winograd forms) — and the reference's own arithmetic: |HIP - float64| <= 1.1 x |float32 CPU
    oracle - float64| (measured in round 5: 7.3e-5 against 1.02e-4; until the GEMM kernel restarted its accumulators every
    64 products it was 1.56e-4 — the float32 CPU path blocks its contractions, one accumulator over K = 256 ... 9216
    does not).  Gradients: every parameter
    within 1e-4 (relative L2) of the float64 oracle run with the HIP forward pass's ReLU gates and
    pooling winners (oracle.unet_oracle.forced_decisions explains why: ~2e-6 of the 1.3e8 gate
    decisions differ between any float32 forward pass and the float64 one, which alone moves
    gradients by ~1e-3 — measured below for the reference's own float32 CPU arithmetic too); and
    against the free-running float64 oracle no farther than the float32 CPU oracle is."""
    import torch.nn.functional as F

    nd = len(crop)
    torch.manual_seed(0)
    oracle = O.OracleUNetModel(**cfg)
    _kaiming(oracle)
    model = get_model(**cfg)
    model.load_state_dict(oracle.state_dict(), strict=True)
    model = model.to(device)
    raw = _blobs(crop, seed=3)
    if head_scale == 0.0:
        with torch.no_grad():
            last = oracle.head[2]
            head_scale = 15.0 / oracle(raw).abs().max().item()
            last.weight.mul_(head_scale)
            last.bias.mul_(head_scale)
        model = get_model(**cfg)
        model.load_state_dict(oracle.state_dict(), strict=True)
        model = model.to(device)

    # the default plan at this size is the one the benchmark runs
    got = model(raw.to(device))
    plan = next(iter(model._plans.values()))
    assert sum(1 for a in plan.algo.values() if a["fwd"] == 2) >= 3, "Winograd F(4x4) expected on the wide layers"
    assert plan.subpixel, "the sub-pixel form of the upsample convolution is expected here"
    torch.manual_seed(2)
    dout = torch.randn(got.shape)
    got.backward(dout.to(device))
    out = got.detach().cpu()
    hip_grads = [p.grad.detach().cpu().double() for p in model.parameters()]

    # ---- forward: float32 oracle (the reference's own arithmetic) and float64 oracle
    oracle(raw).backward(dout)
    ref32 = oracle(raw).detach()
    cpu32_grads = [p.grad.double() for p in oracle.parameters()]
    o64 = O.OracleUNetModel(**cfg).double()
    o64.load_state_dict({k: v.double() for k, v in oracle.state_dict().items()})
    acts64 = []                       # post-ReLU activations of the free-running float64 pass, in call order
    hooks = [m.register_forward_hook(lambda _m, _i, o: acts64.append(o.detach() > 0))
             for m in o64.modules() if isinstance(m, torch.nn.ReLU)]
    with O.gemm_convolutions(o64):
        ref64 = o64(raw.double())
        ref64.backward(dout.double())
    for h in hooks:
        h.remove()
    free64 = [p.grad.clone() for p in o64.parameters()]
    ref64 = ref64.detach()
    assert out.shape == ref32.shape == tuple(ref64.shape)
    scale = ref64.abs().max().item()
    err32 = (out - ref32).abs().max().item()
    err64 = (out.double() - ref64).abs().max().item()
    cpu_err = (ref32.double() - ref64).abs().max().item()
    print(f"{name}: output range {scale:.3f}, |hip - f32 oracle| {err32:.2e}, |hip - f64 oracle| {err64:.2e}, "
          f"|f32 oracle - f64 oracle| {cpu_err:.2e}")
    if head_scale == 1.0:
        assert err32 < 1e-4 and err64 < 1e-4
    else:
        print(f"{name}: head scaled by {head_scale:.1f}: absolute error {err64:.2e} at output range {scale:.1f} "
              f"(relative {err64 / scale:.2e}); the float32 CPU oracle's: {cpu_err:.2e} (relative {cpu_err / scale:.2e})")
        assert 14.0 < scale < 16.5
        assert err64 < 1e-4, err64       # the north-star tolerance, absolute, at a trained network's output range
        assert err64 <= 1.1 * cpu_err, (err64, cpu_err)
        # ... and the distance between the two float32 implementations: each is ~1e-4 from the truth here, so 1e-4 from EACH
        # OTHER cannot be promised (DESIGN.md 4 reads the north-star's 1e-4 against float64 for that reason); what can be: no
        # farther apart than their two distances from float64 together (1.75e-4 in round 5: 1.27e-4 / 1.65e-4 measured)
        assert err32 <= err64 + cpu_err and err32 < 2e-4, (err32, err64, cpu_err)
        with torch.no_grad():            # the inference plan of the same network (fused Winograd forms, keeps nothing)
            out_inf = model(raw.to(device)).cpu()
        inf_plan = [p for p in model._plans.values() if not p.keep]
        assert inf_plan, "a forward without gradients was expected to build the inference plan"
        err_inf = (out_inf.double() - ref64).abs().max().item()
        print(f"{name}: inference plan |hip - f64 oracle| {err_inf:.2e} "
              f"(layers in the fused Winograd form: {sum(1 for a in inf_plan[0].algo.values() if a['fwd'] == 3)})")
        assert err_inf < 1e-4, err_inf
        err_inf32 = (out_inf - ref32).abs().max().item()
        print(f"{name}: inference plan |hip - f32 oracle| {err_inf32:.2e} (bound {err_inf + cpu_err:.2e})")
        assert err_inf32 <= err_inf + cpu_err and err_inf32 < 2e-4, (err_inf32, err_inf, cpu_err)
    out_bar = 2e-4 if head_scale != 1.0 else 1e-4

    # ---- gradients: float64 arithmetic on the HIP forward pass's discrete decisions
    relu_layers = [layer for layer in plan.topo.convs if layer.relu]
    masks = [_planar(plan, layer.out, nd) > 0 for layer in relu_layers]
    pool = F.max_pool2d if nd == 2 else F.max_pool3d
    winners = [pool(_planar(plan, p.src, nd), p.factor[3 - nd:], stride=p.factor[3 - nd:], return_indices=True)[1]
               for p in plan.topo.pools]
    for p in o64.parameters():
        p.grad = None
    with O.gemm_convolutions(o64), O.forced_decisions(o64, masks, winners):
        forced = o64(raw.double())
        forced.backward(dout.double())
    assert (out.double() - forced.detach()).abs().max().item() < out_bar
    worst = worst_free = worst_cpu = 0.0
    for (n, po), g, g32, gf in zip(o64.named_parameters(), hip_grads, cpu32_grads, free64):
        l2 = ((g - po.grad).norm() / (po.grad.norm() + 1e-30)).item()
        worst = max(worst, l2)
        assert l2 < 1e-4, f"{name}: grad of {n}: rel L2 err {l2} against float64 on the same decisions"
        worst_free = max(worst_free, ((g - gf).norm() / gf.norm()).item())
        worst_cpu = max(worst_cpu, ((g32 - gf).norm() / gf.norm()).item())
    flipped = sum(int((m != a).sum()) for m, a in zip(masks, acts64))
    total = sum(m.numel() for m in masks)
    print(f"{name}: worst relative L2 error of a parameter gradient: {worst:.2e} on the same decisions; "
          f"free-running float64: HIP {worst_free:.2e}, float32 CPU oracle {worst_cpu:.2e}; "
          f"{flipped} of {total} ReLU gates differ between the HIP and the float64 forward pass")
    # the free-running distance is set by WHICH handful of gates differ (a gate deep in the low-resolution
    # path carries the gradient of hundreds of output pixels), not by the arithmetic.  The yardstick is the
    # reference's own float32 CPU arithmetic on the same crop and weights: the HIP path may sit a small
    # multiple of ITS distance from the float64 result (measured: 2.9x and 3.9x), not a constant
    assert worst_cpu < 1e-2, worst_cpu
    assert worst_free < 8 * max(worst_cpu, 2.5e-4), (worst_free, worst_cpu)


def test_cfg1_train_driver_50_iterations_replayed_through_the_oracle(device, tmp_path, monkeypatch):
    """BASELINE configs[0]: 2-D 1-channel 256x256 synthetic zarr, num_fmaps=16, 1-level U-Net, 50
    iterations of train() (the reference's CPU plumbing case; here the same toml on cuda:0 —
    cellulus_amd has no CPU path).  Every batch the driver hands to train_iteration is replayed
    through the oracle's train step (cellulus/train.py:160-180 restated on the CPU) from the same
    initial weights: loss trajectory within 1e-4 relative, final weights equal."""
    import tomli

    import cellulus_amd.train as T
    from cellulus_amd.configs import ExperimentConfig
    from cellulus_amd.utils import zarr_io

    monkeypatch.chdir(tmp_path)
    container = str(tmp_path / "data.zarr")
    f = zarr_io.open(container)
    f["train/raw"] = np.concatenate([_blobs((256, 256), seed=s).numpy() for s in range(4)], axis=0)
    f["train/raw"].attrs["axis_names"] = ["s", "c", "y", "x"]
    toml = f"""
normalization_factor = 1.0
[model_config]
num_fmaps = 16
fmap_inc_factor = 3
downsampling_factors = [[2, 2]]

[train_config]
crop_size = [256, 256]
batch_size = 2
max_iterations = 50
num_workers = 0
elastic_deform = false
save_model_every = 1000
save_best_model_every = 1000
save_snapshot_every = 1000
device = "cuda:0"

[train_config.train_data_config]
container_path = "{container}"
dataset_name = "train/raw"
"""
    cfg = ExperimentConfig(**tomli.loads(toml))
    mc = cfg.model_config
    ocfg = dict(in_channels=1, out_channels=2, num_fmaps=mc.num_fmaps, fmap_inc_factor=mc.fmap_inc_factor,
                features_in_last_layer=mc.features_in_last_layer,
                downsampling_factors=[tuple(x) for x in mc.downsampling_factors], num_spatial_dims=2)
    state = {}
    real_step = T.train_iteration

    def spy(batch, model, criterion, optimizer, device):
        if "oracle" not in state:                        # initial weights = the driver's, before step 0
            oracle = O.OracleUNetModel(**ocfg)
            oracle.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, strict=True)
            state["oracle"] = oracle
            state["opt"] = torch.optim.Adam(oracle.parameters(), lr=cfg.train_config.initial_learning_rate,
                                            weight_decay=0.01)                    # train.py:80-82
            state["hip"], state["ref"] = [], []
            state["model"] = model
            state["initial"] = [p.detach().clone() for p in oracle.parameters()]
        cpu_batch = tuple(t.detach().cpu().clone() for t in batch)
        assert cpu_batch[0].shape == (2, 1, 256, 256) and cpu_batch[1].shape == (2, 150040, 2)
        out = real_step(batch, model, criterion, optimizer, device)
        l_ref, _o, _off = O.train_step(state["oracle"], state["opt"], cpu_batch[0], cpu_batch[1], cpu_batch[2],
                                       criterion.temperature, criterion.regularization_weight)
        state["hip"].append(out[0])
        state["ref"].append(l_ref)
        return out

    monkeypatch.setattr(T, "train_iteration", spy)
    torch.manual_seed(0)
    np.random.seed(0)
    import random

    random.seed(0)
    T.train(cfg)
    hip, ref = np.array(state["hip"]), np.array(state["ref"])
    assert len(hip) == 50 and np.isfinite(hip).all()
    rel = np.abs(hip - ref) / np.abs(ref)
    print(f"cfg-1: loss {ref[0]:.3f} -> {ref[-1]:.3f}; max relative loss difference over 50 iterations {rel.max():.2e}")
    assert rel.max() < 1e-4, (rel.argmax(), rel.max())
    assert ref[-1] < ref[0]
    lines = open("loss.csv").read().strip().split("\n")
    assert len(lines) == 51
    # final weights: 50 Adam steps of at most lr = 4e-5 each.  Adam divides by sqrt(v): an element whose
    # gradient is rounding noise (the float atomics of the weight-gradient kernels make its last bits differ
    # from run to run) moves by ~lr per step in a direction the noise decides, so two correct runs can end
    # up to 2 * 50 * lr = 4e-3 apart in such an element.  The bars: no element farther than a quarter of
    # that, and the difference small against what the 50 steps moved (|update| ~ 50 lr sqrt(n)).
    lr = cfg.train_config.initial_learning_rate
    worst, ratio, worst_name = 0.0, 0.0, ""
    for (n, po), (n2, pm), p0 in zip(state["oracle"].named_parameters(), state["model"].named_parameters(),
                                     state["initial"]):
        assert n == n2
        d = (pm.detach().cpu() - po.detach()).abs()
        update = (po.detach() - p0).norm().item()
        if d.max().item() > worst:
            worst, worst_name = d.max().item(), n
        ratio = max(ratio, d.norm().item() / (update + 1e-12))
    print(f"cfg-1: largest weight difference after 50 iterations {worst:.2e} ({worst_name}; 50 lr = {50 * lr:.1e}); "
          f"largest |difference| / |50-step update| of a parameter {ratio:.2e}")
    assert worst < 0.25 * 2 * 50 * lr, (worst_name, worst)
    assert ratio < 0.1, ratio


# (the benchmark's inference tile — one 528^2 tile, 32 noisy forwards — against the oracle's scan: since round 5 on a
#  TRAINED network of the benchmark width, tests/test_gpu_trained_e2e.py::test_trained_benchmark_width_network_at_the_benchmark_tile)


def test_cfg5_infer_512_at_256_feature_maps_end_to_end_against_the_oracle_pipeline(device, tmp_path, monkeypatch):
    """BASELINE configs[4] in ONE call: ``infer()`` (fused predict -> detect -> segment) on a 512^2 sample
    with the cfg-2 network (256 feature maps, one 528^2 tile, num_infer_iterations=2), seeded as a user
    would seed the reference, against the oracle's pipeline seeded the same way:

    * embeddings within 1e-4 of the oracle's scan (which includes the reference's dry run);
    * ``detection`` / ``segmentation`` BIT FOR BIT what the oracle's detect + segment make of the
      embeddings the run wrote (same np.random sub-sampling stream);
    * and the whole chain against the oracle chain on the ORACLE's embeddings (CPU float32 network ->
      mean-shift -> grow/shrink -> size filter).  The two embedding tensors differ by ~1e-6, and the
      embeddings of a RANDOM-weight network are not object-centred: their mean-shift modes are fragile, so
      the chain amplifies a rounding-sized difference into other instances for part of the image (measured:
      12 % of the pixels).  The yardstick is therefore the oracle chain's OWN conditioning: the same oracle
      chain on the oracle's embeddings perturbed by uniform noise of the measured |HIP - oracle| amplitude.
      The HIP run's disagreement with the oracle must stay within 3x (+ 1 point) of the oracle's
      disagreement with its perturbed self; both figures are printed."""
    from cellulus_amd.configs import ExperimentConfig
    from cellulus_amd.infer import infer
    from cellulus_amd.utils import zarr_io
    from oracle import infer_oracle as IO

    monkeypatch.chdir(tmp_path)
    container = str(tmp_path / "data.zarr")
    raw = _blobs((512, 512), seed=5).numpy()
    f = zarr_io.open(container)
    f["test/raw"] = raw
    f["test/raw"].attrs["axis_names"] = ["s", "c", "y", "x"]
    torch.manual_seed(0)
    oracle = O.OracleUNetModel(**CFG2)
    _kaiming(oracle)
    os.makedirs("models")
    torch.save({"model_state_dict": oracle.state_dict()}, "models/best_loss.pth")
    n_it, p, rp = 2, 0.05, 0.1
    cfg = ExperimentConfig(
        model_config=dict(num_fmaps=256, fmap_inc_factor=3, downsampling_factors=[[2, 2]],
                          checkpoint="models/best_loss.pth"),
        object_size=30, normalization_factor=1.0,
        inference_config=dict(
            dataset_config=dict(container_path=container, dataset_name="test/raw"),
            prediction_dataset_config=dict(container_path=container, dataset_name="embeddings"),
            detection_dataset_config=dict(container_path=container, dataset_name="detection",
                                          secondary_dataset_name="embeddings"),
            segmentation_dataset_config=dict(container_path=container, dataset_name="segmentation",
                                             secondary_dataset_name="detection"),
            crop_size=[528, 528], num_infer_iterations=n_it, p_salt_pepper=p, reduction_probability=rp,
            device="cuda:0"))
    torch.manual_seed(42)
    np.random.seed(42)
    infer(cfg)
    bw, min_size = cfg.inference_config.bandwidth, cfg.inference_config.min_size
    assert bw == 15.0 and min_size == 70                      # infer.py:28-39 defaults for object_size 30
    g = zarr_io.open(container, "r")
    emb, det, seg = g["embeddings"][...], g["detection"][...], g["segmentation"][...]
    assert emb.shape == (1, 3, 512, 512) and det.dtype == seg.dtype == np.uint16

    # ---- the oracle chain, seeded the same way (infer() builds its model before the checkpoint is loaded)
    torch.manual_seed(42)
    np.random.seed(42)
    O.OracleUNetModel(**CFG2)
    ref_emb = O.predict_scan(oracle, raw, [528, 528], p, n_it, 1.0, literal_dry_run=False)
    err = np.abs(emb - ref_emb).max()
    assert err < 1e-4, err
    _mask, _cen, ref_labels = IO.detect_sample(ref_emb[0], bw, 1, min_size, rp)
    ref_seg = IO.segment_sample(ref_labels[0].astype(np.uint16).astype(np.int32), None, "cell", 3, 6, min_size)

    # ---- bit for bit on the embeddings the run wrote
    np.random.seed(42)
    _mask, _cen, own_labels = IO.detect_sample(emb[0], bw, 1, min_size, rp)
    np.testing.assert_array_equal(IO.label(det[0, 0]), IO.label(own_labels[0]))
    own_seg = IO.segment_sample(det[0, 0].astype(np.int32), None, "cell", 3, 6, min_size)
    np.testing.assert_array_equal(seg[0, 0], own_seg)

    # ---- the whole chain against the whole oracle chain
    _ids_p, _ids_g, joint = IO.joint_histogram(seg[0, 0].astype(np.int64), ref_seg.astype(np.int64))
    agree = joint.max(axis=1).sum() / seg[0, 0].size          # pixels in the best-matching instance pairs
    identical = bool(np.array_equal(seg[0, 0], ref_seg))
    print(f"cfg-5 infer(): |embeddings - oracle| {err:.2e}; {int(seg.max())} instances (oracle chain "
          f"{int(ref_seg.max())}); segmentation identical to the all-oracle chain: {identical}; "
          f"pixels agreeing in their instance: {agree:.6f}")
    # the oracle chain's own sensitivity to a perturbation of that size
    rng = np.random.default_rng(0)
    np.random.seed(42)
    shaken = ref_emb[0] + rng.uniform(-err, err, size=ref_emb[0].shape)
    _m, _c, shaken_labels = IO.detect_sample(shaken, bw, 1, min_size, rp)
    shaken_seg = IO.segment_sample(shaken_labels[0].astype(np.uint16).astype(np.int32), None, "cell", 3, 6, min_size)
    _a, _b, joint_self = IO.joint_histogram(shaken_seg.astype(np.int64), ref_seg.astype(np.int64))
    agree_self = joint_self.max(axis=1).sum() / ref_seg.size
    print(f"cfg-5 infer(): the oracle chain against itself under a +-{err:.1e} perturbation of its embeddings: "
          f"{agree_self:.6f}")
    # (one draw each of a high-variance quantity — a moved mode changes a whole instance: measured 0.124
    # for the HIP run against 0.070 for the perturbed oracle)
    assert (1 - agree) < 3 * (1 - agree_self) + 0.01, (agree, agree_self)


def test_benchmark_step_batch_8_on_two_streams_matches_the_oracle_step(device):
    """The step `bench.py` times — cfg-2, batch 8, run as two half batches on two HIP streams (plan.DualPlan, the default
    at this size) — against the oracle's train step (cellulus/train.py:160-180 restated on the CPU) on the same batch
    and weights: the loss (a sum over the 1.2 M pairs of the batch) to 1e-6 relative, every parameter's gradient close to
    the float32 CPU path's (free-running: a handful of the 1e9 ReLU decisions of a batch differ between any two float32
    implementations, which moves gradients by ~1e-3 relative — the bar is 1e-2, far below what a lost half batch, a
    missing bucket or a double count would show), and the updated parameters within one Adam step's reach."""
    import bench
    from cellulus_amd.models.plan import DualPlan
    from cellulus_amd.train import train_iteration

    model, crit, opt, batch = bench.build_step_inputs("train2d", 0, device, broadcast=False)
    oracle = O.OracleUNetModel(**CFG2)
    oracle.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, strict=True)
    before = [p.detach().cpu().clone() for p in model.parameters()]
    ref_opt = torch.optim.Adam(oracle.parameters(), lr=4e-5, weight_decay=0.01)
    threads_before = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    raw, anchor, reference = (t.cpu() for t in batch)
    l_ref, o_ref, _off = O.train_step(oracle, ref_opt, raw, anchor, reference, 10.0, 1e-5)
    torch.set_num_threads(threads_before)
    loss, oce, _offsets = train_iteration(batch, model, crit, opt, device)
    assert isinstance(next(iter(model._plans.values())), DualPlan), "the benchmark step runs on two streams"
    assert abs(loss - l_ref) <= 1e-6 * abs(l_ref) and abs(oce - o_ref) <= 1e-6 * abs(o_ref), (loss, l_ref)
    worst, worst_name = 0.0, ""
    flat = model._flat_grad.detach().cpu()
    off = 0
    for (n, po), p0, pm in zip(oracle.named_parameters(), before, model.parameters()):
        g = flat[off:off + po.numel()].view(po.shape)
        off += po.numel()
        rel = ((g - po.grad).norm() / (po.grad.norm() + 1e-30)).item()
        if rel > worst:
            worst, worst_name = rel, n
        # one Adam step moves an element by at most lr (+ weight decay): both ends started from the same weights
        assert (pm.detach().cpu() - po.detach()).abs().max().item() <= 2.1 * 4e-5, n
        assert not torch.equal(pm.detach().cpu(), p0), n
    print(f"cfg-2 batch 8, two streams: loss {loss:.3f} (oracle {l_ref:.3f}); worst relative L2 distance of a parameter "
          f"gradient from the float32 CPU oracle's: {worst:.2e} ({worst_name})")
    assert worst < 1e-2, (worst_name, worst)
