"""Inference leg of bench.py: BASELINE.json configs[4] — 2-D 512x512 inference on one
MI355X: embeddings (2 x 16 salt/pepper-noised U-Net forwards + mean/std), mean-shift
detection and cell post-processing (grow/shrink + connected components + size filter).

The embedding stage runs the benchmark network (num_fmaps=256, inc 3) with random
weights on one 528^2 reflect-padded tile (output 512^2).  Clustering quality depends
on trained weights, so the detect/segment stages are timed on the synthetic
disc embeddings of SURVEY.md §8d (512^2, ~100 objects, bandwidth 15,
reduction_probability 0.1) — the same input the CPU oracle is timed on.
Mpixels/s = 512*512 / (embed + detect + segment) per image.
"""

import time

import numpy as np
import torch


def _sync_time(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def synthetic_embeddings(shape, spacing=48, radius=12, noise=0.3, seed=1):
    """The benchmark's detection input: disc/ball objects on a jittered grid,
    embedding = (centre - pixel) + N(0, noise) in (x, y[, z]) channel order, std = 0.01 inside / 1.0
    outside.  Returns (mean (1, ND, *shape) f64, std (*shape) f64)."""
    rs = np.random.RandomState(seed)
    nd = len(shape)
    grids = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij")
    mean = np.zeros((nd,) + tuple(shape))
    std = np.ones(shape)
    centres = np.stack(np.meshgrid(*[np.arange(spacing // 2, s, spacing) for s in shape], indexing="ij"),
                       -1).reshape(-1, nd)
    for c in centres:
        c = c + rs.randint(-6, 7, size=nd)
        d2 = sum((g - ci) ** 2 for g, ci in zip(grids, c))
        inside = d2 <= radius * radius
        std[inside] = 0.01
        for k in range(nd):            # channel 0 = x offset (last axis)
            ax = nd - 1 - k
            mean[k][inside] = (c[ax] - grids[ax])[inside]
    mean += rs.normal(0, noise, size=mean.shape) * (std < 0.5)
    return mean[np.newaxis].copy(), std


def infer_bench(device, reps=2, with_cpu=True):
    from cellulus_amd.models import get_model
    from cellulus_amd.segment import grow_shrink_on_device
    from cellulus_amd.utils.mean_shift import mean_shift_on_device
    from cellulus_amd.utils.misc import label_on_device
    from cellulus_amd.utils.otsu import threshold_otsu

    size, crop, n_it = 512, 528, 16
    cfg = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64,
               downsampling_factors=[[2, 2]], num_spatial_dims=2)
    torch.manual_seed(0)
    model = get_model(**cfg).to(device)
    for _n, layer in model.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    model.eval()
    model.set_infer(p_salt_pepper=0.01, num_infer_iterations=n_it, device=device)
    model.max_infer_batch = 8
    rng = np.random.default_rng(0)
    raw = torch.from_numpy(np.pad(rng.random((1, 1, size, size), dtype=np.float32),
                                  [(0, 0), (0, 0), (8, 8), (8, 8)], mode="reflect")).to(device)
    noise = torch.rand(1, 2 * n_it, 1, crop, crop, device=device)
    model.infer_on_device(raw, noise=noise)                       # warm-up (plan + packing)
    t_embed, emb = _sync_time(lambda: model.infer_on_device(raw, noise=noise), reps)
    assert tuple(emb.shape) == (1, 3, size, size)

    mean, std = synthetic_embeddings((size, size), spacing=48, radius=12, noise=0.3, seed=1)
    mean_d = torch.from_numpy(mean[0]).to(device)
    std_d = torch.from_numpy(std).to(device)

    def detect_once():
        np.random.seed(1)
        thr = threshold_otsu(std_d)
        labels, centers = mean_shift_on_device(mean_d.clone(), std_d, 15.0, 0.1, thr, None)
        return labels, centers

    detect_once()
    t_detect, (labels, centers) = _sync_time(detect_once, max(reps, 3))

    def segment_once():
        seg = labels.clone()
        grow_shrink_on_device(seg, 3, 6)
        out, n = label_on_device(seg, 70)
        return out, n

    segment_once()
    t_segment, (seg, ncomp) = _sync_time(segment_once, max(reps, 3))

    # mean-shift at full density (reduction_probability 1.0: every foreground pixel is a seed —
    # the reference's 48.8 s case); reported as pair evaluations per second
    def detect_full():
        labels_f, centers_f = mean_shift_on_device(mean_d.clone(), std_d, 15.0, 1.0, 0.5, None)
        return labels_f, centers_f

    detect_full()
    t_full, (labels_full, centers_full) = _sync_time(detect_full, 2)
    nfg = int((std < 0.5).sum())

    total = t_embed + t_detect + t_segment
    from bench import conv_flops
    plan = next(iter(model._plans.values()))
    fwd_flops, _, _ = conv_flops(plan.topo, 1)
    out = {
        "metric": "infer Mpixels/s (embed + mean-shift detect + segment), 2D 512x512, 1 GPU",
        "value": round(size * size / total / 1e6, 4),
        "unit": "Mpixels/s",
        "stage_ms": {"embed": round(t_embed * 1e3, 2), "detect": round(t_detect * 1e3, 3),
                     "segment": round(t_segment * 1e3, 3)},
        "embed_tflops": round(2 * n_it * fwd_flops / t_embed / 1e12, 2),
        "meanshift_rp1": {"ms": round(t_full * 1e3, 2), "seeds": nfg, "clusters": int(len(centers_full))},
        "objects": int(ncomp.item()),
        "clusters": int(len(centers)),
    }
    if with_cpu:
        # CPU baseline leg: the oracle (sklearn's algorithm restated in C, 1 thread, as the reference
        # runs it, + C connected-component labelling) on the same input — the only use of oracle/ here
        from oracle import infer_oracle as IO

        np.random.seed(1)
        t0 = time.perf_counter()
        thr = IO.threshold_otsu(std)
        ref = IO.mean_shift_segmentation(mean.copy(), std, 15.0, 70, 0.1, thr, None)
        t_ms = time.perf_counter() - t0
        t0 = time.perf_counter()
        ref_seg = IO.size_filter(IO.grow_shrink(ref, 3, 6), 70)
        t_sf = time.perf_counter() - t0
        same = bool(np.array_equal(ref_seg, seg.cpu().numpy()))
        out["cpu_oracle"] = {"detect_ms": round(t_ms * 1e3, 1), "segment_ms": round(t_sf * 1e3, 1),
                             "cores": 1, "labels_identical": same}
    return out
