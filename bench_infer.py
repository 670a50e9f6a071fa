"""Inference leg of bench.py: BASELINE.json configs[4] — 2-D 512x512 inference on one
MI355X: embeddings (2 x 16 salt/pepper-noised U-Net forwards + mean/std), mean-shift
detection and cell post-processing (grow/shrink + connected components + size filter).

Three views, all in the `infer` object of the bench line:

* kernel-level stage times on one 528^2 reflect-padded tile (output 512^2) of the benchmark
  network (num_fmaps=256, inc 3) with random weights and device-resident noise; clustering
  quality depends on trained weights, so detect / segment are timed on the synthetic disc
  embeddings of SURVEY.md §8d (512^2, ~100 objects, bandwidth 15, reduction_probability 0.1) —
  the input the CPU leg is timed on.  value = 512*512 / (embed + detect + segment).
* `e2e`: the real `cellulus_amd.infer.infer()` (fused predict -> detect -> segment, zarr in, five
  zarr datasets out, CPU-generator noise drawn one tile ahead) over a synthetic S x 512^2 zarr.
* `roofline` (the embedding GEMM kernel, f32 MFMA) and `streaming` (every HBM-bound kernel of
  detect / segment at batch scale: one launch over 4096^2 pixels = 64 samples of 512^2;
  algorithmic bytes / HIP-event time against the 8 TB/s HBM3E peak).
"""

import os
import tempfile
import time

import numpy as np
import torch

HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md
F32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2516.6     # dense bf16 MFMA; the split precision spends six bf16 products per float32 product
SP_KERNELS = ("gemm_sp_kernel<0>", "gemm_sp_kernel<1>", "gemm_sp2_kernel")   # forward / data gradient, weight gradient


def _sync_time(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def _event_time(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


PROF_KIND = dict(ms_prepare=7, ms_assign=8, cc=9, grow_shrink=10, minmax=11, histogram=12, noise_stats=13)   # clx.h


def _kernel_time(fn, kind, reps=5):
    """(seconds per call between two HIP events on the stream around `reps` calls, seconds per call of the KERNELS of
    profile kind `kind` alone: libclx stamps an event pair on every such launch, the kernel's own start and end — the
    duration `rocprofv3 --kernel-trace --stats` reports).  The first includes what lies between the launches: the
    host's launch rate (Python + ctypes: ~20 us per call) and the dependent-launch gaps of multi-kernel operations."""
    import ctypes

    from cellulus_amd import _clx

    fn()
    torch.cuda.synchronize()
    lib = _clx.load()
    _clx.call("clx_profile_enable", 2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    n_l, ms_l, fl_l = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    lib.clx_profile_read(kind, ctypes.byref(n_l), ctypes.byref(ms_l), ctypes.byref(fl_l))
    _clx.call("clx_profile_enable", 0)
    return e0.elapsed_time(e1) / reps * 1e-3, ms_l.value / reps * 1e-3, n_l.value / reps


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


def streaming_rooflines(device, size=4096, only_mean_shift=False):
    """Every HBM-bound kernel of detect / segment on one size x size image (batch scale: size^2 /
    512^2 samples per launch).  bytes = the algorithm's compulsory traffic (stated per kernel)."""
    from cellulus_amd import _clx
    from cellulus_amd.segment import grow_shrink_on_device
    from cellulus_amd.utils import mean_shift as MS
    from cellulus_amd.utils.misc import label_on_device

    import ctypes

    lib = _clx.load()
    st = _clx.stream_ptr(device)
    Y = X = size
    npix = Y * X
    rng = np.random.default_rng(0)
    out = {}

    def row(name, times, nbytes, what, **extra):
        call_s, kernel_s, launches = times
        out[name] = dict(ms=round(kernel_s * 1e3, 4), GBs=round(nbytes / kernel_s / 1e9, 1),
                         frac=round(nbytes / kernel_s / HBM_PEAK, 4), bytes=what, kernels_per_call=round(launches, 2),
                         call_ms=round(call_s * 1e3, 4), **extra)

    # --- the benchmark's detection input, tiled up to size^2 (objects every 48 px, ~17 % foreground)
    mean, std = synthetic_embeddings((512, 512), spacing=48, radius=12, noise=0.3, seed=1)
    reps = size // 512
    emb0 = torch.from_numpy(np.tile(mean[0], (1, reps, reps))).to(device)
    sd = torch.from_numpy(np.tile(std, (reps, reps))).to(device)
    ws = torch.empty(int(lib.clx_ms_prepare_workspace(npix)), dtype=torch.uint8, device=device)   # plain scratch
    pts = torch.empty((npix, 2), dtype=torch.float64, device=device)
    idx = torch.empty(npix, dtype=torch.int32, device=device)
    nfg = torch.zeros(1, dtype=torch.int32, device=device)
    emb = emb0.clone()
    # (as the product calls it since round 5: without the raster index — clx_ms_assign_dense does not read one)
    t = _kernel_time(lambda: _clx.call("clx_ms_prepare", _clx.ptr(emb), _clx.ptr(sd), 0.5, 2, 1, Y, X, _clx.ptr(pts),
                                       None, _clx.ptr(nfg), _clx.ptr(ws), st), PROF_KIND["ms_prepare"])
    n_fg = int(nfg.item())
    row("ms_prepare", t, npix * (3 * 8 + 2 * 8) + n_fg * (2 * 8),
        "per pixel 24 B read + 16 B written, per foreground pixel 16 B more (the point; no raster index)", nfg=n_fg)
    # the fused path's form: the network's float32 planes in, widened in registers, nothing written per background pixel
    emb32, sd32 = emb0.float(), sd.float()
    t = _kernel_time(lambda: _clx.call("clx_ms_prepare_f32", _clx.ptr(emb32), _clx.ptr(sd32), 0.5, 2, 1, Y, X, _clx.ptr(pts),
                                       None, _clx.ptr(nfg), _clx.ptr(ws), st), PROF_KIND["ms_prepare"])
    assert int(nfg.item()) == n_fg
    row("ms_prepare_f32", t, npix * 4 + n_fg * (2 * 4 + 2 * 8),
        "per pixel 4 B read (std), per foreground pixel 8 B read + 16 B written (infer()'s fused hand-over)", nfg=n_fg)
    del emb32
    # centres: one per object (their true centres), assignment of all foreground pixels
    emb = emb0.clone()
    _clx.call("clx_ms_prepare", _clx.ptr(emb), _clx.ptr(sd), 0.5, 2, 1, Y, X, _clx.ptr(pts), _clx.ptr(idx),
              _clx.ptr(nfg), _clx.ptr(ws), st)
    labels = torch.zeros(npix, dtype=torch.int32, device=device)
    # the cluster centres mean-shift finds on one 512^2 tile, repeated for every tile of the image
    np.random.seed(1)
    _lab, base_centers = MS.mean_shift_on_device(torch.from_numpy(mean[0]).to(device), torch.from_numpy(std).to(device),
                                                 15.0, 0.1, 0.5, None)
    shifts = np.stack(np.meshgrid(np.arange(reps) * 512.0, np.arange(reps) * 512.0, indexing="ij"), -1).reshape(-1, 2)
    centers = (base_centers[None, :, :] + shifts[:, None, ::-1]).reshape(-1, 2)       # (x, y) columns
    order, cstart, corigin, (gx, gy, gz) = MS._center_grid(centers, 15.0)
    order_d, cstart_d = torch.from_numpy(order).to(device), torch.from_numpy(cstart).to(device)
    corigin_c = (ctypes.c_double * 2)(*corigin.tolist())
    cc_sorted = torch.from_numpy(np.ascontiguousarray(centers[order])).to(device)      # as mean_shift_on_device calls it
    t = _kernel_time(lambda: _clx.call("clx_ms_assign_cells", _clx.ptr(pts), _clx.ptr(idx), n_fg, _clx.ptr(cc_sorted),
                                       len(centers), 2, _clx.ptr(order_d), _clx.ptr(cstart_d), corigin_c, 15.0,
                                       gx, gy, gz, _clx.ptr(labels), st), PROF_KIND["ms_assign"])
    row("ms_assign_scatter", t, n_fg * (16 + 4 + 4), "per foreground pixel 20 B read + 4 B label written through the raster "
        "index (rounds 1-4's form: the zero fill of the map it scatters into is NOT in this row)", centres=len(centers))
    # the form the product runs: the whole label map in one pass over the compaction's tiles (flags from `ws`)
    dense = torch.empty(npix, dtype=torch.int32, device=device)
    t = _kernel_time(lambda: _clx.call("clx_ms_assign_dense", _clx.ptr(pts), _clx.ptr(cc_sorted), len(centers), 2,
                                       _clx.ptr(order_d), _clx.ptr(cstart_d), corigin_c, 15.0, gx, gy, gz, _clx.ptr(ws), 0,
                                       1, Y, X, _clx.ptr(dense), st), PROF_KIND["ms_assign"])
    assert torch.equal(dense, labels)
    row("ms_assign", t, n_fg * 16 + npix * 4 + npix // 8, "per foreground pixel 16 B read, per pixel 4 B label written "
        "(background included: no zero fill, no raster index) + 1 flag bit read", centres=len(centers))
    del dense
    if only_mean_shift:
        return dict(pixels_per_launch=npix, samples_of_512x512_per_launch=npix // (512 * 512), kernels=out)
    seg = labels.view(Y, X).clone()
    del emb, emb0, pts, idx
    # --- grow / shrink and connected components + size filter on that label map
    t = _kernel_time(lambda: grow_shrink_on_device(seg.clone(), 3, 6), PROF_KIND["grow_shrink"], reps=3)
    row("grow_shrink", t, npix * 8, "per pixel 4 B read + 4 B written (call_ms includes the clone of the label map)")
    grown = grow_shrink_on_device(seg.clone(), 3, 6)
    t = _kernel_time(lambda: label_on_device(grown, 70), PROF_KIND["cc"], reps=3)
    row("cc_label_filter", t, npix * 8, "per pixel 4 B read + 4 B written")
    # --- Otsu: min/max + 256-bin histogram of the float64 std channel
    mm = torch.empty(_clx.MINMAX_DOUBLES, dtype=torch.float64, device=device)
    x = sd.reshape(-1)
    t = _kernel_time(lambda: _clx.call("clx_minmax_f64", _clx.ptr(x), npix, _clx.ptr(mm), st), PROF_KIND["minmax"])
    row("minmax_f64", t, npix * 8, "per pixel 8 B read")
    edges = torch.linspace(0, 1, 257, dtype=torch.float64, device=device)
    counts = torch.zeros(256, dtype=torch.int64, device=device)
    t = _kernel_time(lambda: _clx.call("clx_histogram_f64", _clx.ptr(x), npix, _clx.ptr(edges), 256, _clx.ptr(counts), st),
                     PROF_KIND["histogram"])
    row("histogram_f64", t, npix * 8, "per pixel 8 B read")
    x32 = sd32.reshape(-1)
    t = _kernel_time(lambda: _clx.call("clx_histogram_f32", _clx.ptr(x32), npix, _clx.ptr(edges), 256, _clx.ptr(counts), st),
                     PROF_KIND["histogram"])
    row("histogram_f32", t, npix * 4, "per pixel 4 B read (infer()'s fused hand-over: min / max come with the std plane)")
    # --- mean / std over the 32 noisy predictions
    T, C, n = 32, 2, 2048 * 2048
    preds = torch.randn(T, C, n, device=device)
    o = torch.empty(C + 1, n, device=device)
    t = _kernel_time(lambda: _clx.call("clx_noise_stats", _clx.ptr(preds), _clx.ptr(o), T, C, n, st),
                     PROF_KIND["noise_stats"])
    row("noise_stats", t, n * (T * C * 4 + (C + 1) * 4), "per pixel 256 B read + 12 B written")
    return dict(pixels_per_launch=npix, samples_of_512x512_per_launch=npix // (512 * 512), peak_GBs=HBM_PEAK / 1e9,
                timed="ms / GBs / frac: the operation's kernels alone (event pairs libclx stamps on each launch: a kernel's own "
                      "start and end on its stream, what rocprofv3 --kernel-trace --stats reports as its duration), summed "
                      "per call; call_ms: HIP events on the stream around the calls, i.e. including the host's launch rate "
                      "from Python and the gaps between dependent launches",
                kernels=out)


def e2e_infer(device, samples=32, size=512, rank=0, world=1):
    """The product's infer() on a synthetic zarr: S x size^2 raw images in, embeddings / detection /
    binary-segmentation / centered-embeddings / segmentation out, default inference settings
    (16 noise iterations, reduction_probability 0.1, cell post-processing).

    Several ranks (every rank calls this; `samples` is then PER RANK — weak scaling): one container, rank 0
    creates the datasets, every rank fills its block of samples (parallel.shard_range), no collective on the
    data path; the time is the slowest rank's, the rate the whole job's."""
    from cellulus_amd.configs import ExperimentConfig
    from cellulus_amd.infer import infer
    from cellulus_amd.models import get_model
    from cellulus_amd.utils import zarr_io

    from bench import _shared_tmpdir, synthetic_raw

    import contextlib
    import io
    import shutil

    mcfg = dict(num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64, downsampling_factors=[[2, 2]])
    cwd = os.getcwd()
    total = samples * world
    tmp = _shared_tmpdir("clx_e2e_", rank, world)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    os.chdir(tmp)
    try:
        container = os.path.join(tmp, "data.zarr")
        if rank == 0:
            base = np.concatenate([synthetic_raw(1, (size, size), seed=s).numpy() for s in range(min(total, 16))], axis=0)
            raw = np.concatenate([base] * ((total + len(base) - 1) // len(base)), axis=0)[:total]
            f = zarr_io.open(container)
            f["test/raw"] = raw
            f["test/raw"].attrs["axis_names"] = ["s", "c", "y", "x"]
            torch.manual_seed(0)
            m = get_model(in_channels=1, out_channels=2, num_spatial_dims=2, **mcfg)
            for _n, layer in m.named_modules():
                if isinstance(layer, torch.nn.modules.conv._ConvNd):
                    torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
            os.makedirs("models", exist_ok=True)
            torch.save({"model_state_dict": m.state_dict()}, "models/best_loss.pth")
            del m
        barrier()

        def config(name):
            return ExperimentConfig(
                model_config=dict(checkpoint="models/best_loss.pth", **mcfg), object_size=30,
                normalization_factor=1.0,
                inference_config=dict(
                    dataset_config=dict(container_path=container, dataset_name="test/raw"),
                    prediction_dataset_config=dict(container_path=container, dataset_name=f"{name}/embeddings"),
                    detection_dataset_config=dict(container_path=container, dataset_name=f"{name}/detection",
                                                  secondary_dataset_name=f"{name}/embeddings"),
                    segmentation_dataset_config=dict(container_path=container,
                                                     dataset_name=f"{name}/segmentation",
                                                     secondary_dataset_name=f"{name}/detection"),
                    crop_size=[size + 16, size + 16], device=str(device)))

        times = []
        for run in ("warm", "timed"):
            torch.manual_seed(1 + rank)
            np.random.seed(1 + rank)
            if rank == 0:
                f_out = zarr_io.open(container)
                for extra in ("binary-segmentation", "centered-embeddings"):
                    if extra in f_out:
                        del f_out[extra]
            barrier()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                infer(config(run))
            barrier()
            times.append(time.perf_counter() - t0)
        seg = zarr_io.open(container, "r")["timed/segmentation"][...] if rank == 0 else None
    finally:
        os.chdir(cwd)
        if world > 1:
            torch.distributed.barrier()
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)
    dt = times[1]
    if world > 1:
        t = torch.tensor(times, dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        times = t.tolist()
        dt = times[1]
        if rank != 0:
            return None
    return dict(mpixels_s=round(total * size * size / dt / 1e6, 4), seconds=round(dt, 3), samples=total, ranks=world,
                samples_per_rank=samples, tile=size,
                ms_per_sample=round(dt / total * 1e3, 2), first_run_seconds=round(times[0], 3),
                objects_per_sample=round(float(np.mean([len(np.unique(s)) - 1 for s in seg[:, 0]])), 1),
                what="cellulus_amd.infer.infer(): zarr -> predict (32 noisy forwards per tile, CPU-generator noise "
                     "prefetched) -> detect (Otsu + mean-shift) -> segment (grow/shrink + size filter) -> 5 zarr "
                     "datasets, random-weight network, model load and plan build included"
                     + ("; samples sharded over the ranks (parallel.shard_range), no collective on the data path, "
                        "time = the slowest rank's between two barriers" if world > 1 else ""))



def bench_release():
    """bench.release_device_memory: collect the reference cycles of the previous leg's models and plans, then empty torch's cache"""
    import gc

    gc.collect()
    torch.cuda.empty_cache()

def infer_sharded(device, rank, world, samples_per_rank=16):
    """The `infer` object of a multi-rank bench line (every rank calls this): the whole job's Mpixels/s through the
    real infer() at the two tile sizes the metric names (512^2 = BASELINE configs[4], 256^2 = the metric string)."""
    big = e2e_infer(device, samples=samples_per_rank, size=512, rank=rank, world=world)
    small = e2e_infer(device, samples=samples_per_rank, size=256, rank=rank, world=world)
    if rank != 0:
        return None
    return {"metric": f"infer Mpixels/s (embed + mean-shift detect + segment) through infer(), 2D 512x512, {world} GPU(s)",
            "value": big["mpixels_s"], "unit": "Mpixels/s", "scaling": "weak", "n_gpus": world, "e2e": big,
            "at_256": dict(small, metric="the same at 2D 256x256 (one 272^2 tile per sample)")}


MFMA_KINDS = {0: "conv_igemm_kernel<128,128,2,2>", 1: "conv_igemm_kernel<128,64,4,1>", 2: "conv_wgrad_kernel",
              3: "gemm_sp_kernel<0>", 6: "chain64_kernels", 14: "wino_fused_kernels", 16: "gemm_sp2_kernel"}


def embed_stage(model, device, size, n_it, reps):
    """2 * n_it noisy forwards + mean/std of one reflect-padded (size + 16)^2 tile, device-resident noise:
    seconds per tile, the result, and what libclx's launch events saw of every MFMA kernel."""
    import ctypes

    from cellulus_amd import _clx

    crop = size + 16
    rng = np.random.default_rng(0)
    raw = torch.from_numpy(np.pad(rng.random((1, 1, size, size), dtype=np.float32),
                                  [(0, 0), (0, 0), (8, 8), (8, 8)], mode="reflect")).to(device)
    torch.manual_seed(1000 + size)                                  # the same noise in every call: results are comparable
    noise = torch.rand(1, 2 * n_it, 1, crop, crop, device=device)
    model.infer_on_device(raw, noise=noise)                       # warm-up (plan + packing)
    _clx.call("clx_profile_enable", 2)
    t_embed, emb = _sync_time(lambda: model.infer_on_device(raw, noise=noise), reps)
    lib = _clx.load()
    prof = {}
    for kind, kname in MFMA_KINDS.items():
        n_l, ms_l, fl_l = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        lib.clx_profile_read(kind, ctypes.byref(n_l), ctypes.byref(ms_l), ctypes.byref(fl_l))
        prof[kname] = (n_l.value / reps, ms_l.value / reps, fl_l.value / reps)      # per tile
    _clx.call("clx_profile_enable", 0)
    assert tuple(emb.shape) == (1, 3, size, size)
    return t_embed, emb, prof


def post_stages(device, size, reps):
    """detect (Otsu + mean-shift, reduction_probability 0.1) and segment (grow/shrink + size filter) on the synthetic
    disc embeddings of SURVEY.md §8d at size^2: seconds each, the label maps, the inputs (for the CPU leg)."""
    from cellulus_amd.segment import grow_shrink_on_device
    from cellulus_amd.utils.mean_shift import mean_shift_on_device
    from cellulus_amd.utils.misc import label_on_device
    from cellulus_amd.utils.otsu import threshold_otsu

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
    return t_detect, t_segment, (labels, centers, seg, ncomp), (mean, std, mean_d, std_d)


def infer_bench(device, reps=2, with_cpu=True, with_e2e=True, with_streaming=True):
    from cellulus_amd.models import get_model
    from cellulus_amd.utils.mean_shift import mean_shift_on_device

    size, n_it = 512, 16
    cfg = dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3, features_in_last_layer=64,
               downsampling_factors=[[2, 2]], num_spatial_dims=2)
    torch.manual_seed(0)
    model = get_model(**cfg).to(device)
    for _n, layer in model.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    model.eval()
    model.set_infer(p_salt_pepper=0.01, num_infer_iterations=n_it, device=device)
    # the metric string names 2-D 256^2 for inference too: the same pipeline on one 272^2 tile, first (its plan is
    # dropped when the 528^2 one is built)
    # `value` is timed with the product's default (the chunks of noisy copies alternate between two plans on two streams
    # where a chunk fills the device); the kernels' own durations — the rooflines — come from a second pass with
    # CLX_INFER_STREAMS=1, every kernel alone on the device (as bench.py does for the training step)
    def both_passes(sz):
        keep = os.environ.get("CLX_INFER_STREAMS")
        t_default, emb_, _prof = embed_stage(model, device, sz, n_it, reps)
        two = getattr(model, "_infer_pair", None) is not None and model._infer_pair[0] is next(iter(model._plans.values()))
        os.environ["CLX_INFER_STREAMS"] = "1"
        try:
            t_one, _e, prof_one = embed_stage(model, device, sz, n_it, reps)
        finally:
            if keep is None:
                del os.environ["CLX_INFER_STREAMS"]
            else:
                os.environ["CLX_INFER_STREAMS"] = keep
        return t_default, emb_, prof_one, t_one, (len(model._infer_pair[2]) if two else 1)

    t_embed_s, _emb_s, prof_s, t_one_s, streams_s = both_passes(256)
    t_detect_s, t_segment_s, (_l, centers_s, _s, ncomp_s), _inputs = post_stages(device, 256, reps)
    t_embed, emb, prof, t_one, streams = both_passes(size)
    changed = dict(getattr(model, "_last_changed_rows", None) or {})
    if changed.get("used"):
        # the same tile with every 1x1 layer dense on every copy (CLX_SPARSE_NOISE=0): what the changed-rows form saves
        os.environ["CLX_SPARSE_NOISE"] = "0"
        try:
            t_dense, emb_dense, _p = embed_stage(model, device, size, n_it, reps)
        finally:
            del os.environ["CLX_SPARSE_NOISE"]
        changed.update(embed_ms=round(t_embed * 1e3, 2), embed_ms_dense=round(t_dense * 1e3, 2),
                       identical_to_dense=bool(torch.equal(emb, emb_dense)),
                       what="the 1x1 layers straight behind the first convolution run once on the clean tile and "
                            "again on the rows of each noisy copy that differ from it (`fraction`: the window-dilated "
                            "noise pixels), the Winograd layer behind them on the output tiles whose input window holds "
                            "such a row (`tile_fraction`); every copy's full tensors exist with the dense forward's "
                            "bits (identical_to_dense; tests/test_gpu_unet.py, test_gpu_fullsize.py); "
                            "CLX_SPARSE_NOISE=0 / CLX_SPARSE_TILES=0 switch the two off")
        changed["fraction"] = round(changed["fraction"], 4)
        if "tile_fraction" in changed:
            changed["tile_fraction"] = round(changed["tile_fraction"], 4)
    t_detect, t_segment, (labels, centers, seg, ncomp), (mean, std, mean_d, std_d) = post_stages(device, size, reps)

    # mean-shift at full density (reduction_probability 1.0: every foreground pixel is a seed —
    # the reference's 48.8 s case); reported as pair evaluations per second
    def detect_full():
        labels_f, centers_f = mean_shift_on_device(mean_d.clone(), std_d, 15.0, 1.0, 0.5, None)
        return labels_f, centers_f

    detect_full()
    t_full, (labels_full, centers_full) = _sync_time(detect_full, 2)
    nfg = int((std < 0.5).sum())

    total = t_embed + t_detect + t_segment
    total_s = t_embed_s + t_detect_s + t_segment_s
    from bench import conv_flops, traffic_lookup
    plan = next(iter(model._plans.values()))
    fwd_flops, _, _ = conv_flops(plan.topo, 1)

    def roofline_of(prof, t_embed_tile, t_value_pass, nstreams, plan_batch):
        dom, (launches, ms, flops) = max(prof.items(), key=lambda kv: kv[1][1])
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0

        def peak_of(kname):
            return BF16_MFMA_PEAK_TFLOPS / 6 if kname in SP_KERNELS else F32_MFMA_PEAK_TFLOPS

        peak = peak_of(dom)
        mfma_ms = sum(v[1] for v in prof.values())
        mfma_fl = sum(v[2] for v in prof.values())
        peak_seconds = sum(v[2] / (peak_of(k) * 1e12) for k, v in prof.items())     # every kernel priced against ITS peak
        traffic, traffic_source = traffic_lookup(dom, "infer")
        return dict(bound="mfma", kernel=dom, achieved=round(achieved, 2), peak=round(peak, 1),
                    unit="TFLOP/s", frac=round(achieved / peak, 4), traffic=traffic, traffic_source=traffic_source,
                    launches_per_tile=int(round(launches)), avg_launch_ms=round(ms / max(launches, 1), 4),
                    forwards_per_launch=int(plan_batch),
                    share_of_embed_stage=round(ms * 1e-3 / t_embed_tile, 4),
                    step_mfma_frac=round(peak_seconds / t_embed_tile, 4),
                    all_mfma_kernels=dict(tflops=round(mfma_fl / (mfma_ms * 1e-3) / 1e12, 2) if mfma_ms else 0.0,
                                          ms_per_tile=round(mfma_ms, 3),
                                          share_of_embed_stage=round(mfma_ms * 1e-3 / t_embed_tile, 4)),
                    per_kernel={k: dict(launches_per_tile=int(round(v[0])), ms_per_tile=round(v[1], 3),
                                        tflops=round(v[2] / (v[1] * 1e-3) / 1e12, 2) if v[1] else 0.0)
                                for k, v in prof.items() if v[0]},
                    timed_in=f"a second pass on ONE stream (every kernel alone on the device: {t_embed_tile * 1e3:.2f} ms per tile); "
                             f"the `value` pass runs the chunks on {nstreams} stream(s): {t_value_pass * 1e3:.2f} ms per tile",
                    one_stream_embed_ms=round(t_embed_tile * 1e3, 2), streams_value_pass=nstreams,
                    step_mfma_frac_value_pass=round(peak_seconds / t_value_pass, 4),
                    note="achieved = FLOPs the dominant kernel executes / HIP-event time of its launches; "
                         "share_of_embed_stage = its launches' time / the embedding stage's wall time per tile (the "
                         "rest: the other MFMA kernels under all_mfma_kernels, Winograd transforms, first-layer and "
                         "pooling / upsampling kernels, noise injection, mean/std: profiles/*_infer_tile_kernels.txt); "
                         "step_mfma_frac = executed FLOPs of ALL MFMA kernels of a tile / the embedding stage's wall "
                         "time (one-stream pass; step_mfma_frac_value_pass: the `value` pass) / peak; the HBM-bound "
                         "kernels of detect / segment are under `streaming`")

    out = {
        "metric": "infer Mpixels/s (embed + mean-shift detect + segment), 2D 512x512, 1 GPU",
        "value": round(size * size / total / 1e6, 4),
        "unit": "Mpixels/s",
        "stage_ms": {"embed": round(t_embed * 1e3, 2), "detect": round(t_detect * 1e3, 3),
                     "segment": round(t_segment * 1e3, 3)},
        "embed_tflops": round(2 * n_it * fwd_flops / t_embed / 1e12, 2),
        "embed_tflops_note": "direct-form FLOPs of the 32 dense forwards / embedding time: a work rate, NOT a utilisation "
                             "(the Winograd layers execute 1/4 of their direct multiplies, the changed-rows form of the "
                             "first level's 1x1 layers a tenth of theirs); utilisation is roofline.frac / step_mfma_frac, "
                             "from the FLOPs the kernels execute",
        "changed_rows": changed,
        "roofline": roofline_of(prof, t_one, t_embed, streams, model.infer_chunk(2 * n_it, (size + 16, size + 16))),
        "at_256": {
            "metric": "infer Mpixels/s (embed + mean-shift detect + segment), 2D 256x256 (one 272^2 tile), 1 GPU",
            "value": round(256 * 256 / total_s / 1e6, 4), "unit": "Mpixels/s",
            "stage_ms": {"embed": round(t_embed_s * 1e3, 2), "detect": round(t_detect_s * 1e3, 3),
                         "segment": round(t_segment_s * 1e3, 3)},
            "roofline": roofline_of(prof_s, t_one_s, t_embed_s, streams_s, model.infer_chunk(2 * n_it, (272, 272))),
            "objects": int(ncomp_s.item()), "clusters": int(len(centers_s))},
        "meanshift_rp1": {"ms": round(t_full * 1e3, 2), "seeds": nfg, "clusters": int(len(centers_full))},
        "objects": int(ncomp.item()),
        "clusters": int(len(centers)),
    }
    del model, plan, emb
    bench_release()
    if with_streaming:
        try:
            out["streaming"] = streaming_rooflines(device)
            bench_release()
            # the two mean-shift kernels again at 256 samples per launch: what a launch of 64 samples loses is ramp and
            # launch cost, not bandwidth
            out["streaming"]["mean_shift_at_8192"] = streaming_rooflines(device, 8192, only_mean_shift=True)
        except Exception as e:
            out["streaming"] = {"error": f"{type(e).__name__}: {e}"}
    if with_e2e:
        try:
            out["e2e"] = e2e_infer(device)
            out["e2e"]["vs_kernel_only"] = round(out["e2e"]["mpixels_s"] / out["value"], 3)
            out["at_256"]["e2e"] = e2e_infer(device, size=256)
            out["at_256"]["e2e"]["vs_kernel_only"] = round(out["at_256"]["e2e"]["mpixels_s"] / out["at_256"]["value"], 3)
        except Exception as e:
            out["e2e"] = {"error": f"{type(e).__name__}: {e}"}
    if with_cpu:
        # CPU baseline leg — the only use of oracle/ here.  (1) the reference's own library call:
        # sklearn.cluster.MeanShift exactly as cellulus/utils/mean_shift.py:62-76 drives it (serial),
        # when scikit-learn is importable; (2) the oracle's C restatement of the same algorithm on 1 core
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
                             "cores": 1, "kind": "port (C restatement of sklearn MeanShift + scipy EDT + C labelling)",
                             "labels_identical": same}
        try:
            import sklearn
            from sklearn.cluster import MeanShift

            m = mean[0].copy()
            m[0] += np.arange(size)[None, :]
            m[1] += np.arange(size)[:, None]
            Xall = np.moveaxis(m, 0, -1)[std < thr].reshape(-1, 2)
            np.random.seed(1)
            t0 = time.perf_counter()
            keep = np.random.rand(len(Xall)) < 0.1
            ms_ = MeanShift(bandwidth=15.0, cluster_all=False, seeds=None)      # mean_shift.py:62-66
            ms_.fit(Xall[keep])
            lab = ms_.predict(Xall)
            t_sk = time.perf_counter() - t0
            out["cpu_sklearn"] = {"detect_ms": round(t_sk * 1e3, 1), "cores": 1, "kind": "reference library",
                                  "what": f"sklearn {sklearn.__version__} MeanShift(bandwidth=15, cluster_all=False)"
                                          ".fit(10 % of the foreground).predict(all), n_jobs=None as the reference",
                                  "clusters": int(len(ms_.cluster_centers_)), "foreground": int(len(Xall)),
                                  "labels_used": int(lab.max() + 1)}
        except Exception as e:
            out["cpu_sklearn"] = {"error": f"{type(e).__name__}: {e}"}
    return out
