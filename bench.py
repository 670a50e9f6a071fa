#!/usr/bin/env python3
"""Benchmark of the Cellulus hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload train2d|train3d]

A "step" is one `train_iteration` (U-Net forward, fused gather + OCE loss,
backward, [RCCL all-reduce SUM], Adam) on one batch of synthetic crops per GPU,
at BASELINE.json configs[1]: 2-D 1x256x256 crops, num_fmaps=256,
fmap_inc_factor=3, downsampling [[2,2]], batch 8 per GPU.  Inputs are resident
in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     — the dominant kernel (f32 MFMA implicit-GEMM convolution):
                 algorithmic FLOPs of its launches / their HIP-event durations,
                 against the 157.3 TFLOP/s f32 MFMA peak.
  cpu_baseline — the oracle's CPU train step (plain PyTorch, all host cores) on
                 a bounded sample of the same workload; baseline only.
  infer        — inference throughput (embed + mean-shift + CC) in Mpixels/s.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense

WORKLOADS = {
    "train2d": dict(
        name="2D 1x256x256 crops, num_fmaps=256, fmap_inc_factor=3, downsampling=[[2,2]], batch 8/GPU",
        model=dict(in_channels=1, out_channels=2, num_fmaps=256, fmap_inc_factor=3,
                   features_in_last_layer=64, downsampling_factors=[[2, 2]], num_spatial_dims=2),
        crop=(256, 256), batch=8, kappa=10.0, density=0.1),
    "train3d": dict(
        name="3D 1x64x64x64 crops, num_fmaps=64, fmap_inc_factor=3, downsampling=[[2,2,2]], batch 8/GPU",
        model=dict(in_channels=1, out_channels=3, num_fmaps=64, fmap_inc_factor=3,
                   features_in_last_layer=64, downsampling_factors=[[2, 2, 2]], num_spatial_dims=3),
        crop=(64, 64, 64), batch=8, kappa=10.0, density=0.1),
    "tiny": dict(  # smoke-sized, for debugging the harness only
        name="2D 1x64x64 crops, num_fmaps=16, batch 2/GPU (harness check, not a benchmark)",
        model=dict(in_channels=1, out_channels=2, num_fmaps=16, fmap_inc_factor=3,
                   features_in_last_layer=64, downsampling_factors=[[2, 2]], num_spatial_dims=2),
        crop=(64, 64), batch=2, kappa=4.0, density=0.1),
}


def synthetic_raw(batch, crop, seed):
    """Gaussian blobs (sigma 6) on a jittered 48-px grid + noise, in [0,1]."""
    rs = np.random.RandomState(seed)
    nd = len(crop)
    grids = np.meshgrid(*[np.arange(c, dtype=np.float32) for c in crop], indexing="ij")
    out = np.zeros((batch, 1) + tuple(crop), dtype=np.float32)
    for b in range(batch):
        img = np.zeros(crop, dtype=np.float32)
        centres = np.stack(np.meshgrid(*[np.arange(24, c, 48) for c in crop], indexing="ij"), -1).reshape(-1, nd)
        for c in centres:
            c = c + rs.randint(-6, 7, size=nd)
            d2 = sum((g - ci) ** 2 for g, ci in zip(grids, c))
            img += np.exp(-d2 / (2 * 6.0 ** 2))
        img += rs.normal(0, 0.02, size=crop).astype(np.float32)
        out[b, 0] = np.clip(img, 0, 1)
    return torch.from_numpy(out)


def sample_pairs(batch, crop, kappa, density, seed):
    """Pair coordinates drawn exactly as cellulus/datasets/zarr_dataset.py:177-251."""
    from cellulus_amd.datasets.zarr_dataset import ZarrDataset

    ds = ZarrDataset.__new__(ZarrDataset)
    ds.num_spatial_dims = len(crop)
    ds.kappa = kappa
    ds.density = density
    ds.output_shape = tuple(int(c - 16) for c in crop)
    ds.unbiased_shape = tuple(int(c - 2 * kappa) for c in ds.output_shape)
    np.random.seed(seed)
    anchors, refs = [], []
    for _ in range(batch):
        a, r = ds.sample_coordinates()
        anchors.append(a)
        refs.append(r)
    return (torch.from_numpy(np.stack(anchors).astype(np.int64)),
            torch.from_numpy(np.stack(refs).astype(np.int64)))


def conv_flops(topo, batch):
    """Algorithmic FLOPs (2*M*N*K) of every convolution: forward, dgrad, wgrad."""
    fwd = 0
    per_layer = {}
    for layer in topo.convs:
        m = batch * layer.out_shape[0] * layer.out_shape[1] * layer.out_shape[2]
        f = 2 * m * layer.cout * layer.cin * layer.taps
        per_layer[layer.name] = f
        fwd += f
    first = per_layer[topo.convs[0].name]
    train = 3 * fwd - first   # the first layer needs no data gradient
    return fwd, train, per_layer


class ConvTimer:
    """Brackets every clx_conv_fwd launch (forward + dgrad use) with HIP events on the
    launch stream and attributes algorithmic FLOPs to it."""

    def __init__(self):
        self.records = []
        self._orig = None

    def install(self):
        from cellulus_amd import _clx

        self._orig = _clx.call
        timer = self

        def call(name, *args):
            if name not in ("clx_conv_fwd", "clx_conv_wgrad"):
                return timer._orig(name, *args)
            d = args[0]._obj
            taps = d.KD * d.KH * d.KW
            ctot = d.src[0].C + (d.src[1].C if d.nsrc == 2 else 0)
            od, oh, ow = (d.ID + 2 * d.PD - d.KD + 1, d.IH + 2 * d.PH - d.KH + 1, d.IW + 2 * d.PW - d.KW + 1)
            m = d.B * od * oh * ow
            n = d.N
            flops = 2.0 * m * n * ctot * taps
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            timer._orig(name, *args)
            e1.record()
            big = (n > 64) if name == "clx_conv_fwd" else (n > 64 and ctot > 64)
            kind = name[4:] if d.PD + d.PH + d.PW == 0 or name == "clx_conv_wgrad" else "conv_dgrad"
            timer.records.append((name, big, flops, e0, e1, (kind, m, n, ctot * taps, d.nsrc)))

        _clx.call = call
        import cellulus_amd.models.plan as plan_mod
        plan_mod._clx.call = call

    def detail(self, steps):
        """per-layer table (averaged over the timed steps): kind, M, N, K, ms, TFLOP/s"""
        per_step = len(self.records) // max(steps, 1)
        rows = []
        for i in range(per_step):
            ms = [self.records[i + s * per_step][3].elapsed_time(self.records[i + s * per_step][4])
                  for s in range(steps)]
            _n, _b, flops, _e0, _e1, shape = self.records[i]
            t = float(np.median(ms))
            rows.append((shape, t, flops / (t * 1e-3) / 1e12))
        return rows

    def uninstall(self):
        from cellulus_amd import _clx

        if self._orig is not None:
            _clx.call = self._orig

    def summary(self):
        out = {}
        for name, big, flops, e0, e1, _shape in self.records:
            key = (name, big)
            ms = e0.elapsed_time(e1)
            agg = out.setdefault(key, [0, 0.0, 0.0])
            agg[0] += 1
            agg[1] += flops
            agg[2] += ms
        return out


def cpu_baseline(workload, sample_batch, seed):
    """Oracle train step on the host cores (plain PyTorch fp32) — reported beside the GPU number."""
    from oracle import unet_oracle as O

    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    torch.manual_seed(seed)
    model = O.OracleUNetModel(**workload["model"])
    for _n, layer in model.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    opt = torch.optim.Adam(model.parameters(), lr=4e-5, weight_decay=0.01)
    raw = synthetic_raw(sample_batch, workload["crop"], seed)
    anchor, reference = sample_pairs(sample_batch, workload["crop"], workload["kappa"], workload["density"], seed)
    t0 = time.perf_counter()
    O.train_step(model, opt, raw, anchor, reference, 10.0, 1e-5)
    dt = time.perf_counter() - t0
    return dict(value=sample_batch / dt, unit="crops/s", cores=threads, kind="port",
                sample=f"1 train step (forward+loss+backward+Adam) of the PyTorch-CPU oracle on {sample_batch} "
                       f"crop(s) of the same workload, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="train2d", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-infer", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1, help="crops in the CPU baseline sample")
    args = ap.parse_args()

    from cellulus_amd import parallel
    from cellulus_amd.criterions import get_loss
    from cellulus_amd.models import get_model
    from cellulus_amd.optim import Adam
    from cellulus_amd.train import train_iteration

    rank, world, local_rank = parallel.init_from_env()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path exists)")
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    wl = WORKLOADS[args.workload]

    torch.manual_seed(0)
    model = get_model(**wl["model"]).to(device)
    for _n, layer in model.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    flat, _ = model.flatten_parameters()
    parallel.broadcast_(flat, 0)
    nd = wl["model"]["num_spatial_dims"]
    criterion = get_loss(temperature=10.0, regularizer_weight=1e-5, density=wl["density"],
                         num_spatial_dims=nd, device=device)
    optimizer = Adam(model.parameters(), lr=4e-5, weight_decay=0.01)

    B = wl["batch"]
    raw = synthetic_raw(B, wl["crop"], seed=rank).to(device)
    anchor, reference = sample_pairs(B, wl["crop"], wl["kappa"], wl["density"], seed=rank)
    anchor, reference = anchor.to(device), reference.to(device)
    batch = (raw, anchor, reference)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    timer = ConvTimer()
    timer.install()          # installed before the warm-up so lazy HIP-event setup is not timed
    for _ in range(args.warmup):
        train_iteration(batch, model, criterion, optimizer, device)
    torch.cuda.synchronize()
    timer.records.clear()
    import gc

    gc.collect()
    gc.disable()     # a cyclic-GC pause inside one step would be charged to the GPU path
    from cellulus_amd import _clx as _clx_mod
    _clx_mod.call("clx_profile_enable", 2)      # HIP events around every MFMA kernel launch
    barrier()
    t0 = time.perf_counter()
    step_times = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        loss, oce, _ = train_iteration(batch, model, criterion, optimizer, device)
        step_times.append(time.perf_counter() - ts)
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if os.environ.get("CLX_BENCH_DETAIL") and rank == 0:
        print("  per-step wall ms:", [round(t * 1e3, 2) for t in step_times], file=sys.stderr)
    timer.uninstall()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()

    if world > 1:
        torch.distributed.barrier()
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    crops_per_s = world * B * args.steps / dt
    plan = next(iter(model._plans.values()))
    fwd_flops, train_flops, _ = conv_flops(plan.topo, 1)

    # ---- roofline of the dominant kernel: executed MFMA FLOPs of its launches / their
    # HIP-event durations (events recorded inside libclx around the kernel launch itself)
    import ctypes

    lib = _clx_mod.load()
    kinds = {0: "conv_igemm_kernel<128,128,2,2>", 1: "conv_igemm_kernel<128,64,4,1>", 2: "conv_wgrad_kernel"}
    prof = {}
    for kind, kname in kinds.items():
        n_l, ms_l, fl_l = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        lib.clx_profile_read(kind, ctypes.byref(n_l), ctypes.byref(ms_l), ctypes.byref(fl_l))
        prof[kname] = (n_l.value, ms_l.value, fl_l.value)
    _clx_mod.call("clx_profile_enable", 0)
    dom_name, (launches, ms, flops) = max(prof.items(), key=lambda kv: kv[1][1])
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    mfma_ms = sum(v[1] for v in prof.values())
    mfma_fl = sum(v[2] for v in prof.values())
    summ = timer.summary()
    conv_ms = sum(v[2] for v in summ.values())
    plan_algo = getattr(plan, "algo", {})
    n_wino = sum(1 for a in plan_algo.values() if a.get("fwd"))
    wino_tile = max([{1: 2, 2: 4}.get(a.get("fwd"), 0) for a in plan_algo.values()] or [0])
    # HBM bytes per launch of that kernel: PMC counters cannot be read from inside the process, so the
    # figure comes from the committed digest of the separate rocprofv3 --pmc passes (tools/hbm_traffic.py)
    traffic, traffic_source = None, None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"hbm_traffic_{args.workload}.json")
    if os.path.exists(tpath):
        with open(tpath) as fh:
            tdoc = json.load(fh)
        for kname, row in tdoc["kernels"].items():
            if kname.replace(" ", "") == dom_name:
                traffic, traffic_source = row["bytes_per_launch"], "profiles/" + os.path.basename(tpath)
    roofline = dict(
        bound="mfma", kernel=dom_name,
        achieved=round(achieved, 2), peak=F32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
        frac=round(achieved / F32_MFMA_PEAK_TFLOPS, 4), traffic=traffic,
        traffic_unit="bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", traffic_source=traffic_source,
        launches_per_step=int(launches // args.steps),
        avg_launch_ms=round(ms / max(launches, 1), 4),
        note="achieved = FLOPs the kernel executes (2*M*N*K per GEMM, real extents) / HIP-event time of "
             "its launches; Winograd F(2x2) / F(4x4) layers execute 4/9 / 1/4 of the direct-convolution FLOPs",
        all_mfma_kernels=dict(tflops=round(mfma_fl / (mfma_ms * 1e-3) / 1e12, 2) if mfma_ms else 0.0,
                              ms_per_step=round(mfma_ms / args.steps, 3)),
        conv_calls_ms_per_step=round(conv_ms / args.steps, 3),
        winograd_layers=n_wino, winograd_tile=wino_tile,
        direct_equivalent_tflops=round(crops_per_s / world * train_flops / 1e12, 2),
    )

    out = {
        "metric": "train crops/sec (U-Net fwd+bwd + OCE loss + Adam)",
        "value": round(crops_per_s, 3),
        "unit": "crops/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl["name"], "global_batch": world * B,
                   "parallelism": f"dp{world}", "gflop_per_crop_train": round(train_flops / 1e9, 1)},
        "loss": round(float(loss), 4),
        "roofline": roofline,
    }
    if os.environ.get("CLX_BENCH_DETAIL"):
        for (kind, m, n, k, nsrc), t, tf in timer.detail(args.steps):
            print(f"  {kind:11s} M={m:8d} N={n:5d} K={k:6d} src={nsrc} {t:8.3f} ms {tf:7.1f} TF/s", file=sys.stderr)
    if world == 1 and not args.no_infer:
        try:
            from bench_infer import infer_bench

            out["infer"] = infer_bench(device)
        except Exception as e:  # the train line must survive an inference-side failure
            out["infer"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, args.cpu_sample, seed=0)
    print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
