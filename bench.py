#!/usr/bin/env python3
"""Benchmark of the Cellulus hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload train2d|train3d]

A "step" is one `train_iteration` (U-Net forward, fused gather + OCE loss,
backward, [RCCL all-reduce SUM], Adam) on one batch of synthetic crops per GPU,
at BASELINE.json configs[1]: 2-D 1x256x256 crops, num_fmaps=256,
fmap_inc_factor=3, downsampling [[2,2]], batch 8 per GPU.  Inputs are resident
in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

`--gpus N` with N > 1: either the caller starts the ranks (`python -m
torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, WORLD_SIZE set) or —
plain `python bench.py --gpus N` — this process starts N children itself, one per GPU,
BEFORE anything touches the GPU, waits for them and relays rank 0's JSON line.

Extra objects on that line:
  roofline     — the dominant kernel (f32 MFMA implicit-GEMM convolution):
                 FLOPs its launches execute / their HIP-event durations,
                 against the 157.3 TFLOP/s f32 MFMA peak.
  train3d      — BASELINE.json configs[3] (3-D 64^3, 64 fmaps) timed the same way,
                 with its own roofline object.
  infer        — inference throughput (embed + mean-shift + CC) in Mpixels/s: kernel-level stages, rooflines and the
                 real infer() on one GPU; with --gpus N the real infer() over a zarr whose samples are sharded over the
                 ranks (no collective), whole-job Mpixels/s at 512^2 and at 256^2.
  train_e2e    — the real train() (zarr -> loader processes with the default augmentation and the
                 pair stream -> H2D -> step -> logging), steady-state crops/s; with --gpus N data-parallel, every
                 rank with its own loader processes (train.loader_policy), whole-job crops/s.
  cpu_baseline — the oracle's CPU train step (plain PyTorch, host cores) on
                 a bounded sample of the same workload; baseline only (1 GPU).
  gpu_library_baseline — the same plain-PyTorch model moved to the GPU (what the reference does with device = "cuda:0":
                 MIOpen convolutions + autograd): train step and one infer-mode tile; baseline only (1 GPU).
  ranks_seen, per_rank_ms_per_step, allreduce_ms_exposed — what the data-parallel run saw.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6  # dense bf16 MFMA; the split precision f32x3bf16 spends six bf16 products per f32 product
SP_KERNELS = ("gemm_sp_kernel<0>", "gemm_sp_kernel<1>", "gemm_sp2_kernel")   # forward / data gradient (K > 1024), weight gradient, forward / data gradient on 128 x 128 tiles          # priced against BF16_MFMA_PEAK_TFLOPS / 6 (f32-equivalent FLOPs)

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


# ------------------------------------------------------------------------------------------------
# launching the ranks (no torch import, no HIP call in this part)
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port



def release_device_memory():
    """Between the legs of a run: the previous leg's models and plans hold each other in reference cycles (modules, plans,
    closures), so their tensors stay LIVE until a generation-2 collection — collect first, then hand torch's cached blocks
    back (eight ranks rehearsing on one GPU ran it out of memory with the previous leg's 27 GB still allocated)."""
    import gc

    import torch

    gc.collect()
    torch.cuda.empty_cache()

def self_launch(n, argv):
    """Start one child per GPU, wait, relay rank 0's stdout.  Children are FRESH processes
    (Popen of this very file), so nothing GPU-initialised is ever replaced or forked."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(_free_port())
    env["WORLD_SIZE"] = str(n)
    env["LOCAL_WORLD_SIZE"] = str(n)          # one node: what torch.distributed.run would export
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        renv = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=renv, cwd=os.getcwd(),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() is not None and p.returncode != 0:
                failed = (r, p.returncode)
        time.sleep(0.2)
    if failed is not None:          # a dead rank leaves the others waiting in a collective: end them
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    out = procs[0].stdout.read().decode("utf-8", "replace") if procs[0].stdout else ""
    sys.stdout.write(out)
    sys.stdout.flush()
    for r, p in enumerate(procs):
        if p.returncode != 0 and failed is None:
            failed = (r, p.returncode)
    if failed is not None:
        print(f"bench.py: rank {failed[0]} of {n} exited with code {failed[1]}", file=sys.stderr)
        return 1
    return 0


# ------------------------------------------------------------------------------------------------
# synthetic inputs
# ------------------------------------------------------------------------------------------------
def synthetic_raw(batch, crop, seed):
    """Gaussian blobs (sigma 6) on a jittered 48-px grid + noise, in [0,1]."""
    import numpy as np
    import torch

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
    import numpy as np
    import torch

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


def traffic_lookup(kernel, wl_key):
    """HBM bytes per launch of `kernel` from the committed digest of the separate rocprofv3 --pmc passes
    (tools/hbm_traffic.py -> profiles/hbm_traffic_<workload>.json); (None, None) if the digest has no such kernel.
    The digest's names carry every template argument (`conv_igemm_kernel<128, 128, 2, 2, 1>`), the profile slots
    of libclx the tile shape only (`conv_igemm_kernel<128,128,2,2>`): every instantiation of that tile counts,
    weighted by its launches."""
    tpath = os.path.join(ROOT, "profiles", f"hbm_traffic_{wl_key}.json")
    if not os.path.exists(tpath):
        return None, None
    with open(tpath) as fh:
        tdoc = json.load(fh)
    stem = kernel.replace(" ", "").rstrip(">")
    rows = [row for kname, row in tdoc["kernels"].items()
            if kname.replace(" ", "").rstrip(">") == stem or kname.replace(" ", "").startswith(stem + ",")]
    if not rows:
        return None, None
    n_l = sum(r["launches"] for r in rows)
    return int(sum(r["bytes_per_launch"] * r["launches"] for r in rows) / max(n_l, 1)), "profiles/" + os.path.basename(tpath)


class ConvTimer:
    """Brackets every clx_conv_fwd / clx_conv_wgrad CALL (all launches of the call: transforms +
    GEMMs) with HIP events on the launch stream; CLX_BENCH_DETAIL=1 prints the per-layer table."""

    def __init__(self):
        self.records = []
        self._orig = None

    def install(self):
        import torch

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

    def detail(self, steps):
        """per-layer table (median over the timed steps): kind, M, N, K, ms, TFLOP/s"""
        import numpy as np

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

    def total_ms(self):
        return sum(e0.elapsed_time(e1) for _n, _b, _f, e0, e1, _s in self.records)


def build_step_inputs(wl_key, rank, device, broadcast=True):
    """Model (seed 0, Kaiming-normal weights as train.py:65-68, rank 0's copy on every rank), loss, optimizer and
    one HBM-resident batch of rank `rank` (crops and pair coordinates seeded by the rank)."""
    import torch

    from cellulus_amd import parallel
    from cellulus_amd.criterions import get_loss
    from cellulus_amd.models import get_model
    from cellulus_amd.optim import Adam

    wl = WORKLOADS[wl_key]
    torch.manual_seed(0)
    model = get_model(**wl["model"]).to(device)
    for _n, layer in model.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    flat, _ = model.flatten_parameters()
    if broadcast:
        parallel.broadcast_(flat, 0)
    nd = wl["model"]["num_spatial_dims"]
    criterion = get_loss(temperature=10.0, regularizer_weight=1e-5, density=wl["density"],
                         num_spatial_dims=nd, device=device)
    optimizer = Adam(model.parameters(), lr=4e-5, weight_decay=0.01)
    B = wl["batch"]
    raw = synthetic_raw(B, wl["crop"], seed=rank).to(device)
    anchor, reference = sample_pairs(B, wl["crop"], wl["kappa"], wl["density"], seed=rank)
    return model, criterion, optimizer, (raw, anchor.to(device), reference.to(device))


# ------------------------------------------------------------------------------------------------
# the timed workload
# ------------------------------------------------------------------------------------------------
def run_workload(wl_key, args, rank, world, device):
    """W warm-up steps, then exactly K timed steps between barrier + synchronize; returns the
    metrics dict on rank 0 (None elsewhere).  Every rank calls this (collectives inside)."""
    import ctypes
    import gc

    import torch

    from cellulus_amd import _clx, parallel
    from cellulus_amd import train as train_mod
    from cellulus_amd.train import train_iteration

    wl = WORKLOADS[wl_key]
    model, criterion, optimizer, batch = build_step_inputs(wl_key, rank, device)
    B = wl["batch"]
    torch.cuda.reset_peak_memory_stats(device)
    first_losses = []

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    timer = ConvTimer()
    detail = bool(os.environ.get("CLX_BENCH_DETAIL"))
    if detail:
        os.environ["CLX_STREAMS"] = "1"      # the per-layer table times calls one after the other: one stream
        timer.install()      # installed before the warm-up so lazy HIP-event setup is not timed
    for _ in range(args.warmup):
        first_losses.append(train_iteration(batch, model, criterion, optimizer, device)[0])
    torch.cuda.synchronize()
    timer.records.clear()
    gc.collect()
    gc.disable()     # a cyclic-GC pause inside one step would be charged to the GPU path
    _clx.call("clx_profile_enable", 2)      # HIP events around every MFMA kernel launch, on its stream
    _lib0 = _clx.load()
    _lib0.clx_profile_clock(None, None, 1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, oce, _ = train_iteration(batch, model, criterion, optimizer, device)
        first_losses.append(loss)
    barrier()
    dt_local = time.perf_counter() - t0
    gc.enable()
    timer.uninstall()

    lib = _clx.load()
    kinds = {0: "conv_igemm_kernel<128,128,2,2>", 1: "conv_igemm_kernel<128,64,4,1>", 2: "conv_wgrad_kernel",
             3: "gemm_sp_kernel<0>", 4: "gemm_sp_kernel<1>", 6: "chain64_kernels", 14: "wino_fused_kernels",
             16: "gemm_sp2_kernel"}
    hbm_kinds = {5: "sp_split_kernel", 15: "wino_transform_kernels"}       # HBM-bound launches libclx stamps as well (no FLOPs)
    hbm_prof = {}

    def read_clock(reset=True):
        """MHz the MFMA kernels ran at since the last reset (clx_profile_clock), None if nothing was recorded"""
        c, w = ctypes.c_double(), ctypes.c_double()
        lib.clx_profile_clock(ctypes.byref(c), ctypes.byref(w), 1 if reset else 0)
        return round(100.0 * c.value / w.value, 1) if w.value > 0 else None

    def read_profile():
        prof = {}
        for kind, kname in kinds.items():
            n_l, ms_l, fl_l = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            lib.clx_profile_read(kind, ctypes.byref(n_l), ctypes.byref(ms_l), ctypes.byref(fl_l))
            prof[kname] = (n_l.value, ms_l.value, fl_l.value)
        for kind, kname in hbm_kinds.items():
            n_l, ms_l, fl_l = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            lib.clx_profile_read(kind, ctypes.byref(n_l), ctypes.byref(ms_l), ctypes.byref(fl_l))
            hbm_prof[kname] = (n_l.value, ms_l.value)
        _clx.call("clx_profile_enable", 0)
        return prof

    prof = read_profile()
    clock_mhz = read_clock()
    # ---- what the data-parallel run saw
    dt, per_rank, ranks_seen, exposed = dt_local, [dt_local], 1, None
    dp = {}
    if world > 1:
        # what every rank holds after the K steps: the same parameters, the same bucket ranges (they depend on the
        # launch plan only), and its share of the device memory (ranks that share a device in a dress rehearsal)
        mine = model._flat.clone()
        parallel.broadcast_(mine, 0)
        same = torch.tensor([1.0 if torch.equal(mine, model._flat) else 0.0], device=device)
        torch.distributed.all_reduce(same, op=torch.distributed.ReduceOp.MIN)
        ranges = [None] * world
        torch.distributed.all_gather_object(ranges, [list(r) for r in getattr(model, "_last_bucket_ranges", [])])
        peaks = [None] * world
        torch.distributed.all_gather_object(peaks, round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 3))
        dp = dict(params_identical_on_all_ranks=bool(same.item() == 1.0),
                  bucket_ranges_identical_on_all_ranks=all(r == ranges[0] for r in ranges),
                  per_rank_peak_mem_gb=peaks)
        del mine
        ones = torch.ones(1, dtype=torch.float32, device=device)
        torch.distributed.all_reduce(ones)            # RCCL (or the backend under test) counts the ranks
        ranks_seen = int(round(ones.item()))
        t = torch.tensor([dt_local], dtype=torch.float64, device=device)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(gathered, t)
        per_rank = [g.item() for g in gathered]
        dt = max(per_rank)                            # MAX over ranks
        # the same K steps with the gradient exchange switched off (each rank on its own): the
        # difference is what the all-reduce costs on the step's critical path
        real_world = parallel.world_size
        parallel.world_size = lambda: 1
        try:
            train_iteration(batch, model, criterion, optimizer, device)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                train_iteration(batch, model, criterion, optimizer, device)
            torch.cuda.synchronize()
            solo = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
        finally:
            parallel.world_size = real_world
        torch.distributed.all_reduce(solo, op=torch.distributed.ReduceOp.MAX)
        exposed = (dt - solo.item()) / args.steps * 1e3
        assert train_mod.parallel is parallel
    # Two half batches on two streams (plan.DualPlan, the default at this size): launches of the two streams
    # share the device, so a launch's event time is no longer the kernel's own time.  The kernels are therefore
    # timed a second time, alone, in K more steps of the same workload on ONE stream (every rank takes part);
    # `value` stays the two-stream figure, the roofline object says which pass its numbers are from.
    from cellulus_amd.models.plan import DualPlan
    overlapped, one_stream_dt, clock_one_stream = None, None, None
    if isinstance(next(iter(model._plans.values())), DualPlan):
        overlapped = prof
        keep_env = os.environ.get("CLX_STREAMS")
        os.environ["CLX_STREAMS"] = "1"
        try:
            model._plans = {}
            for _ in range(max(1, min(2, args.warmup))):
                train_iteration(batch, model, criterion, optimizer, device)
            gc.collect()
            gc.disable()
            _clx.call("clx_profile_enable", 2)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                train_iteration(batch, model, criterion, optimizer, device)
            barrier()
            one_stream_dt = time.perf_counter() - t1
            gc.enable()
            prof = read_profile()
            clock_one_stream = read_clock()
        finally:
            if keep_env is None:
                del os.environ["CLX_STREAMS"]
            else:
                os.environ["CLX_STREAMS"] = keep_env

    if rank != 0:
        return None

    crops_per_s = world * B * args.steps / dt
    plan = next(iter(model._plans.values()))
    _fwd_flops, train_flops, _ = conv_flops(plan.topo, 1)

    # ---- roofline of the dominant kernel: executed MFMA FLOPs of its launches / their
    # HIP-event durations (events recorded inside libclx around the kernel launch itself)
    dom_name, (launches, ms, flops) = max(prof.items(), key=lambda kv: kv[1][1])
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0

    def peak_of(kname):
        return BF16_MFMA_PEAK_TFLOPS / 6 if kname in SP_KERNELS else F32_MFMA_PEAK_TFLOPS

    peak = peak_of(dom_name)
    mfma_ms = sum(v[1] for v in prof.values())
    mfma_fl = sum(v[2] for v in prof.values())
    # what the step makes of the matrix cores: every kernel's FLOPs priced against ITS peak (seconds at peak / step time)
    peak_seconds = sum(v[2] / (peak_of(k) * 1e12) for k, v in prof.items())
    plan_algo = getattr(plan, "algo", {})
    n_wino = sum(1 for a in plan_algo.values() if a.get("fwd"))
    wino_tile = max([{1: 2, 2: 4}.get(a.get("fwd"), 0) for a in plan_algo.values()] or [0])
    # HBM bytes per launch of that kernel: PMC counters cannot be read from inside the process, so the
    # figure comes from the committed digest of the separate rocprofv3 --pmc passes (tools/hbm_traffic.py)
    traffic, traffic_source = traffic_lookup(dom_name, wl_key)
    roofline = dict(
        bound="mfma", kernel=dom_name,
        achieved=round(achieved, 2), peak=round(peak, 1), unit="TFLOP/s",
        frac=round(achieved / peak, 4), traffic=traffic,
        traffic_unit="bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", traffic_source=traffic_source,
        launches_per_step=int(launches // args.steps),
        avg_launch_ms=round(ms / max(launches, 1), 4),
        note="achieved = FLOPs the kernel executes (2*M*N*K per GEMM, real extents) / HIP-event time of "
             "its launches; Winograd F(2x2) / F(4x4) layers execute 4/9 / 1/4 of the direct-convolution FLOPs",
        step_mfma_frac=round(peak_seconds / dt, 4),
        all_mfma_kernels=dict(tflops=round(mfma_fl / (mfma_ms * 1e-3) / 1e12, 2) if mfma_ms else 0.0,
                              ms_per_step=round(mfma_ms / args.steps, 3)),
        per_kernel={k: dict(launches_per_step=int(v[0] // args.steps), ms_per_step=round(v[1] / args.steps, 3),
                            tflops=round(v[2] / (v[1] * 1e-3) / 1e12, 2) if v[1] else 0.0,
                            **({"peak": round(peak_of(k), 1)} if k in SP_KERNELS else {}))
                    for k, v in prof.items() if v[0]},
        hbm_bound_kernels={k: dict(launches_per_step=int(v[0] // args.steps), ms_per_step=round(v[1] / args.steps, 3))
                           for k, v in hbm_prof.items() if v[0]},
        winograd_layers=n_wino, winograd_tile=wino_tile,
        direct_equivalent_tflops=round(crops_per_s / world * train_flops / 1e12, 2),
    )
    # the 157.3 TFLOP/s peak is the matrix cores at 2.4 GHz; under this load the part runs slower (power): the clock the
    # GEMM kernels actually saw, and the fraction of the peak AT THAT CLOCK
    mhz = clock_one_stream if overlapped is not None else clock_mhz
    if mhz:
        roofline["shader_clock_mhz"] = mhz
        roofline["shader_clock_mhz_value_pass"] = clock_mhz
        roofline["frac_at_measured_clock"] = round(achieved / (peak * mhz / 2400.0), 4)
        roofline["shader_clock_note"] = ("s_memtime / s_memrealtime ticks of the middle block of every GEMM launch of the pass the "
                                         "kernel numbers are from; `frac` stays against the 2.4-GHz peak")
    roofline["step_mfma_frac_note"] = ("every MFMA kernel's FLOPs priced against ITS peak (157.3 TFLOP/s float32 MFMA; 419.4 "
                                       "f32-equivalent for the split-precision products): seconds at peak / the step's wall time (the "
                                       "`value` pass): what the whole step makes of the matrix cores")
    if overlapped is not None:
        o_l, o_ms, o_fl = overlapped[dom_name]
        roofline["timed_in"] = ("a second pass of the same K steps on ONE stream inside this run (each kernel alone on the "
                                "device); the `value` pass runs two half batches on two streams, whose launches overlap")
        roofline["one_stream_ms_per_step"] = round(one_stream_dt / args.steps * 1e3, 3)
        roofline["two_stream_pass"] = dict(
            launches_per_step=int(o_l // args.steps), avg_launch_ms=round(o_ms / max(o_l, 1), 4),
            tflops_while_sharing_the_device=round(o_fl / (o_ms * 1e-3) / 1e12, 2) if o_ms else 0.0,
            note="event time of a launch that shares the device with the other stream's launches")
    if detail:
        roofline["conv_calls_ms_per_step"] = round(timer.total_ms() / args.steps, 3)
        for (kind, m, n, k, nsrc), t, tf in timer.detail(args.steps):
            print(f"  {kind:11s} M={m:8d} N={n:5d} K={k:6d} src={nsrc} {t:8.3f} ms {tf:7.1f} TF/s", file=sys.stderr)
    out = {
        "value": round(crops_per_s, 3),
        "unit": "crops/s",
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "config": {"workload": wl["name"], "global_batch": world * B,
                   "parallelism": f"dp{world}", "gflop_per_crop_train": round(train_flops / 1e9, 1),
                   "streams_per_gpu": 2 if overlapped is not None else 1},
        "loss": round(float(loss), 4),
        "roofline": roofline,
    }
    if world > 1:
        out["ranks_seen"] = ranks_seen
        out["per_rank_ms_per_step"] = [round(t / args.steps * 1e3, 3) for t in per_rank]
        out["allreduce_ms_exposed"] = round(exposed, 3)
        out["backend"] = torch.distributed.get_backend()
        # what one step puts on the wire per rank: the flat f32 gradient (+ 4 float64 loss sums), in the
        # ranges parallel.GradientBuckets issued (suffixes of the flat buffer, in backward order)
        ranges = getattr(model, "_last_bucket_ranges", None) or [(0, model._flat_grad.numel())]
        out["allreduce_bytes"] = int(model._flat_grad.numel() * 4 + 4 * 8)
        out["allreduce_bucket_bytes"] = [int((hi - lo) * 4) for lo, hi in ranges]
        out["grad_bucket_mb"] = float(os.environ.get("CLX_GRAD_BUCKET_MB", "4"))
        out.update(dp)
    # the loss of every step from the first warm-up step on (several ranks: the all-reduced SUM over the ranks'
    # crops — the loss is a sum over pairs); step 0 runs on the seeded initial weights
    out["losses_from_first_step"] = [float(x) for x in first_losses]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # the timed steps read HBM-resident inputs: no loader process runs beside them (train_e2e has the loader)
    out["host_cores_per_rank"] = max(1, cores // world)
    out["loader_procs"] = 0
    del model, optimizer, plan
    return out


def _shared_tmpdir(prefix, rank, world):
    """A scratch directory every rank sees: rank 0 makes it, the others learn its path."""
    import tempfile

    import torch

    box = [tempfile.mkdtemp(prefix=prefix) if rank == 0 else None]
    if world > 1:
        torch.distributed.broadcast_object_list(box, src=0)
    return box[0]


def train_e2e(wl_key, device, iterations=140, settle=40, workers=8, rank=0, world=1):
    """The REAL ``cellulus_amd.train.train()`` at the benchmark configuration over a synthetic zarr: zarr
    reads, random crops, the reference's default elastic augmentation (train_config.py:124), the reference's
    np.random pair stream in the loader processes (train.py:38-44: DataLoader(num_workers=8)), H2D of every
    batch (train.py:161-166), loss logging — everything ``value`` leaves out.  Steady-state crops/s between
    iteration ``settle`` and the end (the start-up — loader processes, plan build — is reported beside it).

    Several ranks (every rank calls this): ``train()`` runs data-parallel exactly as ``torch.distributed.run`` would
    start it — each rank its own loader processes (``train.loader_policy``: the host's cores are shared) and crops,
    gradients all-reduced — and ``value`` is the whole job's crops/s at the slowest rank's pace."""
    import contextlib
    import io
    import shutil

    import numpy as np
    import torch

    import cellulus_amd.train as T
    from cellulus_amd.configs import ExperimentConfig
    from cellulus_amd.utils import zarr_io

    wl = WORKLOADS[wl_key]
    crop = list(wl["crop"])
    nd = len(crop)
    settle = min(settle, iterations // 3)
    tmp = _shared_tmpdir("clx_e2e_", rank, world)
    cwd = os.getcwd()
    stamps, mem, waits = [], [], []
    real = T.train_iteration
    real_stage = T._DevicePrefetcher._stage

    def stage_spy(self):
        t = time.perf_counter()
        real_stage(self)
        waits.append(time.perf_counter() - t)

    def spy(*a, **k):
        out = real(*a, **k)
        stamps.append(time.perf_counter())
        mem.append(torch.cuda.memory_allocated(device))
        return out

    try:
        os.chdir(tmp)
        if rank == 0:
            f = zarr_io.open("data.zarr")
            # images larger than the crop, as the augmentation needs (zarr_dataset.py:123-132)
            big = tuple(int(c * 1.5) for c in crop)
            f["train/raw"] = np.concatenate([synthetic_raw(1, big, s).numpy() for s in range(16)])
            f["train/raw"].attrs["axis_names"] = ["s", "c"] + ["z", "y", "x"][-nd:]
        if world > 1:
            torch.distributed.barrier()
        m = wl["model"]
        cfg = ExperimentConfig(
            normalization_factor=1.0, object_size=30,
            model_config=dict(num_fmaps=m["num_fmaps"], fmap_inc_factor=m["fmap_inc_factor"],
                              features_in_last_layer=m["features_in_last_layer"],
                              downsampling_factors=[list(x) for x in m["downsampling_factors"]]),
            train_config=dict(crop_size=crop, batch_size=wl["batch"], max_iterations=iterations, num_workers=workers,
                              kappa=wl["kappa"], density=wl["density"], device=str(device),
                              save_model_every=10 ** 6, save_best_model_every=10 ** 6, save_snapshot_every=10 ** 6,
                              train_data_config=dict(container_path="data.zarr", dataset_name="train/raw")))
        policy = T.loader_policy(world, cfg.train_config.num_workers)
        T.train_iteration = spy
        T._DevicePrefetcher._stage = stage_spy
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            T.train(cfg)
        total = time.perf_counter() - t0
    finally:
        T.train_iteration = real
        T._DevicePrefetcher._stage = real_stage
        os.chdir(cwd)
        if world > 1:
            torch.distributed.barrier()
        if rank == 0:
            shutil.rmtree(tmp, ignore_errors=True)
    steady = (stamps[-1] - stamps[settle]) / (len(stamps) - 1 - settle)
    gaps = np.diff(np.asarray(stamps[settle:]))
    w = np.asarray(waits[settle + 1:]) if len(waits) > settle + 1 else np.zeros(1)
    if world > 1:
        t = torch.tensor([steady], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        steady = t.item()
        if rank != 0:
            return None
    return dict(
        value=round(world * wl["batch"] / steady, 3), unit="crops/s", ms_per_iteration=round(steady * 1e3, 3),
        iterations=iterations, steady_from_iteration=settle, seconds_total=round(total, 2), ranks=world,
        iteration_ms_p95=round(float(np.percentile(gaps, 95)) * 1e3, 2), iteration_ms_max=round(float(gaps.max()) * 1e3, 2),
        # the main thread's wait for the loader's next batch + the H2D enqueue, per iteration (rank 0)
        loader_wait_ms=dict(mean=round(float(w.mean()) * 1e3, 3), p95=round(float(np.percentile(w, 95)) * 1e3, 3),
                            max=round(float(w.max()) * 1e3, 3)),
        loader_procs=policy["loader_procs"], loader_procs_note="per rank",
        pair_sampler="device (clx_sample_pairs)" if policy["device_pairs"] else "np.random in the loader processes (reference stream)",
        loader_policy=policy["why"],
        elastic_deform=bool(cfg.train_config.elastic_deform), host_cores=policy["host_cores_per_rank"],
        # (a reading may or may not include the prefetched next batch, 40 MB: compare window maxima)
        device_mem_growth_mb=round((max(mem[-20:]) - max(mem[settle:settle + 20])) / 2 ** 20, 3),
        what="cellulus_amd.train.train(): synthetic zarr (16 images 1.5x the crop) -> loader processes (random crop, "
             "elastic augmentation, pair sampling) -> pinned H2D prefetch -> train_iteration -> loss.csv; "
             "checkpoints / snapshots at cadences beyond the run")


def cpu_baseline(workload, device, seed, budget_s=90.0):
    """The oracle's train step on the host cores (plain PyTorch f32 — which IS the reference's
    CPU path: nn.ConvNd / MaxPool / Upsample / autograd / Adam on device='cpu'), warmed up, at the
    fastest of a few thread counts; and the HIP step's loss on the SAME crop and weights beside it."""
    import numpy as np
    import torch

    from oracle import unet_oracle as O

    from cellulus_amd.criterions import get_loss
    from cellulus_amd.models import get_model
    from cellulus_amd.optim import Adam
    from cellulus_amd.train import train_iteration

    logical = os.cpu_count() or 1
    try:
        import psutil

        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        physical = max(1, logical // 2)
    torch.manual_seed(seed)
    model = O.OracleUNetModel(**workload["model"])
    for _n, layer in model.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(model.parameters(), lr=4e-5, weight_decay=0.01)
    full = int(workload["batch"])
    raw = synthetic_raw(full, workload["crop"], seed)
    anchor, reference = sample_pairs(full, workload["crop"], workload["kappa"], workload["density"], seed)

    t_begin = time.perf_counter()
    # thread count: one forward per candidate (after one untimed forward that pays the start-up)
    candidates = sorted({c for c in (physical, 64, 32) if 1 <= c <= logical}, reverse=True)
    probe = {}
    with torch.no_grad():
        torch.set_num_threads(candidates[0])
        model(raw[:1])
        for c in candidates:
            torch.set_num_threads(c)
            t0 = time.perf_counter()
            model(raw[:1])
            probe[c] = time.perf_counter() - t0
            if time.perf_counter() - t_begin > budget_s / 3:
                break
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)

    # warm-up step on crop 0 (its loss is what the HIP path is checked against below)
    t0 = time.perf_counter()
    l_cpu, _o, _off = O.train_step(model, opt, raw[:1], anchor[:1], reference[:1], 10.0, 1e-5)
    t_warm = time.perf_counter() - t0
    # timed: a second step — on the workload's full batch if the budget allows (a warm step costs about as
    # much per crop as the warm-up step did), else on two crops, else on one
    left = budget_s - (time.perf_counter() - t_begin)
    nb = full if 1.1 * full * t_warm < left else 2 if 2.2 * t_warm < left else 1
    t0 = time.perf_counter()
    O.train_step(model, opt, raw[:nb], anchor[:nb], reference[:nb], 10.0, 1e-5)
    dt = time.perf_counter() - t0

    # the same crop and weights through the HIP path
    gm = get_model(**workload["model"])
    gm.load_state_dict(state0)
    gm = gm.to(device)
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=workload["density"],
                    num_spatial_dims=workload["model"]["num_spatial_dims"], device=device)
    gopt = Adam(gm.parameters(), lr=4e-5, weight_decay=0.01)
    l_gpu, _o, _off = train_iteration((raw[:1], anchor[:1], reference[:1]), gm, crit, gopt, device)
    return dict(value=round(nb / dt, 4), unit="crops/s", cores=threads, kind="port",
                sample=f"PyTorch-CPU oracle (the reference's own CPU ops), 1 warm-up train step ({t_warm:.1f} s, "
                       f"1 crop) then 1 timed train step on {nb} crop(s) of the same workload — batch {full} — ({dt:.1f} s), "
                       f"torch threads = {threads} (forward probe s: "
                       + ", ".join(f"{c}: {probe[c]:.2f}" for c in probe) + f"; host has {logical} logical cores)",
                loss_cpu=round(float(l_cpu), 4), loss_hip=round(float(l_gpu), 4),
                loss_abs_diff=float(np.abs(l_cpu - l_gpu)),
                loss_rel_diff=float(np.abs(l_cpu - l_gpu) / max(abs(l_cpu), 1e-12)))


def gpu_library_baseline(workload, device, seed=0, steps=3, with_infer=True):
    """What the REFERENCE does on this GPU with ``device = "cuda:0"`` (cellulus/train.py:60-62: the model is moved to the
    device and every ``nn.ConvNd`` runs in the vendor library — MIOpen on ROCm): the oracle's plain-PyTorch model
    (oracle/unet_oracle.py: nn.Conv2d / MaxPool / Upsample / autograd / torch.optim.Adam) moved to the device, one
    warm-up and ``steps`` timed train steps on the workload's batch, and one infer-mode tile (the 2 x 16 noisy forwards
    of cellulus/models/unet.py:73-100 at 512^2).  A baseline like ``cpu_baseline``: measured beside the product, never
    inside it and never inside the timed region of ``value``."""
    import torch

    from oracle import unet_oracle as O

    torch.manual_seed(seed)
    model = O.OracleUNetModel(**workload["model"])
    for _n, layer in model.named_modules():
        if isinstance(layer, torch.nn.modules.conv._ConvNd):
            torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")
    model = model.to(device)
    opt = torch.optim.Adam(model.parameters(), lr=4e-5, weight_decay=0.01)
    full = int(workload["batch"])
    raw = synthetic_raw(full, workload["crop"], seed).to(device)
    anchor, reference = (t.to(device) for t in sample_pairs(full, workload["crop"], workload["kappa"], workload["density"], seed))

    def step():
        return O.train_step(model, opt, raw, anchor, reference, 10.0, 1e-5)

    t0 = time.perf_counter()
    loss0 = step()[0]
    torch.cuda.synchronize(device)
    t_warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    out = dict(value=round(full / dt, 3), unit="crops/s", ms_per_step=round(dt * 1e3, 2), steps=steps,
               warmup_s=round(t_warm, 2), first_loss=round(float(loss0), 4), kind="reference arithmetic in the vendor library",
               what=f"oracle/unet_oracle.py (plain nn.Conv{len(workload['crop'])}d model, autograd, torch.optim.Adam) on {torch.cuda.get_device_name(device)}: "
                    f"torch {torch.__version__}, MIOpen convolutions (float32), batch {full}, the same synthetic crops and pairs; "
                    "solver / kernel names: profiles/r05_gpu_library_baseline_kernels.txt")
    if with_infer and len(workload["crop"]) == 2:
        del opt
        model.zero_grad(set_to_none=True)
        release_device_memory()
        size, n_it = 512, 16
        model.eval()
        model.set_infer(0.01, n_it)
        tile = torch.rand(1, 1, size + 16, size + 16, device=device)
        noise = torch.rand(1, 2 * n_it, 1, size + 16, size + 16, device=device)
        with torch.no_grad():
            model(tile, noise=noise)                      # warm-up
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            emb = model(tile, noise=noise)
            torch.cuda.synchronize(device)
        t_tile = time.perf_counter() - t0
        assert tuple(emb.shape) == (1, 3, size, size)
        out["infer_tile"] = dict(value=round(size * size / t_tile / 1e6, 4), unit="Mpixels/s (embedding stage only)",
                                 ms_per_tile=round(t_tile * 1e3, 1),
                                 what=f"one {size}^2 tile: 2 x {n_it} batch-1 forwards + std_mean, as cellulus/models/unet.py:73-100 runs them")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="train2d", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gpu-library-baseline", action="store_true")
    ap.add_argument("--no-infer", action="store_true")
    ap.add_argument("--no-train3d", action="store_true")
    ap.add_argument("--no-train-e2e", action="store_true")
    ap.add_argument("--e2e-iterations", type=int, default=140, help="iterations of the real train() behind `train_e2e`")
    ap.add_argument("--infer-samples", type=int, default=16,
                    help="samples PER RANK of the sharded infer() runs behind `infer` when --gpus > 1")
    ap.add_argument("--streams", type=int, default=0, choices=[0, 1, 2],
                    help="0 (default): the product's default, two half batches on two streams per GPU (CLX_STREAMS "
                         "unset).  1: one stream — every kernel alone on the device, the run the roofline numbers and "
                         "the PMC profiles are taken from")
    ap.add_argument("--precision", default="both", choices=["f32", "f32x3bf16", "both"],
                    help="both (default): the line in the product's default arithmetic (f32x3bf16: the plain products on the "
                         "bf16 matrix cores from an exact three-way split of the float32 operands, csrc/gemm_sp.hip), then the "
                         "2-D workload and the inference tile AGAIN with CLX_PRECISION=f32 (float32 MFMA everywhere, the "
                         "arithmetic of rounds 1-5) as the objects `train2d_f32_mfma` / `infer_f32_mfma`; f32 / f32x3bf16: "
                         "the whole line in that precision only")
    args = ap.parse_args()
    if args.streams:
        os.environ["CLX_STREAMS"] = str(args.streams)
    if args.precision != "both":
        os.environ["CLX_PRECISION"] = args.precision

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch

    from cellulus_amd import parallel

    rank, world, local_rank = parallel.init_from_env()
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path exists)")
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    from cellulus_amd.models.plan import precision_name

    line_precision = precision_name()
    res = run_workload(args.workload, args, rank, world, device)
    res_f32 = infer_f32 = None
    if args.precision == "both" and line_precision != "f32" and args.workload == "train2d":
        keep_env = os.environ.get("CLX_PRECISION")
        os.environ["CLX_PRECISION"] = "f32"
        release_device_memory()
        try:
            res_f32 = run_workload(args.workload, args, rank, world, device)
            if world == 1 and not args.no_infer:
                from bench_infer import infer_bench

                release_device_memory()
                infer_f32 = infer_bench(device, with_cpu=False, with_e2e=True, with_streaming=False)
        finally:
            if keep_env is None:
                del os.environ["CLX_PRECISION"]
            else:
                os.environ["CLX_PRECISION"] = keep_env
    res3d = None
    if args.workload == "train2d" and not args.no_train3d:
        release_device_memory()
        res3d = run_workload("train3d", args, rank, world, device)
    # ---- the other half of the metric and the real train(), on every rank when there are several
    infer_obj = e2e_obj = None
    if world > 1:
        # (no try/except here: a rank that fails must EXIT, so that the launcher ends the ranks waiting in a barrier)
        if not args.no_infer:
            release_device_memory()
            from bench_infer import infer_sharded

            infer_obj = infer_sharded(device, rank, world, samples_per_rank=args.infer_samples)
        if not args.no_train_e2e and args.workload in ("train2d", "train3d"):
            release_device_memory()
            e2e_obj = train_e2e(args.workload, device, iterations=args.e2e_iterations, rank=rank, world=world)
        torch.distributed.barrier()
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    out = {
        "metric": "train crops/sec (U-Net fwd+bwd + OCE loss + Adam)",
        "value": res["value"],
        "unit": "crops/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
    }
    out.update({k: v for k, v in res.items() if k not in out})
    if line_precision == "f32x3bf16":
        out["dtype"] = "f32 (exact 3xbf16 operand split, 6 products, f32 accumulate)"
        out["dtype_note"] = ("results and every tensor in HBM are float32; the plain products (1x1 layers, the transform-domain "
                             "products of the 2-D Winograd layers: forward, data gradient, weight gradient) multiply an exact "
                             "three-way bfloat16 split of their float32 operands on the bf16 matrix cores, six products per "
                             "float32 product, float32 accumulation (roofline peak of those kernels: 2516.6 / 6 = 419.4 TFLOP/s "
                             "f32-equivalent); |HIP - float64 oracle| at a trained network's output scale 7.2e-5, the float32 "
                             "MFMA kernels' 7.3e-5 (profiles/r06_parity_trained_scale_2d*.txt); CLX_PRECISION=f32 = float32 MFMA "
                             "everywhere: the objects train2d_f32_mfma / infer_f32_mfma of this line")
    if res_f32 is not None:
        out["train2d_f32_mfma"] = dict(
            metric="train crops/sec, the same workload with CLX_PRECISION=f32 (float32 MFMA everywhere: the arithmetic of "
                   "rounds 1-5)", dtype="f32", steps=args.steps, warmup=args.warmup, **res_f32)
    if infer_f32 is not None:
        out["infer_f32_mfma"] = dict(infer_f32, dtype="f32",
                                     metric=infer_f32["metric"] + " with CLX_PRECISION=f32 (float32 MFMA) on the embedding network")
    if res3d is not None:
        out["train3d"] = dict(metric="train crops/sec, BASELINE configs[3]", steps=args.steps, warmup=args.warmup,
                              **res3d)
    if infer_obj is not None:
        out["infer"] = infer_obj
    if e2e_obj is not None:
        out["train_e2e"] = e2e_obj
    if world == 1 and not args.no_infer:
        try:
            release_device_memory()
            from bench_infer import infer_bench

            out["infer"] = infer_bench(device)
        except Exception as e:  # the train line must survive an inference-side failure
            out["infer"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_train_e2e and args.workload in ("train2d", "train3d"):
        try:
            release_device_memory()
            out["train_e2e"] = train_e2e(args.workload, device, iterations=args.e2e_iterations)
            # 8 loader processes are the reference's default; the same run with 16 says whether the host side (6 ms per
            # crop per process since the pair offsets come from libclx's restatement of numpy's stream) bounds it
            more = train_e2e(args.workload, device, iterations=args.e2e_iterations, workers=16)
            out["train_e2e"]["with_16_loader_procs"] = {k: more[k] for k in ("value", "unit", "ms_per_iteration", "loader_procs")}
        except Exception as e:
            out["train_e2e"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(WORKLOADS[args.workload], device, seed=0)
        except Exception as e:
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    # (--no-cpu-baseline switches BOTH baselines off: the profiling scripts pass it to keep foreign kernels out of their traces)
    if world == 1 and not args.no_gpu_library_baseline and not args.no_cpu_baseline and args.workload in ("train2d", "train3d"):
        try:
            release_device_memory()
            out["gpu_library_baseline"] = gpu_library_baseline(WORKLOADS[args.workload], device)
        except Exception as e:
            out["gpu_library_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
