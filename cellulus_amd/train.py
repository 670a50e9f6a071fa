"""Training driver — drop-in for ``cellulus/train.py`` (``train``,
``train_iteration``, ``save_model``, ``save_snapshot``) on libclx.

Same control flow, checkpoint dictionary (``iteration, lowest_loss,
model_state_dict, optim_state_dict, logger_data``), file names
(``models/best_loss.pth``, ``models/<iter:06d>.pth``, ``loss.csv``,
``snapshots.zarr/<iter>/{raw,prediction}``) and console output as the
reference.  What differs is underneath: ``train_iteration`` is one stream of
HIP launches (pack -> U-Net forward -> fused gather/OCE/scatter -> backward ->
[RCCL all-reduce SUM] -> one fused Adam launch) with a single host
synchronisation for the two returned floats.

Launched under ``torch.distributed.run`` (WORLD_SIZE > 1) it trains
data-parallel: each rank draws its own crops, gradients are SUM-all-reduced (the
loss is a sum over pairs, so this equals one process at the global batch);
rank 0 alone logs, checkpoints and snapshots.
"""

import gc
import os

import numpy as np
import torch
from tqdm import tqdm

from . import _clx, parallel
from .criterions import get_loss
from .criterions.oce_loss import OCELoss
from .datasets import get_dataset
from .models import get_model
from .models.unet import UNetModel
from .optim import Adam
from .utils import get_logger
from .utils import zarr_io


def _require_hip_device(device_str):
    device = torch.device(device_str)
    if device.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError(
            f"device={device_str!r}: cellulus_amd runs on HIP devices only ('cuda:N' is HIP device N "
            "on ROCm) and none is usable here; there is no CPU path.")
    return device


def loader_policy(world, num_workers):
    """How the input pipeline of ONE rank uses the host (the reference is one process with
    ``num_workers`` loader processes, train.py:38-44).

    * One rank: the reference's behaviour — ``num_workers`` loader processes, pair coordinates from the
      ``np.random`` stream inside them.  ``CLX_DEVICE_PAIRS=1`` opts into the device sampler.
    * Several ranks share one host: each gets ``cores // local_world`` cores, where ``local_world`` is the number
      of ranks ON THIS HOST (``LOCAL_WORLD_SIZE`` as torch.distributed.run exports it; the global world size when
      it is unset, i.e. one node).  The loader processes are capped to
      that share minus one (the rank's own Python), and the pair coordinates are drawn on the device by
      default (``CLX_DEVICE_PAIRS=0`` keeps the np.random stream): at 8 ranks x 8 crops x 5 steps/s the
      np.random stream alone needed 8 cores PER RANK when numpy drew it (about 1 since libclx restates it), the
      device sampler 2.5 for crops + augmentation."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    local_world = world
    if world > 1 and os.environ.get("LOCAL_WORLD_SIZE", "").isdigit():
        local_world = max(1, min(world, int(os.environ["LOCAL_WORLD_SIZE"])))
    per_rank = max(1, cores // max(local_world, 1))
    env = os.environ.get("CLX_DEVICE_PAIRS")
    procs = int(num_workers)
    if world > 1:
        device_pairs = env != "0"
        cap = max(1, per_rank - 1)
        why = f"{local_world} of {world} ranks share this host's {cores} cores"
        if procs > cap:
            why += f": num_workers {procs} capped to {cap}"
            procs = cap
        why += "; CLX_DEVICE_PAIRS=0 keeps np.random pairs" if device_pairs and env is None else ""
    else:
        device_pairs = env == "1"
        why = "single process: the reference's loader" + (" + CLX_DEVICE_PAIRS=1" if device_pairs else "")
    return dict(loader_procs=procs, host_cores_per_rank=per_rank, device_pairs=device_pairs, why=why)


def train(experiment_config):
    print(experiment_config)
    rank, world, local_rank = parallel.init_from_env()
    is_main = rank == 0

    if is_main and not os.path.exists("models"):
        os.makedirs("models")

    train_config = experiment_config.train_config
    model_config = experiment_config.model_config

    device_str = train_config.device
    if world > 1:
        device_str = f"cuda:{local_rank}"
    device = _require_hip_device(device_str)
    torch.cuda.set_device(device)

    # create train dataset
    train_dataset = get_dataset(
        dataset_config=train_config.train_data_config,
        crop_size=tuple(train_config.crop_size),
        elastic_deform=train_config.elastic_deform,
        control_point_spacing=train_config.control_point_spacing,
        control_point_jitter=train_config.control_point_jitter,
        density=train_config.density,
        kappa=train_config.kappa,
        normalization_factor=experiment_config.normalization_factor,
    )

    # the input pipeline's share of the host: every loader process draws the np.random pair stream of its
    # crops (3 ms of one core per 256^2 crop since libclx restates that stream) next to the zarr reads and the augmentation
    policy = loader_policy(world, train_config.num_workers)
    if is_main:
        print(f"[cellulus_amd] input pipeline: {policy['loader_procs']} loader processes per rank "
              f"({policy['host_cores_per_rank']} host cores per rank, world size {world}), pair coordinates "
              f"drawn {'on the device (clx_sample_pairs)' if policy['device_pairs'] else 'in the loader processes (np.random, the reference stream)'}"
              f" [{policy['why']}]")
    pair_sampler = None
    if policy["device_pairs"]:
        from .datasets.zarr_dataset import DevicePairSampler

        # 2 x B x 150 040 x 2 int64 = 38 MB per step less to sample, pickle and upload
        train_dataset.skip_pairs = True
        pair_sampler = DevicePairSampler(train_dataset, device, seed=torch.initial_seed() + 7919 * rank)

    # create train dataloader (every rank draws its own random crops)
    train_dataloader = torch.utils.data.DataLoader(
        dataset=train_dataset,
        batch_size=train_config.batch_size,
        drop_last=True,
        num_workers=policy["loader_procs"],
        pin_memory=True,
        # batches come back in worker order: a deeper queue per worker absorbs one slow crop (15 MB per batch)
        # (CLX_LOADER_PREFETCH: hosts with a small /dev/shm — loader processes x this many batches live there)
        prefetch_factor=max(1, int(os.environ.get("CLX_LOADER_PREFETCH", "4"))) if policy["loader_procs"] > 0 else None,
        collate_fn=_collate_narrow if max(train_config.crop_size) < 32768 else None,
    )

    # set model
    model = get_model(
        in_channels=train_dataset.get_num_channels(),
        out_channels=train_dataset.get_num_spatial_dims(),
        num_fmaps=model_config.num_fmaps,
        fmap_inc_factor=model_config.fmap_inc_factor,
        features_in_last_layer=model_config.features_in_last_layer,
        downsampling_factors=[tuple(factor) for factor in model_config.downsampling_factors],
        num_spatial_dims=train_dataset.get_num_spatial_dims(),
    )
    model = model.to(device)

    # the pair sampler draws coordinates for an output of extent crop_size - 16 (zarr_dataset.py:94): a
    # network whose output is smaller (deeper, other factors) makes the reference fail in its first
    # iteration with an IndexError out of select_and_add_coordinates — say so before any work is done
    from .models.plan import build_topology

    topo = build_topology(model.in_channels, model.out_channels, model.num_fmaps, model.fmap_inc_factor,
                          model.features_in_last_layer, model.downsampling_factors, model.num_spatial_dims,
                          tuple(train_config.crop_size))
    out_extent = tuple(topo.out_shape[3 - model.num_spatial_dims:])
    if any(o < d for o, d in zip(out_extent, train_dataset.output_shape)):
        raise IndexError(
            f"the network maps crop_size {tuple(train_config.crop_size)} to an output of extent {out_extent}, but "
            f"the pair sampler draws coordinates for {train_dataset.output_shape} (crop_size - 16): choose a "
            "model / crop_size whose output is crop_size - 16")

    # initialize model weights
    if model_config.initialize:
        for _name, layer in model.named_modules():
            if isinstance(layer, torch.nn.modules.conv._ConvNd):
                torch.nn.init.kaiming_normal_(layer.weight, nonlinearity="relu")

    # set loss
    criterion = get_loss(
        regularizer_weight=train_config.regularizer_weight,
        temperature=train_config.temperature,
        density=train_config.density,
        num_spatial_dims=train_dataset.get_num_spatial_dims(),
        device=device,
    )

    # set optimizer: Adam with coupled L2 (train.py:80-82)
    flat, _ = model.flatten_parameters()
    parallel.broadcast_(flat, src=0)
    optimizer = Adam(model.parameters(), lr=train_config.initial_learning_rate, weight_decay=0.01)

    # set logger
    logger = get_logger(keys=["loss", "oce_loss"], title="loss")

    # resume training
    start_iteration = 0
    lowest_loss = 1e6
    epoch_loss = 0
    num_iterations = 0
    if model_config.checkpoint is not None:
        print(f"Resuming model from {model_config.checkpoint}")
        state = torch.load(model_config.checkpoint, map_location=device, weights_only=False)
        start_iteration = state["iteration"] + 1
        lowest_loss = state["lowest_loss"]
        model.load_state_dict(state["model_state_dict"], strict=True)
        optimizer.load_state_dict(state["optim_state_dict"])
        logger.data = state["logger_data"]

    gc_freeze = os.environ.get("CLX_GC_FREEZE", "1") != "0"
    # call `train_iteration`
    for iteration, batch in tqdm(
        zip(range(start_iteration, train_config.max_iterations),
            _DevicePrefetcher(train_dataloader, device, pair_sampler, start_iteration)),
        disable=not is_main,
    ):
        loss, oce_loss, prediction = train_iteration(
            batch, model=model, criterion=criterion, optimizer=optimizer, device=device)
        if iteration == start_iteration + 2 and gc_freeze:
            # everything that lives as long as the run (modules, plans, the loader's machinery: ~3e5 tracked objects) leaves
            # the collector's generations: a generation-2 collection then scans what the loop allocates, not the whole heap
            # (measured in the benchmark's process: 34-97 ms each, an iteration of 100-200 ms once in ~130; CLX_GC_FREEZE=0
            # leaves the collector alone)
            gc.collect()
            gc.freeze()
        if not is_main:
            continue
        print(f"===> loss: {loss:.6f}, oce loss: {oce_loss:.6f}")
        logger.add(key="loss", value=loss)
        logger.add(key="oce_loss", value=oce_loss)
        logger.write()
        logger.plot()

        # Check if lowest loss
        epoch_loss += loss
        num_iterations += 1
        if iteration % train_config.save_best_model_every == 0:
            is_lowest = epoch_loss / (num_iterations) < lowest_loss
            lowest_loss = min(epoch_loss / num_iterations, lowest_loss)
            if is_lowest:
                save_model(_state(iteration, lowest_loss, model, optimizer, logger), iteration, is_lowest)
            epoch_loss = 0
            num_iterations = 0

        # Save model at specific intervals
        if (iteration % train_config.save_model_every == 0
                or iteration == train_config.max_iterations - 1):
            save_model(_state(iteration, lowest_loss, model, optimizer, logger), iteration)

        # Save snapshots at specific intervals
        if iteration % train_config.save_snapshot_every == 0:
            save_snapshot(batch, prediction, iteration)
    if gc_freeze:
        gc.unfreeze()
    if is_main:
        logger.plot(force=True)


def _collate_narrow(samples):
    """The DataLoader's batch assembly with the pair coordinates as int16: they index a crop (< 32768 per axis), and
    at the benchmark configuration the two int64 arrays are 38 of the 40 MB a batch moves through the loader
    processes' shared memory, the pinning thread and the H2D copy.  ``_DevicePrefetcher`` widens them on the device:
    ``train_iteration`` sees the reference's int64 tensors (``cellulus/train.py:166-173``)."""
    raw = torch.from_numpy(np.stack([s[0] for s in samples]))
    anchor = np.stack([s[1] for s in samples])
    reference = np.stack([s[2] for s in samples])
    # (a dataset that yields coordinates outside an int16 keeps its int64 arrays: the bad-coordinate check of the gather
    #  must see the values as they are, not wrapped)
    if min(anchor.min(initial=0), reference.min(initial=0)) >= 0 and max(anchor.max(initial=0), reference.max(initial=0)) < 32768:
        anchor, reference = anchor.astype(np.int16), reference.astype(np.int16)
    return raw, torch.from_numpy(anchor), torch.from_numpy(reference)


class _DevicePrefetcher:
    """Stages batch i+1 (pinned host memory -> HBM, 40 MB at the 2-D configuration: raw plus two
    int64 coordinate arrays) on a copy stream while step i computes, so that the transfer is off
    the step's critical path; the batches come out as device tensors, which
    ``train_iteration``'s ``.to(device)`` passes through."""

    def __init__(self, loader, device, pair_sampler=None, first_step=0):
        self.it = iter(loader)
        self.device = device
        self.stream = torch.cuda.Stream(device)
        self.next = None
        self.pair_sampler = pair_sampler
        self.step = first_step
        self._stage()

    def _stage(self):
        try:
            batch = next(self.it)
        except StopIteration:
            self.next = None
            return
        with torch.cuda.stream(self.stream):
            if self.pair_sampler is not None:            # the loader sent crops only: draw the pairs here
                raw = batch[0].to(self.device, non_blocking=True)
                anchor, reference = self.pair_sampler.sample(raw.shape[0], self.step)
                self.next = (raw, anchor, reference)
            else:
                staged = [t.to(self.device, non_blocking=True) for t in batch]
                # (the loader narrowed the coordinates for the trip: _collate_narrow)
                self.next = tuple(t.to(torch.int64) if t.dtype == torch.int16 else t for t in staged)
            self.step += 1

    def __iter__(self):
        return self

    def __next__(self):
        if self.next is None:
            raise StopIteration
        current = torch.cuda.current_stream(self.device)
        current.wait_stream(self.stream)
        batch = self.next
        for t in batch:
            t.record_stream(current)
        self._stage()
        return batch


def _state(iteration, lowest_loss, model, optimizer, logger):
    return {
        "iteration": iteration,
        "lowest_loss": lowest_loss,
        "model_state_dict": model.state_dict(),
        "optim_state_dict": optimizer.state_dict(),
        "logger_data": logger.data,
    }


def train_iteration(batch, model, criterion, optimizer, device):
    """One optimisation step (train.py:160-180): returns (loss, oce_loss, offsets)."""
    raw, anchor_coordinates, reference_coordinates = batch
    raw, anchor_coordinates, reference_coordinates = (
        raw.to(device, non_blocking=True),
        anchor_coordinates.to(device, non_blocking=True),
        reference_coordinates.to(device, non_blocking=True),
    )
    model.train()

    if isinstance(model, UNetModel) and isinstance(criterion, OCELoss) and model.mode == "train":
        loss, oce_loss, offsets = _fused_step(
            model, criterion, optimizer, raw, anchor_coordinates, reference_coordinates)
        return loss, oce_loss, offsets

    # generic autograd composition (custom model / criterion objects)
    offsets = model(raw)
    embeddings_anchor = model.select_and_add_coordinates(offsets, anchor_coordinates)
    embeddings_reference = model.select_and_add_coordinates(offsets, reference_coordinates)
    loss, oce_loss, regularization_loss = criterion(embeddings_anchor, embeddings_reference)
    optimizer.zero_grad()
    loss.backward()
    if parallel.world_size() > 1:
        for p in model.parameters():
            if p.grad is not None:
                parallel.all_reduce_sum_(p.grad)
    optimizer.step()
    return loss.item(), oce_loss.item(), offsets


def _fused_step(model, criterion, optimizer, raw, anchor, reference):
    """forward -> fused gather + OCE + scatter -> backward -> all-reduce -> Adam, no autograd."""
    if raw.dtype != torch.float32:
        raw = raw.float()
    if anchor.dtype != torch.int64:
        anchor = anchor.long()
    if reference.dtype != torch.int64:
        reference = reference.long()
    anchor = anchor.contiguous()
    reference = reference.contiguous()
    device = raw.device
    model.flatten_parameters()
    params = model._ordered_params()
    grads = model.attach_flat_grads()
    plan = model._plan_for(raw, keep=True)
    plan.pack_weights(params, model._param_version(), need_dgrad=True)
    B = raw.shape[0]
    ND = model.out_channels            # one offset channel per spatial dimension
    if anchor.ndim != 3 or anchor.shape != reference.shape or anchor.shape[0] != B or anchor.shape[2] != ND:
        raise ValueError(f"coordinates must be (B={B}, P, {ND}); got {tuple(anchor.shape)} and "
                         f"{tuple(reference.shape)}")
    sums = torch.empty(4, dtype=torch.float64, device=device)     # loss, oce, reg, bad-coordinate count
    _clx.zero_many(sums)
    geometry = []

    def loss_fn(offsets, lo, hi):
        """gather + OCE + scatter of crops lo..hi-1 on the current stream; adds into `sums` (the loss is a sum over
        pairs: the halves of a two-stream step add up to the batch's)."""
        Z, Y, X = (1, offsets.shape[2], offsets.shape[3]) if ND == 2 else tuple(offsets.shape[2:])
        geometry[:] = [Z, Y, X]
        a, r = anchor[lo:hi], reference[lo:hi]
        if plan.deterministic:
            # CLX_DETERMINISTIC=1: fixed-point scatter of the anchor gradients, loss sums in block order
            doffsets = torch.empty_like(offsets)
            need = int(_clx.load().clx_oce_pairs_det_scratch_bytes(hi - lo, ND, Z * Y * X))
            scratch = getattr(model, "_det_loss_scratch", None)
            if scratch is None or scratch.numel() < need or scratch.device != device:
                scratch = model._det_loss_scratch = torch.empty(need, dtype=torch.uint8, device=device)
            _clx.call("clx_oce_pairs_fused_det", _clx.ptr(offsets), _clx.ptr(a), _clx.ptr(r),
                      _clx.ptr(doffsets), _clx.ptr(sums), hi - lo, a.shape[1], ND, Z, Y, X,
                      float(criterion.temperature), float(criterion.regularization_weight), _clx.ptr(scratch),
                      _clx.stream_ptr(device))
        else:
            doffsets = torch.empty_like(offsets)
            _clx.zero_many(doffsets)
            _clx.call("clx_oce_pairs_fused", _clx.ptr(offsets), _clx.ptr(a), _clx.ptr(r),
                      _clx.ptr(doffsets), _clx.ptr(sums), hi - lo, a.shape[1], ND, Z, Y, X,
                      float(criterion.temperature), float(criterion.regularization_weight),
                      _clx.stream_ptr(device))
        return doffsets

    early = []
    buckets = None
    if parallel.world_size() > 1 and parallel.bucket_bytes() > 0:
        # gradient buckets go out while the rest of the backward pass runs (parallel.GradientBuckets)
        buckets = parallel.GradientBuckets(model._flat_grad, grads)

    def after_loss():
        """on the caller's stream, once the loss of every crop is enqueued (or waited for) there"""
        if buckets is not None:
            buckets.add(sums)
        if parallel.world_size() == 1 and os.environ.get("CLX_LOSS_EARLY", "1") != "0":
            # One process: the sums are final HERE, before the backward pass.  They leave on a side stream into
            # pinned memory now, and the host reads them after it has enqueued backward, update and packing —
            # by then the copy is 25 ms old, so train_iteration returns while the device still works and the
            # next call's Python runs under this step's kernels.  (Several ranks: the sums are all-reduced with
            # the gradients and read at the end, below.)
            early.append(_early_readback(model, sums, device))

    if buckets is not None:
        offsets = plan.train_pass(raw, params, grads, model._flat_grad, loss_fn, after_loss,
                                  on_layer_done=lambda i: buckets.params_done(2 * i, 2 * i + 1))
        buckets.finish()
        model._last_bucket_ranges = list(buckets.issued)      # (lo, hi) element ranges, in issue order
    else:
        offsets = plan.train_pass(raw, params, grads, model._flat_grad, loss_fn, after_loss)
        if parallel.world_size() > 1:
            parallel.all_reduce_sum_(model._flat_grad)
            parallel.all_reduce_sum_(sums)
            model._last_bucket_ranges = [(0, model._flat_grad.numel())]
    Z, Y, X = geometry
    early = early[0] if early else None
    if early is None and parallel.world_size() > 1 and os.environ.get("CLX_LOSS_EARLY", "1") != "0":
        # several ranks: the reduced sums exist once the last bucket is in; the copy then runs beside the update
        # and the packing, and the host returns while those still execute
        early = _early_readback(model, sums, device)
    from .criterions.oce_loss import raise_on_bad_coordinates
    from .optim import Adam as ClxAdam

    if isinstance(optimizer, ClxAdam) and os.environ.get("CLX_ADAM_EARLY", "1") != "0":
        # The reference raises IndexError out of the coordinate indexing, i.e. before optimizer.step()
        # (cellulus/train.py:171-179).  Looking at the bad-coordinate count is the step's one host
        # synchronisation; with the update enqueued first — guarded on the device by that very count — and the
        # next step's packed weights right behind it, the device has work while the host waits, reads, and
        # walks back into the next call (0.27 + 0.3 ms per step at the benchmark configuration otherwise idle).
        optimizer.step(guard=sums[3:4])
        plan.pack_weights(params, model._param_version(), need_dgrad=True)
        if early is not None:
            early[1].synchronize()
            host = early[0].clone()
        else:
            host = sums.cpu()
        bad = int(host[3].item())
        if bad:
            optimizer.undo_step()              # the kernel did nothing; take the step counters back too
        raise_on_bad_coordinates(bad, (Z, Y, X)[3 - ND:])
    else:
        host = sums.cpu()
        raise_on_bad_coordinates(int(host[3].item()), (Z, Y, X)[3 - ND:])
        optimizer.step()
    host = host.to(torch.float32)
    return host[0].item(), host[1].item(), offsets


def _early_readback(model, sums, device):
    """sums (device, float64[4]) -> the model's pinned host buffer through a side stream that waits for the
    loss kernel only; returns (host buffer, event recorded behind the copy)."""
    cache = getattr(model, "_early_readback_cache", None)
    if cache is None or cache[0] != device:
        cache = (device, torch.empty(4, dtype=torch.float64).pin_memory(), torch.cuda.Stream(device=device))
        model._early_readback_cache = cache
    _dev, host, side = cache
    produced = torch.cuda.Event()
    produced.record(torch.cuda.current_stream(device))
    done = torch.cuda.Event()
    with torch.cuda.stream(side):
        side.wait_event(produced)
        host.copy_(sums, non_blocking=True)
        sums.record_stream(side)
        done.record(side)
    return host, done


def save_model(state, iteration, is_lowest=False):
    if is_lowest:
        file_name = os.path.join("models", "best_loss.pth")
        torch.save(state, file_name)
        print(f"Best model weights saved at iteration {iteration}")
    else:
        file_name = os.path.join("models", str(iteration).zfill(6) + ".pth")
        torch.save(state, file_name)
        print(f"Checkpoint saved at iteration {iteration}")


def save_snapshot(batch, prediction, iteration):
    """snapshots.zarr/<iteration>/{raw,prediction} with mean-subtracted offsets (train.py:194-224)."""
    raw, anchor_coordinates, reference_coordinates = batch
    num_spatial_dims = len(raw.shape) - 2
    axis_names = ["s", "c"] + ["t", "z", "y", "x"][-num_spatial_dims:]
    prediction_offset = tuple(
        (a - b) / 2
        for a, b in zip(raw.shape[-num_spatial_dims:], prediction.shape[-num_spatial_dims:]))
    f = zarr_io.open("snapshots.zarr", "a")
    f[f"{iteration}/raw"] = raw.detach().cpu().numpy()
    f[f"{iteration}/raw"].attrs["axis_names"] = axis_names
    f[f"{iteration}/raw"].attrs["resolution"] = [1] * num_spatial_dims
    prediction_cpu = prediction.detach().cpu().numpy()
    flat = np.reshape(prediction_cpu, (prediction_cpu.shape[0], prediction_cpu.shape[1], -1))
    mean_prediction = np.mean(flat, 2)
    prediction_cpu -= mean_prediction[(...,) + (np.newaxis,) * num_spatial_dims]
    f[f"{iteration}/prediction"] = prediction_cpu
    f[f"{iteration}/prediction"].attrs["axis_names"] = axis_names
    f[f"{iteration}/prediction"].attrs["offset"] = prediction_offset
    f[f"{iteration}/prediction"].attrs["resolution"] = [1] * num_spatial_dims
