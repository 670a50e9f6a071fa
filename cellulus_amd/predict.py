"""Tiled embedding inference — drop-in for ``cellulus/predict.py:9-142``.

The reference drives gunpowder (ZarrSource -> Normalize -> Pad(reflect) ->
torch Predict -> ZarrWrite -> Scan).  gunpowder is absent, so the scan is
restated: the raw image is reflect-padded by the network context
(``(crop - out) // 2`` = 8), tiles of ``crop_size`` are visited with a stride of
the output tile, the last tile of every axis is shifted back inside the
image (overlaps are overwritten by the later tile, as gunpowder's Scan does),
and every tile is the (mean, std) of ``2 * num_infer_iterations`` salt/pepper
noised forwards (``UNetModel.infer_on_device``).  The reference's dry-run forward on
a zero tile (predict.py:32-39, run in infer mode, hence ``2 * num_infer_iterations``
``torch.rand`` draws before the first tile) is replayed as draws, so that a seeded
run sees the reference's noise.  Output: float64 zarr dataset
``(S, D+1, *spatial)`` with ``axis_names`` / ``resolution`` / ``offset``.

Samples are independent units: under torch.distributed every rank predicts
a contiguous block of samples, no collective.
"""

import itertools

import numpy as np
import torch

from . import _clx, parallel
from .configs.inference_config import InferenceConfig
from .datasets.meta_data import DatasetMetaData
from .datasets.zarr_dataset import default_normalization_factor
from .models.plan import build_topology
from .utils import zarr_io


def tile_offsets(size, tile):
    """Start offsets of output tiles along one axis: stride = tile, last one shifted inside."""
    if size < tile:
        raise RuntimeError(
            f"image extent {size} is smaller than the network's output tile {tile}: reduce crop_size")
    offs = list(range(0, size - tile + 1, tile))
    if offs[-1] + tile < size:
        offs.append(size - tile)
    return offs


class NoisePrefetcher:
    """The uniform random numbers of the infer-mode forward (unet.py:81: one ``torch.rand`` per noisy
    copy on the CPU generator), drawn by a background thread one tile ahead of the GPU.

    The calls are the reference's, in the reference's order, so a seeded run reproduces the same
    numbers and leaves the generator in the same state:

    * first the ``dry_run`` draws.  The reference finds the output shape by calling the model on a
      zero tile AFTER ``set_infer`` (predict.py:21-39), i.e. it runs the noise loop of unet.py:75-88
      once before the scan: ``2 * num_infer_iterations`` calls of ``torch.rand(1, C, *crop)`` whose
      numbers never reach an image.  Here the shape comes from the launch plan, so the draws are made
      and discarded;
    * then tile after tile, copy after copy, each ``torch.rand(1, C, *crop)``.

    What changes is WHEN they are drawn: while the previous tile's forwards run, into pinned memory,
    so that neither the draw (tens of ms per 512^2 tile) nor the upload sits between two tiles'
    kernels."""

    def __init__(self, num_tiles, copies, tile_shape, depth=2, dry_run=0):
        import queue
        import threading

        self.q = queue.Queue(maxsize=depth)
        self.remaining = num_tiles
        pin = torch.cuda.is_available()

        def work():
            try:
                if dry_run:
                    scratch = torch.empty(tuple(tile_shape), dtype=torch.float32)
                    for _ in range(dry_run):
                        torch.rand(*tile_shape, out=scratch)
                for _ in range(num_tiles):
                    buf = torch.empty((copies,) + tuple(tile_shape), dtype=torch.float32, pin_memory=pin)
                    for t in range(copies):
                        torch.rand(*tile_shape, out=buf[t])
                    self.q.put(buf)
            except BaseException as e:          # surfaces in the consumer
                self.q.put(e)

        self.thread = threading.Thread(target=work, name="clx-noise", daemon=True)
        self.thread.start()

    def next(self):
        assert self.remaining > 0, "more tiles requested than announced"
        self.remaining -= 1
        item = self.q.get()
        if isinstance(item, BaseException):
            raise item
        return item

    def finish(self):
        """Blocks until every announced draw has happened (the generator state is then final)."""
        while self.remaining > 0:
            self.next()
        self.thread.join()


class PredictScan:
    """The tile scan of one sample (predict.py:28-135 restated): reflect-pad by the network
    context, visit tiles of ``crop_size`` with the stride of the output tile, (mean, std) of the
    noisy forwards per tile.  ``predict_sample`` returns the (D+1, *spatial) float32 result on the
    device; the float64 cast happens where it is written."""

    def __init__(self, model, inference_config, meta, normalization_factor, raw_dtype, device):
        nd = meta.num_spatial_dims
        self.model, self.device, self.nd = model, device, nd
        self.crop = tuple(int(c) for c in inference_config.crop_size)
        if len(self.crop) != nd:
            raise ValueError(f"crop_size must have {nd} entries, got {self.crop}")
        topo = build_topology(model.in_channels, model.out_channels, model.num_fmaps, model.fmap_inc_factor,
                              model.features_in_last_layer, model.downsampling_factors, nd, self.crop)
        self.out_tile = tuple(topo.out_shape[3 - nd:])
        self.context = tuple((c - o) // 2 for c, o in zip(self.crop, self.out_tile))
        self.factor = normalization_factor
        if self.factor is None:
            self.factor = default_normalization_factor(raw_dtype)
        self.spatial = tuple(meta.spatial_array)
        self.offsets = [tile_offsets(s, t) for s, t in zip(self.spatial, self.out_tile)]
        self.pad = [(0, 0)] + [(c, c) for c in self.context]
        self.tiles_per_sample = int(np.prod([len(o) for o in self.offsets]))
        # the tiles PARTITION the image (no last tile shifted back inside): a statistic folded over the tiles' outputs
        # is then the statistic of the image
        self.tiles_partition = all(s % t == 0 for s, t in zip(self.spatial, self.out_tile))
        self.noise = None

    def start_noise(self, num_samples):
        """Announce how many samples will be predicted: their noise is then drawn ahead of the GPU.
        Called once per ``predict()`` / ``infer()``, as the reference's dry-run forward is
        (predict.py:32-39); its ``2 * num_infer_iterations`` draws come first.  With several ranks
        every rank is a process of its own with its own generator and does what a reference process
        restricted to that rank's block of samples would do: dry run, then its tiles.  World size 1
        is the reference's sequence."""
        tile_shape = (1, self.model.in_channels) + self.crop
        copies = 2 * int(self.model.num_infer_iterations)
        self.noise = NoisePrefetcher(num_samples * self.tiles_per_sample, copies, tile_shape, dry_run=copies)

    def predict_sample(self, raw, want_std_minmax=False):
        """(D+1, *spatial) float32 device tensor; with want_std_minmax ALSO a float32[2] device tensor holding the
        minimum and maximum of the std channel — or None where the tiles overlap (a later tile overwrites part of an
        earlier one, whose values must not count)."""
        mm = None
        # (the statistics kernel folds the range into its pass for up to 64 predictions per pixel, i.e. 32 noise
        #  iterations — the reference bounds num_infer_iterations nowhere: beyond that, no range here and the Otsu
        #  threshold finds it with a pass of its own, minmax_on_device)
        if want_std_minmax and self.tiles_partition and 2 * int(self.model.num_infer_iterations) <= _clx.NOISE_MINMAX_MAX_T:
            mm = torch.empty(_clx.NOISE_MINMAX_FLOATS, dtype=torch.float32, device=self.device)     # [0..1] + partials
        first = True
        raw = raw.astype(np.float32) * np.float32(self.factor)             # gp.Normalize
        raw = np.pad(raw, self.pad, mode="reflect")                        # gp.Pad(mode="reflect")
        raw_d = torch.from_numpy(raw).to(self.device)
        result = torch.empty((self.nd + 1,) + self.spatial, dtype=torch.float32, device=self.device)
        for off in itertools.product(*self.offsets):
            in_sl = (slice(None),) + tuple(slice(o, o + c) for o, c in zip(off, self.crop))
            tile = raw_d[in_sl].unsqueeze(0).contiguous()
            rnd = None
            if self.noise is not None:           # (T, 1, C, *crop) -> (1, T, C, *crop)
                buf = self.noise.next()
                rnd = buf.view((1, buf.shape[0]) + tuple(buf.shape[2:]))
            emb = self.model.infer_on_device(tile, noise=rnd, std_minmax=None if mm is None else (mm, first))[0]
            first = False
            out_sl = (slice(None),) + tuple(slice(o, o + t) for o, t in zip(off, self.out_tile))
            result[out_sl] = emb
        return (result, mm) if want_std_minmax else result


def predict(model: torch.nn.Module, inference_config: InferenceConfig, normalization_factor: float) -> None:
    dataset_config = inference_config.dataset_config
    meta = DatasetMetaData.from_dataset_config(dataset_config)
    nd = meta.num_spatial_dims
    device = torch.device(inference_config.device)
    if parallel.world_size() > 1:
        device = torch.device("cuda", torch.cuda.current_device())
    model.set_infer(p_salt_pepper=inference_config.p_salt_pepper,
                    num_infer_iterations=inference_config.num_infer_iterations, device=device)

    raw_ds = zarr_io.open(dataset_config.container_path, "r")[dataset_config.dataset_name]
    f = zarr_io.open(inference_config.prediction_dataset_config.container_path)
    # one creator (create_dataset refuses an existing dataset, as zarr's does; if rank 0 fails every
    # rank raises with it instead of waiting at a barrier; detect.py / segment.py do the same)
    parallel.rank0_first(lambda: f.create_dataset(
        inference_config.prediction_dataset_config.dataset_name,
        shape=(meta.num_samples, nd + 1, *meta.spatial_array),
        dtype=float,
    ))
    ds = f[inference_config.prediction_dataset_config.dataset_name]

    scan = PredictScan(model, inference_config, meta, normalization_factor, raw_ds.dtype, device)
    lo, hi = parallel.shard_range(meta.num_samples)
    scan.start_noise(hi - lo)
    # sample i's copy back to the host and its zarr write happen after sample i+1's kernels are
    # enqueued, so the GPU never waits for the host between samples
    pending = None
    for sample in range(lo, hi):
        result = scan.predict_sample(raw_ds[sample])
        if pending is not None:
            ds[pending[0]] = pending[1].cpu().numpy().astype(np.float64)
        pending = (sample, result)
    if pending is not None:
        ds[pending[0]] = pending[1].cpu().numpy().astype(np.float64)
    scan.noise.finish()

    if parallel.world_size() > 1:
        torch.distributed.barrier()
    if parallel.rank() == 0:
        ds.attrs["axis_names"] = ["s", "c"] + ["t", "z", "y", "x"][-nd:]
        ds.attrs["resolution"] = (1,) * nd
        ds.attrs["offset"] = (0,) * nd
