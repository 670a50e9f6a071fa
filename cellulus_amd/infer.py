"""Inference driver — drop-in for ``cellulus/infer.py:16-80``: predict -> detect ->
segment -> evaluate over zarr datasets, same defaults for bandwidth and min_size."""

import os

import numpy as np
import torch

from . import parallel
from .datasets.meta_data import DatasetMetaData
from .detect import detect
from .evaluate import evaluate
from .models import get_model
from .predict import predict
from .segment import segment
from .train import _require_hip_device


def _stages_chain(cfg):
    """True if detect reads what predict writes and segment reads what detect writes."""
    p, d, s = cfg.prediction_dataset_config, cfg.detection_dataset_config, cfg.segmentation_dataset_config
    return (p is not None and d is not None and s is not None
            and d.container_path == p.container_path and d.secondary_dataset_name == p.dataset_name
            and s.container_path == d.container_path and s.secondary_dataset_name == d.dataset_name)


def fused_stages(model, inference_config, normalization_factor, device):
    """The three stages of ``infer.py:69-77`` sample by sample instead of dataset by dataset: the
    embeddings go from the U-Net to the mean-shift and on to the post-processing in HBM, the
    zarr datasets — the same ones, with the same contents — are written by a background thread
    while the next stage computes.  What the staged path reads back from disk is reproduced
    exactly: float32 embeddings widened to float64 (on the host for the datasets; in registers by the
    kernels that consume them on the device: clx_histogram_f32, clx_ms_prepare_f32 — the widening is exact,
    so thresholds, points and labels are the staged path's bits without a float64 copy in HBM), label maps
    through their uint16 storage type.
    Every stage draws from its own generator in the staged order too (torch for the noise,
    numpy for the mean-shift sub-sampling), so interleaving them changes no random number."""
    from concurrent.futures import ThreadPoolExecutor

    from .detect import _create, _labels_to_host, detect_sample
    from .predict import PredictScan
    from .segment import segment_sample
    from .utils import zarr_io

    dataset_config = inference_config.dataset_config
    meta = DatasetMetaData.from_dataset_config(dataset_config)
    nd = meta.num_spatial_dims
    spatial = tuple(meta.spatial_array)
    model.set_infer(p_salt_pepper=inference_config.p_salt_pepper,
                    num_infer_iterations=inference_config.num_infer_iterations, device=device)
    raw_ds = zarr_io.open(dataset_config.container_path, "r")[dataset_config.dataset_name]
    f = zarr_io.open(inference_config.prediction_dataset_config.container_path)
    names = dict(emb=inference_config.prediction_dataset_config.dataset_name,
                 det=inference_config.detection_dataset_config.dataset_name, bin="binary-segmentation",
                 cen="centered-embeddings", seg=inference_config.segmentation_dataset_config.dataset_name)
    def create():                    # one creator (create_dataset refuses an existing dataset)
        _create(f, names["emb"], (meta.num_samples, nd + 1, *spatial), float, nd)
        _create(f, names["det"], (meta.num_samples, inference_config.num_bandwidths, *spatial), np.uint16, nd)
        _create(f, names["bin"], (meta.num_samples, 1, *spatial), np.uint16, nd)
        _create(f, names["cen"], (meta.num_samples, nd + 1, *spatial), float, nd)
        _create(f, names["seg"], (meta.num_samples, inference_config.num_bandwidths, *spatial), np.uint16, nd)

    parallel.rank0_first(create)     # every rank raises with rank 0 instead of waiting at a barrier
    ds_emb, ds_det, ds_bin, ds_cen, ds_seg = (f[names[k]] for k in ("emb", "det", "bin", "cen", "seg"))
    # samples are independent units: every rank takes a contiguous block, no collective on the data path
    lo, hi = parallel.shard_range(meta.num_samples)

    scan = PredictScan(model, inference_config, meta, normalization_factor, raw_ds.dtype, device)
    scan.start_noise(hi - lo)               # the noise of sample i+1 is drawn while sample i computes
    pending = []
    with ThreadPoolExecutor(max_workers=1) as writer:
        def write(ds, key, value):
            def job():
                ds[key] = value
            pending.append(writer.submit(job))

        # Two streams: the embeddings of sample i+1 are enqueued on the main stream BEFORE sample i's
        # detection / post-processing (small kernels with host round trips in between: foreground
        # count, centre de-duplication) run on a second stream — the U-Net forwards never wait for
        # the host.  Every stage still sees its samples in order and draws from its own generator.
        main = torch.cuda.current_stream(device)
        post = torch.cuda.Stream(device)

        def enqueue_predict(sample):
            raw = raw_ds[sample]
            # float32 (D+1, *spatial) + the std channel's (min, max) where the tiles partition the image
            emb, std_minmax = scan.predict_sample(raw, want_std_minmax=True)
            done = torch.cuda.Event()
            done.record(main)
            return raw, (emb, std_minmax), done

        nxt = enqueue_predict(lo) if hi > lo else None
        for sample in range(lo, hi):
            raw, (emb_d, std_minmax), done = nxt
            nxt = enqueue_predict(sample + 1) if sample + 1 < hi else None
            with torch.cuda.stream(post):
                post.wait_event(done)
                emb_d.record_stream(post)
                if std_minmax is not None:
                    std_minmax.record_stream(post)
                embeddings = emb_d.cpu().numpy().astype(np.float64)       # what predict() writes (predict.py:104-112)
                write(ds_emb, sample, embeddings)

                def emit(kind, index, value, sample=sample):
                    if kind == "binary":
                        write(ds_bin, (sample, 0, Ellipsis), value)
                    elif kind == "centered":
                        write(ds_cen, sample, value)
                    else:
                        write(ds_det, (sample, index, Ellipsis), _labels_to_host(value))

                detections = detect_sample(embeddings, inference_config, nd, device, sample, emb_d=emb_d,
                                           std_minmax=std_minmax, emit=emit)
                for bandwidth_factor, labels in enumerate(detections):
                    # through the uint16 storage type, as segment() reads it back
                    seg_d = (labels.to(device=device, dtype=torch.int32) & 0xFFFF).contiguous()
                    out = segment_sample(seg_d, raw[0], inference_config, device)
                    write(ds_seg, (sample, bandwidth_factor, Ellipsis), out.cpu().numpy())
                post.synchronize()
            while len(pending) > 16:                              # bound the host copies in flight
                pending.pop(0).result()
        for job in pending:
            job.result()
    scan.noise.finish()
    if parallel.world_size() > 1:
        torch.distributed.barrier()


def infer(experiment_config):
    print(experiment_config)
    rank, world, local_rank = parallel.init_from_env()

    inference_config = experiment_config.inference_config
    normalization_factor = experiment_config.normalization_factor
    model_config = experiment_config.model_config

    dataset_meta_data = DatasetMetaData.from_dataset_config(inference_config.dataset_config)

    if inference_config.bandwidth is None:
        inference_config.bandwidth = 0.5 * experiment_config.object_size

    if inference_config.min_size is None:
        if dataset_meta_data.num_spatial_dims == 2:
            inference_config.min_size = int(0.1 * np.pi * (experiment_config.object_size ** 2) / 4)
        elif dataset_meta_data.num_spatial_dims == 3:
            inference_config.min_size = int(
                0.1 * 4.0 / 3.0 * np.pi * (experiment_config.object_size ** 3) / 8)

    # set model
    model = get_model(
        in_channels=dataset_meta_data.num_channels,
        out_channels=dataset_meta_data.num_spatial_dims,
        num_fmaps=model_config.num_fmaps,
        fmap_inc_factor=model_config.fmap_inc_factor,
        features_in_last_layer=model_config.features_in_last_layer,
        downsampling_factors=[tuple(factor) for factor in model_config.downsampling_factors],
        num_spatial_dims=dataset_meta_data.num_spatial_dims,
    )

    # set device
    device_str = inference_config.device if world == 1 else f"cuda:{local_rank}"
    device = _require_hip_device(device_str)
    torch.cuda.set_device(device)
    model = model.to(device)

    # load checkpoint
    if model_config.checkpoint is not None and os.path.exists(model_config.checkpoint):
        state = torch.load(model_config.checkpoint, map_location=device, weights_only=False)
        model.load_state_dict(state["model_state_dict"], strict=True)
    else:
        assert False, f"Model weights do not exist at this location :{model_config.checkpoint}!"

    # set in eval mode
    model.eval()

    if _stages_chain(inference_config) and os.environ.get("CLX_FUSED_INFER", "1") != "0":
        # predict -> detect -> segment per sample with the hand-off in device memory
        fused_stages(model, inference_config, normalization_factor, device)
    else:
        # get predicted embeddings...
        if inference_config.prediction_dataset_config is not None:
            predict(model, inference_config, normalization_factor)
        # ...turn them into a detection ...
        if inference_config.detection_dataset_config is not None:
            detect(inference_config)
        # ...and post-process the detection to obtain an instance segmentation
        if inference_config.segmentation_dataset_config is not None:
            segment(inference_config)
    # ...and evaluate if ground-truth exists
    if inference_config.evaluation_dataset_config is not None and rank == 0:
        if world > 1:
            torch.distributed.barrier()
        evaluate(inference_config)
    elif inference_config.evaluation_dataset_config is not None and world > 1:
        torch.distributed.barrier()
