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
