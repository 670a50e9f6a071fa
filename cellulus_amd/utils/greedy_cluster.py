"""Greedy clustering of embeddings on libclx — drop-in for
``cellulus/utils/greedy_cluster.py`` (``Cluster2d`` lines 5-120, ``Cluster3d`` lines
123-253; selected by ``clustering = "greedy"``, ``cellulus/detect.py:162-192``).

The reference's per-seed host loop (two ``.item()`` syncs and ~10 tensor ops per
seed) runs as one persistent-workgroup kernel (``clx_greedy_cluster``); the
element-wise preparation (coordinate add, seed-map normalisation, foreground
gather) is plain torch on the device in the reference's dtypes: float32 in 2-D
(``prediction.float()``, line 83), the prediction's dtype in 3-D (line 213).
"""

import numpy as np
import torch

from .. import _clx


def _cluster(prediction, fg_mask, nd, dtype, bandwidth, min_object_size, seed_thresh,
             min_unclustered_sum, device):
    device = torch.device(device)
    if device.type != "cuda":
        raise _clx.ClxError("greedy clustering runs on HIP devices only; there is no CPU path")
    pred = torch.from_numpy(np.ascontiguousarray(prediction)).to(device=device, dtype=dtype)
    spatial = tuple(pred.shape[1:])
    # xym / xyzm of the reference: float32 linspace grids, channel 0 = x (last axis)
    grids = []
    for c in range(nd):
        ax = nd - 1 - c
        shape = [1] * nd
        shape[ax] = spatial[ax]
        g = torch.linspace(0, spatial[ax] - 1, spatial[ax], dtype=torch.float32, device=device)
        grids.append(g.view(shape).expand(spatial))
    embeddings = pred[0:nd] + torch.stack(grids, 0)
    seed_map = pred[nd:nd + 1]
    seed_map_min = seed_map.min()
    seed_map_max = seed_map.max()
    seed_map = (seed_map - seed_map_max) / (seed_map_min - seed_map_max)
    mask = torch.from_numpy(np.ascontiguousarray(fg_mask).astype(bool)).to(device)
    emb_m = embeddings[mask[None].expand_as(embeddings)].view(nd, -1).contiguous()
    seed_m = seed_map[mask[None]].contiguous()
    n = int(seed_m.numel())
    instance = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
    result = torch.zeros(2, dtype=torch.int32, device=device)
    if n > 0:
        ws = torch.empty(2 * (n + 16), dtype=torch.uint8, device=device)
        _clx.call("clx_greedy_cluster", _clx.ptr(emb_m), _clx.ptr(seed_m), n, nd,
                  1 if emb_m.dtype == torch.float64 else 0, float(bandwidth), int(min_object_size),
                  float(seed_thresh), int(min_unclustered_sum), _clx.ptr(ws), _clx.ptr(instance),
                  _clx.ptr(result), _clx.stream_ptr(device))
    instance_map = torch.zeros(spatial, dtype=torch.int16)
    if n > 0:
        instance_map[mask.cpu()] = instance[:n].to(torch.int16).cpu()
    return instance_map


class Cluster2d:
    """Greedy clustering of embeddings on 2-D samples (same constructor / cluster() signature
    as the reference)."""

    def __init__(self, width, height, fg_mask, device):
        self.width, self.height = width, height
        self.fg_mask = np.asarray(fg_mask)
        self.device = device

    def cluster(self, prediction, bandwidth, min_object_size, seed_thresh=0.9, min_unclustered_sum=0):
        """prediction: (3, H, W) -> int16 instance map (H, W) as a CPU tensor."""
        prediction = np.asarray(prediction)
        return _cluster(prediction, self.fg_mask[:prediction.shape[1], :prediction.shape[2]], 2,
                        torch.float32, bandwidth, min_object_size, seed_thresh, min_unclustered_sum,
                        self.device)


class Cluster3d:
    """Greedy clustering of embeddings on 3-D samples."""

    def __init__(self, width, height, depth, fg_mask, device):
        self.width, self.height, self.depth = width, height, depth
        self.fg_mask = np.asarray(fg_mask)
        self.device = device

    def cluster(self, prediction, bandwidth, min_object_size, seed_thresh=0.9, min_unclustered_sum=0):
        """prediction: (4, D, H, W) -> int16 instance map (D, H, W) as a CPU tensor."""
        prediction = np.asarray(prediction)
        dtype = torch.float64 if prediction.dtype == np.float64 else torch.float32
        return _cluster(prediction, self.fg_mask, 3, dtype, bandwidth, min_object_size, seed_thresh,
                        min_unclustered_sum, self.device)
