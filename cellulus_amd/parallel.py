"""Data-parallel training over RCCL (one process per GPU).

The reference is single-device; the MI355X build shards
crops across ranks.  The loss is a SUM over pairs (oce_loss.py:58-60), so the
gradient of the global batch is the SUM of the per-rank gradients: one
all-reduce(SUM) over the flat f32 gradient buffer, NO division by world size.
xGMI is point-to-point, so a single large bucket (the whole 38.5 MB gradient at
the benchmark config) is the right granularity for the ring.
"""

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK/WORLD_SIZE/MASTER_* if set; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks: CLX_DIST_BACKEND=gloo lets several ranks share one GPU (RCCL refuses that),
    # CLX_LOCAL_DEVICE pins every rank to one device index
    backend = backend or os.environ.get("CLX_DIST_BACKEND")
    if "CLX_LOCAL_DEVICE" in os.environ:
        local_rank = int(os.environ["CLX_LOCAL_DEVICE"])
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def all_reduce_sum_(flat, async_op=False):
    """In-place SUM all-reduce of a flat buffer (no averaging)."""
    if world_size() == 1:
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)


def broadcast_(flat, src=0):
    """Makes every rank start from rank `src`'s parameters."""
    if world_size() > 1:
        dist.broadcast(flat, src=src)


def shard_range(n, r=None, w=None):
    """Contiguous block partition of n independent units (tiles, samples) over ranks."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(n, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)
