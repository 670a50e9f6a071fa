"""Data-parallel training over RCCL (one process per GPU).

The reference is single-device; the MI355X build shards
crops across ranks.  The loss is a SUM over pairs (oce_loss.py:58-60), so the
gradient of the global batch is the SUM of the per-rank gradients: one
all-reduce(SUM) over the flat f32 gradient buffer, NO division by world size.
The backward pass completes that buffer from its tail to its head (the flat order
is the forward order of the layers), so the reduction is issued in a few large
contiguous buckets as soon as their layers are done and runs on RCCL's stream
under the rest of the backward pass (``GradientBuckets``).  xGMI is point-to-point
and the ring is per-link bound: buckets are few and large (>= CLX_GRAD_BUCKET_MB,
default 4 MB; at the benchmark config the 21 MB 768x768x3x3 layer is one bucket),
not the 25 MB / many-small-tensors scheme of an NVSwitch-tuned DDP.
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
    if "CLX_LOCAL_DEVICE" in os.environ and world > 1 and torch.cuda.is_available():
        # several ranks on one device: a rank may not let torch's caching allocator keep more than its share (blocks cached
        # from an earlier phase of the run count against the others: the eight-rank rehearsal of BASELINE configs[2] peaks
        # at 26 GB per rank of live tensors and, uncapped, ran one GPU of 288 GB out of memory now and then).  At the cap
        # the allocator gives its unused blocks back and retries before it reports out-of-memory
        share = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        torch.cuda.set_per_process_memory_fraction(min(1.0, 0.92 / share), local_rank)
    return rank, world, local_rank


def ranks_sharing_device():
    """Ranks that run on THIS rank's device: 1, or — under the test hook CLX_LOCAL_DEVICE, which pins every rank of the
    host to one device (the 8-rank dress rehearsal of BASELINE configs[2] on one GPU) — the ranks on this host."""
    if "CLX_LOCAL_DEVICE" not in os.environ:
        return 1
    return max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))


def free_device_memory(device):
    """Bytes this process may still take on `device`: what the device has free plus what torch's allocator holds
    unused, capped by the rank's share of the device where several ranks run on it (they start together: what is free
    NOW says nothing about what the others are about to take)."""
    free = torch.cuda.mem_get_info(device)[0] + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
    share = ranks_sharing_device()
    if share > 1:
        total = torch.cuda.get_device_properties(device).total_memory
        free = min(free, total // share - torch.cuda.memory_allocated(device))
    return max(0, free)


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def all_reduce_sum_(flat, async_op=False):
    """In-place SUM all-reduce of a flat buffer (no averaging)."""
    if world_size() == 1:
        return None
    return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)


def bucket_bytes():
    """Smallest gradient bucket issued before the backward pass has finished; 0 = one all-reduce at the end."""
    return int(float(os.environ.get("CLX_GRAD_BUCKET_MB", "4")) * (1 << 20))


class GradientBuckets:
    """SUM all-reduce of the flat gradient, overlapped with the backward pass.

    ``views`` are the per-parameter views of ``flat`` in buffer order, without gaps.
    ``params_done(i, ...)`` marks parameters whose gradient kernels are enqueued on the current
    stream; whenever the completed SUFFIX of the buffer has grown by ``min_bytes`` the range is
    all-reduced asynchronously (the collective waits for the current stream at issue time and then
    runs on the backend's own stream).  ``finish()`` reduces what is left and makes the current
    stream wait for every bucket.  Every rank issues the same ranges in the same order: they
    depend on the launch plan only."""

    def __init__(self, flat, views, min_bytes=None):
        self.flat = flat
        self.lo, off = [], 0
        for v in views:
            assert v.data_ptr() == flat.data_ptr() + off * flat.element_size(), "views must tile the flat buffer"
            self.lo.append(off)
            off += v.numel()
        assert off == flat.numel()
        self.done = [False] * len(views)
        self.cursor = len(views)                 # parameters [cursor:] are complete
        self.issued_lo = flat.numel()            # elements [issued_lo:] are already being reduced
        self.min_elems = max(1, (bucket_bytes() if min_bytes is None else min_bytes) // flat.element_size())
        self.works, self.issued = [], []

    def _issue(self, lo):
        if lo < self.issued_lo:
            self.works.append(all_reduce_sum_(self.flat[lo:self.issued_lo], async_op=True))
            self.issued.append((lo, self.issued_lo))
            self.issued_lo = lo

    def add(self, tensor):
        """An extra tensor (the loss sums) reduced alongside."""
        self.works.append(all_reduce_sum_(tensor, async_op=True))

    def params_done(self, *indices):
        for i in indices:
            self.done[i] = True
        while self.cursor > 0 and self.done[self.cursor - 1]:
            self.cursor -= 1
        lo = self.lo[self.cursor] if self.cursor < len(self.lo) else self.flat.numel()
        if self.issued_lo - lo >= self.min_elems:
            self._issue(lo)

    def finish(self):
        self._issue(0)
        for w in self.works:
            if w is not None:
                w.wait()
        self.works = []


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


def rank0_first(fn):
    """Runs ``fn`` on rank 0 only, then lets every rank continue — or makes every rank raise TOGETHER
    if rank 0 failed (e.g. ``create_dataset`` over an existing dataset): a bare ``barrier()`` after a
    rank-0-only step would leave the other ranks waiting until the collective times out."""
    if world_size() == 1:
        fn()
        return
    error = None
    if rank() == 0:
        try:
            fn()
        except BaseException as e:                       # noqa: B902 - re-raised below on every rank
            error = e
    box = [None if error is None else f"{type(error).__name__}: {error}"]
    dist.broadcast_object_list(box, src=0)
    if rank() == 0 and error is not None:
        raise error
    if box[0] is not None:
        raise RuntimeError(f"rank 0 failed: {box[0]}")
