"""Mean-shift instance clustering on libclx — drop-in for
``cellulus/utils/mean_shift.py`` (``mean_shift_segmentation``,
``segment_with_meanshift``, ``AnchorMeanshift``), i.e. for
``sklearn.cluster.MeanShift(bandwidth, cluster_all=False, seeds).fit(X_red)``
followed by ``.predict(X)``, all in float64.

Device side: coordinate add + foreground compaction (raster order), the
per-seed flat-kernel iterations, the nearest-centre assignment.  Host side
(small, sequential by definition): the reference's ``np.random.rand``
sub-sampling mask (same global RNG call, so seeding numpy reproduces it) and
sklearn's sort + greedy de-duplication of converged centres.
"""

import numpy as np
import torch

from .. import _clx

MAX_ITER = 300   # sklearn.cluster.MeanShift default
GRID_MIN_POINTS = 2048   # below this a brute-force sweep of the (L2-resident) fit set is cheaper


def _bucket(fit, bandwidth):
    """Bucket the fit points into uniform-grid cells of edge just above the bandwidth
    (clx_ms_bucket: counting sort by cell, original order inside a cell).  Returns (sorted points,
    cell_start int32 (ncells+1), origin (host), cell edge, (nx, ny, nz))."""
    import ctypes

    n, nd = fit.shape
    cell = float(bandwidth) * (1.0 + 1e-9)        # strictly larger than the query radius
    ext_d = torch.empty(_clx.ROWS_EXTENT_DOUBLES, dtype=torch.float64, device=fit.device)
    _clx.call("clx_rows_extent_f64", _clx.ptr(fit), n, nd, _clx.ptr(ext_d), _clx.stream_ptr(fit.device))
    ext = ext_d[:2 * nd].cpu().numpy().reshape(2, nd)                  # one small D2H copy
    origin = ext[0].copy()
    dims = np.floor((ext[1] - origin) / cell).astype(np.int64) + 1
    nx, ny = int(dims[0]), int(dims[1])
    nz = int(dims[2]) if nd == 3 else 1
    ncells = nx * ny * nz
    lib = _clx.load()
    ws = torch.empty(int(lib.clx_ms_bucket_workspace(n, ncells)), dtype=torch.uint8, device=fit.device)
    fit_sorted = torch.empty_like(fit)
    cell_start = torch.empty(ncells + 1, dtype=torch.int32, device=fit.device)
    origin_c = (ctypes.c_double * nd)(*origin.tolist())
    _clx.call("clx_ms_bucket", _clx.ptr(fit), n, nd, origin_c, cell, nx, ny, nz, _clx.ptr(fit_sorted),
              _clx.ptr(cell_start), _clx.ptr(ws), _clx.stream_ptr(fit.device))
    return fit_sorted, cell_start, origin, cell, (nx, ny, nz)



def _center_grid(centers, cell):
    """Host side of clx_ms_assign_grid: the (few hundred) centres sorted by uniform-grid cell."""
    nd = centers.shape[1]
    origin = centers.min(axis=0)
    coords = np.floor((centers - origin) / cell).astype(np.int64)
    dims = coords.max(axis=0) + 1
    nx, ny = int(dims[0]), int(dims[1])
    nz = int(dims[2]) if nd == 3 else 1
    cid = coords[:, 0] + nx * coords[:, 1] + (nx * ny * coords[:, 2] if nd == 3 else 0)
    order = np.argsort(cid, kind="stable").astype(np.int32)
    cell_start = np.zeros(nx * ny * nz + 1, dtype=np.int32)
    np.cumsum(np.bincount(cid, minlength=nx * ny * nz), out=cell_start[1:])
    return order, cell_start, origin.astype(np.float64), (nx, ny, nz)


def dedup_centers(centers, counts, bandwidth):
    """sklearn MeanShift.fit post-processing (_mean_shift.py: center_intensity_dict, sort by (count, centre) descending,
    greedy removal of centres within `bandwidth`) — libclx's host function clx_ms_dedup_centers (two sorts + the greedy
    pass over a hash grid); `_dedup_centers_numpy` below is the same in numpy, kept as the tests' second opinion."""
    import ctypes

    centers = np.ascontiguousarray(centers, dtype=np.float64)
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    n, nd = centers.shape
    out = np.empty((max(n, 1), nd), dtype=np.float64)
    kept = ctypes.c_int(0)
    _clx.call("clx_ms_dedup_centers", centers.ctypes.data_as(ctypes.c_void_p), counts.ctypes.data_as(ctypes.c_void_p),
              n, nd, float(bandwidth), out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(kept))
    if kept.value == 0:
        raise ValueError(
            "No point was within bandwidth=%f of any seed. Try a different seeding strategy "
            "or increase the bandwidth." % bandwidth)
    return out[:kept.value].copy()


def _dedup_centers_numpy(centers, counts, bandwidth):
    centers = np.asarray(centers, dtype=np.float64)
    counts = np.asarray(counts)
    keep = counts > 0
    if not keep.any():
        raise ValueError(
            "No point was within bandwidth=%f of any seed. Try a different seeding strategy "
            "or increase the bandwidth." % bandwidth)
    centers, counts = centers[keep], counts[keep]
    # dict semantics: identical centre tuples collapse, the LAST count wins
    _, first_idx, inverse = np.unique(centers, axis=0, return_index=True, return_inverse=True)
    inverse = np.asarray(inverse).reshape(-1)
    last_count = np.zeros(len(first_idx), dtype=counts.dtype)
    last_count[inverse] = counts          # later duplicates overwrite earlier ones
    centers = centers[first_idx]
    counts = last_count
    # sorted(items, key=(count, centre_tuple), reverse=True)
    keys = tuple(centers[:, d] for d in range(centers.shape[1] - 1, -1, -1)) + (counts,)
    order = np.lexsort(keys)[::-1]
    centers = centers[order]
    unique = np.ones(len(centers), dtype=bool)
    bw2 = bandwidth * bandwidth
    for i in range(len(centers)):
        if unique[i]:
            diff = centers - centers[i]
            d2 = np.zeros(len(centers))
            for d in range(centers.shape[1]):
                d2 += diff[:, d] * diff[:, d]
            unique[d2 <= bw2] = False
            unique[i] = True
    return centers[unique]


def _prepare_workspace(nbytes, dev):
    """clx_ms_prepare's scratch: the flag words, the tile counts and their prefix — which clx_ms_assign_dense of the SAME
    call reads again after host-side work (de-duplication, uploads).  Allocated per call (npix / 8 + 12 bytes per tile:
    32 KB for a 512^2 image, from torch's caching allocator) and held by the caller until the assignment has been
    enqueued: a second detection on the same stream — another thread, a pipelined caller — cannot overwrite it."""
    return torch.empty(nbytes, dtype=torch.uint8, device=dev)


def mean_shift_on_device(emb, std, bandwidth, reduction_probability, threshold, seeds=None):
    """emb: (ND, *spatial) f64 device tensor (coordinates are ADDED IN PLACE, as the
    reference does to its argument); std: (*spatial) f64 device tensor.
    Both float32 instead (the network's output handed over in device memory by the fused predict -> detect path):
    the values are read as float64 — what the staged path reads back from the float64 `embeddings` dataset — and
    emb is left UNTOUCHED (that caller has no use for the coordinate-added copy, cellulus/detect.py:155-160).
    Returns (labels int32 device tensor of shape spatial — background 0 —, cluster centres)."""
    _clx.require_device(emb, "embedding")
    nd = emb.shape[0]
    spatial = tuple(emb.shape[1:])
    assert len(spatial) == nd and tuple(std.shape) == spatial
    assert emb.dtype == std.dtype and emb.dtype in (torch.float64, torch.float32)
    assert emb.is_contiguous() and std.is_contiguous()
    prepare = "clx_ms_prepare" if emb.dtype == torch.float64 else "clx_ms_prepare_f32"
    Z, Y, X = (1,) * (3 - nd) + spatial
    npix = Z * Y * X
    dev = emb.device
    st = _clx.stream_ptr(dev)
    lib = _clx.load()
    ws = _prepare_workspace(int(lib.clx_ms_prepare_workspace(npix)), dev)
    pts = torch.empty((npix, nd), dtype=torch.float64, device=dev)
    nfg_d = torch.empty(1, dtype=torch.int32, device=dev)
    labels = torch.empty(spatial, dtype=torch.int32, device=dev)      # written in full by clx_ms_assign_dense
    _clx.zero_many(nfg_d)
    _clx.call(prepare, _clx.ptr(emb), _clx.ptr(std), float(threshold), nd, Z, Y, X,
              _clx.ptr(pts), None, _clx.ptr(nfg_d), _clx.ptr(ws), st)
    nfg = int(nfg_d.item())
    if nfg == 0:      # mean_shift.py:83-84,92-93 -> all -1, +1 -> 0
        _clx.zero_many(labels)
        return labels, np.zeros((0, nd))
    pts = pts[:nfg]
    if reduction_probability < 1.0:
        keep = np.random.rand(nfg) < reduction_probability      # mean_shift.py:69
        # the host drew the mask, so it knows the rows: a gather of known size, not a masked select that has to
        # report its size back
        rows = torch.from_numpy(np.flatnonzero(keep).astype(np.int32)).to(dev, non_blocking=True)
        fit = torch.empty((rows.shape[0], nd), dtype=torch.float64, device=dev)
        _clx.call("clx_gather_rows_f64", _clx.ptr(pts), _clx.ptr(rows), rows.shape[0], nd, _clx.ptr(fit), st)
    else:
        fit = pts
    if fit.shape[0] == 0:
        raise ValueError("Found array with 0 sample(s) (shape=(0, %d)) while a minimum of 1 is "
                         "required by MeanShift." % nd)
    if seeds is None:
        seeds_d = fit
    else:
        seeds_d = torch.as_tensor(np.ascontiguousarray(np.asarray(seeds, dtype=np.float64)), device=dev)
        if seeds_d.ndim != 2 or seeds_d.shape[1] != nd:
            raise ValueError(f"seeds must have shape (n, {nd})")
    ns = seeds_d.shape[0]
    # converged seeds and their neighbour counts in ONE allocation: one copy to the host instead of two
    res = torch.empty(ns * (nd * 8 + 8), dtype=torch.uint8, device=dev)
    centers = res[:ns * nd * 8].view(torch.float64).view(ns, nd)
    counts = res[ns * nd * 8: ns * nd * 8 + ns * 4].view(torch.int32)
    iters = res[ns * nd * 8 + ns * 4:].view(torch.int32)
    if fit.shape[0] >= GRID_MIN_POINTS:
        import ctypes

        fit_sorted, cell_start, origin, cell, (nx, ny, nz) = _bucket(fit, bandwidth)
        if seeds is None:
            seeds_d = fit_sorted          # neighbouring wavefronts then walk neighbouring cells
        origin_c = (ctypes.c_double * nd)(*origin.tolist())
        _clx.call("clx_ms_iterate_grid", _clx.ptr(fit_sorted), fit_sorted.shape[0], _clx.ptr(cell_start),
                  origin_c, cell, nx, ny, nz, _clx.ptr(seeds_d), ns, nd, float(bandwidth), MAX_ITER,
                  _clx.ptr(centers), _clx.ptr(counts), _clx.ptr(iters), st)
    else:
        _clx.call("clx_ms_iterate", _clx.ptr(fit), fit.shape[0], _clx.ptr(seeds_d), ns, nd,
                  float(bandwidth), MAX_ITER, _clx.ptr(centers), _clx.ptr(counts), _clx.ptr(iters), st)
    res_h = res.cpu().numpy()
    cluster_centers = dedup_centers(res_h[:ns * nd * 8].view(np.float64).reshape(ns, nd),
                                    res_h[ns * nd * 8: ns * nd * 8 + ns * 4].view(np.int32), float(bandwidth))
    ncc = cluster_centers.shape[0]
    import ctypes

    # nearest centre of every foreground pixel: the centres in a uniform grid (any edge gives the exact nearest centre —
    # the search widens its block until the winner is closer than the block's reach; the bandwidth is the edge at which
    # a pixel's own 3^ND block decides; doubled while stray centres would make the grid larger than 2^26 cells)
    cell = float(bandwidth)
    order, cstart, corigin, (gx, gy, gz) = _center_grid(cluster_centers, cell)
    while gx * gy * gz > 1 << 26:
        cell *= 2.0
        order, cstart, corigin, (gx, gy, gz) = _center_grid(cluster_centers, cell)
    corigin_c = (ctypes.c_double * nd)(*corigin.tolist())
    # centres in cell order | their ids | the cells' offsets: one buffer, one upload
    nb_c, nb_o = ncc * nd * 8, ncc * 4
    tables = np.empty(nb_c + nb_o + cstart.nbytes, dtype=np.uint8)
    tables[:nb_c].view(np.float64)[:] = cluster_centers[order].reshape(-1)
    tables[nb_c:nb_c + nb_o].view(np.int32)[:] = order
    tables[nb_c + nb_o:].view(np.int32)[:] = cstart
    tables_d = torch.from_numpy(tables).to(dev)
    cc_sorted = tables_d[:nb_c].view(torch.float64).view(ncc, nd)
    order_d = tables_d[nb_c:nb_c + nb_o].view(torch.int32)
    cstart_d = tables_d[nb_c + nb_o:].view(torch.int32)
    # the whole label map in one pass over the compaction's tiles (their flags are still in `ws`): no zero fill, no
    # scatter through a raster index (which clx_ms_prepare was not asked for)
    _clx.call("clx_ms_assign_dense", _clx.ptr(pts), _clx.ptr(cc_sorted), ncc, nd, _clx.ptr(order_d),
              _clx.ptr(cstart_d), corigin_c, cell, gx, gy, gz, _clx.ptr(ws), 0 if emb.dtype == torch.float64 else 1,
              Z, Y, X, _clx.ptr(labels), st)
    return labels, cluster_centers


def _default_device():
    if not torch.cuda.is_available():
        raise _clx.ClxError("mean-shift needs a HIP device; cellulus_amd has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def mean_shift_segmentation(
    embedding_mean,
    embedding_std,
    bandwidth,
    min_size,
    reduction_probability,
    threshold,
    seeds,
    device=None,
):
    """Same contract as mean_shift.py:6-45: embedding_mean (1, ND, *spatial) float64 numpy array
    — pixel coordinates are added to it IN PLACE, like the reference —, embedding_std
    (*spatial); returns the int32 label map (background 0)."""
    device = torch.device(device) if device is not None else _default_device()
    if embedding_mean.dtype != np.float64:
        raise TypeError("embedding_mean must be float64 (the reference reads float64 zarr data)")
    emb = torch.from_numpy(np.ascontiguousarray(embedding_mean[0])).to(device)
    std = torch.from_numpy(np.ascontiguousarray(embedding_std, dtype=np.float64)).to(device)
    labels, _ = mean_shift_on_device(emb, std, bandwidth, reduction_probability, threshold, seeds)
    embedding_mean[0] = emb.cpu().numpy()      # the reference mutates its input (mean_shift.py:15-32)
    return labels.cpu().numpy()


def segment_with_meanshift(embedding, bandwidth, mask, reduction_probability, cluster_all, seeds):
    """mean_shift.py:48-57 — embedding: torch (1, ND, *spatial) f64 with coordinates already
    added; mask: (1, *spatial) bool."""
    anchor_mean_shift = AnchorMeanshift(bandwidth, reduction_probability=reduction_probability,
                                        cluster_all=cluster_all, seeds=seeds)
    return anchor_mean_shift(embedding, mask=mask) + 1


class AnchorMeanshift:
    """mean_shift.py:60-121; `cluster_all` is accepted and — as in the reference, whose labels
    come from MeanShift.predict — has no effect."""

    def __init__(self, bandwidth, reduction_probability, cluster_all, seeds):
        self.bandwidth = bandwidth
        self.reduction_probability = reduction_probability
        self.cluster_all = cluster_all
        self.seeds = seeds

    def compute_masked_ms(self, embedding, mask=None):
        device = embedding.device if embedding.is_cuda else _default_device()
        emb = embedding.to(device=device, dtype=torch.float64).contiguous().clone()
        spatial = tuple(emb.shape[1:])
        if mask is None:
            std = torch.zeros(spatial, dtype=torch.float64, device=device)
        else:
            assert tuple(mask.shape) == spatial
            m = torch.as_tensor(np.asarray(mask.cpu() if torch.is_tensor(mask) else mask)).to(device)
            std = torch.where(m.bool(), 0.0, 2.0).to(torch.float64)
        # coordinates were already added by the caller: undo the kernel's in-place add
        nd = emb.shape[0]
        for c in range(nd):
            ax = nd - 1 - c
            shape = [1] * nd
            shape[ax] = spatial[ax]
            emb[c] -= torch.arange(spatial[ax], dtype=torch.float64, device=device).view(shape)
        labels, _ = mean_shift_on_device(emb, std, self.bandwidth, self.reduction_probability, 1.0, self.seeds)
        return labels.cpu().numpy().astype(np.int32) - 1

    def __call__(self, embedding, mask=None):
        out = []
        for j in range(len(embedding)):
            out.append(self.compute_masked_ms(embedding[j], mask[j] if mask is not None else None))
        return np.stack(out)
