"""Streaming (HBM-bound) kernels of the inference path at batch scale: algorithmic bytes /
HIP-event time vs the 8 TB/s HBM3E peak (MI355X_MICROARCH.md).  Usage: python tools/bench_stream.py"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from cellulus_amd import _clx  # noqa: E402
from cellulus_amd.segment import grow_shrink_on_device  # noqa: E402
from cellulus_amd.utils.misc import label_on_device  # noqa: E402

dev = torch.device("cuda:0")
HBM = 8.0e12


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    out = {}
    Y = X = 8192
    npix = Y * X
    rng = np.random.default_rng(0)
    st = _clx.stream_ptr(dev)
    lib = _clx.load()
    # --- ms_prepare: reads (ND+1)*8 B, writes ND*8 B per pixel (+ (ND*8+4) per fg pixel)
    emb = torch.randn(2, Y, X, dtype=torch.float64, device=dev)
    sd = torch.rand(Y, X, dtype=torch.float64, device=dev)
    ws = torch.empty(int(lib.clx_ms_prepare_workspace(npix)), dtype=torch.uint8, device=dev)
    pts = torch.empty((npix, 2), dtype=torch.float64, device=dev)
    idx = torch.empty(npix, dtype=torch.int32, device=dev)
    nfg = torch.zeros(1, dtype=torch.int32, device=dev)
    t = timeit(lambda: _clx.call("clx_ms_prepare", _clx.ptr(emb), _clx.ptr(sd), 0.2, 2, 1, Y, X, _clx.ptr(pts),
                                 _clx.ptr(idx), _clx.ptr(nfg), _clx.ptr(ws), st))
    n_fg = int(nfg.item())
    b = npix * (3 * 8 + 2 * 8) + npix * 8 + n_fg * (2 * 8 + 2 * 8 + 4)   # scatter pass re-reads std + fg emb
    out["ms_prepare"] = dict(ms=t * 1e3, GBs=b / t / 1e9, frac=b / t / HBM, npix=npix, nfg=n_fg)
    # --- ms_assign: reads ND*8+4 per fg pixel, writes 4 B
    centers = torch.rand(121, 2, dtype=torch.float64, device=dev) * X
    labels = torch.zeros(npix, dtype=torch.int32, device=dev)
    t = timeit(lambda: _clx.call("clx_ms_assign", _clx.ptr(pts), _clx.ptr(idx), n_fg, _clx.ptr(centers), 121, 2,
                                 _clx.ptr(labels), st))
    b = n_fg * (16 + 4 + 4)
    out["ms_assign"] = dict(ms=t * 1e3, GBs=b / t / 1e9, frac=b / t / HBM, pair_evals_per_s=n_fg * 121 / t)
    del emb, sd, pts, idx
    # --- CC + size filter: algorithmic 4 B read + 4 B write per pixel (x ~6 internal passes)
    blocks = rng.integers(0, 60, size=(Y // 32, X // 32)).astype(np.int32)
    seg = torch.from_numpy(np.kron(blocks, np.ones((32, 32), dtype=np.int32))).to(dev)
    t = timeit(lambda: label_on_device(seg, 70), reps=3)
    out["cc_label_filter"] = dict(ms=t * 1e3, GBs=npix * 8 / t / 1e9, frac=npix * 8 / t / HBM)
    # --- grow/shrink: 4 B read + 4 B write per pixel (two EDTs inside)
    t = timeit(lambda: grow_shrink_on_device(seg.clone(), 3, 6), reps=3)
    out["grow_shrink"] = dict(ms=t * 1e3, GBs=npix * 8 / t / 1e9, frac=npix * 8 / t / HBM)
    # --- histogram + minmax: 8 B per pixel each
    x = torch.rand(npix, dtype=torch.float64, device=dev)
    mm = torch.empty(2, dtype=torch.float64, device=dev)
    t = timeit(lambda: _clx.call("clx_minmax_f64", _clx.ptr(x), npix, _clx.ptr(mm), st))
    out["minmax_f64"] = dict(ms=t * 1e3, GBs=npix * 8 / t / 1e9, frac=npix * 8 / t / HBM)
    edges = torch.linspace(0, 1, 257, dtype=torch.float64, device=dev)
    counts = torch.zeros(256, dtype=torch.int64, device=dev)
    t = timeit(lambda: _clx.call("clx_histogram_f64", _clx.ptr(x), npix, _clx.ptr(edges), 256, _clx.ptr(counts), st))
    out["histogram_f64"] = dict(ms=t * 1e3, GBs=npix * 8 / t / 1e9, frac=npix * 8 / t / HBM)
    # --- noise stats: T*C*4 read + (C+1)*4 write per pixel
    T, C, n = 32, 2, 2048 * 2048
    preds = torch.randn(T, C, n, device=dev)
    o = torch.empty(C + 1, n, device=dev)
    t = timeit(lambda: _clx.call("clx_noise_stats", _clx.ptr(preds), _clx.ptr(o), T, C, n, st))
    b = n * (T * C * 4 + (C + 1) * 4)
    out["noise_stats"] = dict(ms=t * 1e3, GBs=b / t / 1e9, frac=b / t / HBM)
    for k, v in out.items():
        print(k, json.dumps({kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}))


if __name__ == "__main__":
    main()
