"""End-to-end infer() with the output datasets written as zarr-python writes them (Blosc / LZ4 / byte shuffle,
the default) against plain chunks (CLX_ZARR_COMPRESSOR=none), and the encoder's speed on one embedding chunk.
Usage: python tests/diag/e2e_compress_ab.py"""
import os, sys, time, ctypes
sys.path.insert(0, ".")
import numpy as np, torch
import bench_infer
from cellulus_amd.utils import zarr_io
dev=torch.device("cuda:0")
# raw encoder speed on a typical embedding chunk
a=np.random.default_rng(0).normal(size=(3,512,512)).cumsum(axis=-1)
t0=time.perf_counter(); c=zarr_io._encode(a.tobytes(), zarr_io.DEFAULT_COMPRESSOR, 8); dt=time.perf_counter()-t0
print(f"encode 6.3 MB f64 chunk: {dt*1e3:.1f} ms = {a.nbytes/dt/1e6:.0f} MB/s, ratio {len(c)/a.nbytes:.2f}")
for mode in ("default","none","default","none"):
    os.environ["CLX_ZARR_COMPRESSOR"]="" if mode=="default" else "none"
    r=bench_infer.e2e_infer(dev, samples=32)
    print(mode, {k:r[k] for k in ("mpixels_s","seconds","ms_per_sample")})
