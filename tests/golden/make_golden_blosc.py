"""Generates tests/golden/g12_blosc.npz: chunks compressed by the REAL c-blosc (imagecodecs 2021.8.26
under /opt/conda/bin/python3.9 — what numcodecs.Blosc, zarr's default compressor, wraps) with the
arrays they decode to.  Run:  /opt/conda/bin/python3.9 tests/golden/make_golden_blosc.py
Data only: compressed bytes + expected arrays."""
import os

import imagecodecs
import numpy as np

rng = np.random.default_rng(0)
out = {}


def blobs(shape):
    g = np.meshgrid(*[np.arange(s, dtype=np.float32) for s in shape], indexing="ij")
    img = np.zeros(shape, np.float32)
    for c in rng.uniform(0, 1, size=(12, len(shape))) * np.array(shape):
        img += np.exp(-sum((a - b) ** 2 for a, b in zip(g, c)) / 50.0)
    return img


cases = {
    "f32_lz4_shuffle": (blobs((1, 1, 64, 64)), dict(compressor="lz4", level=5, shuffle=1)),            # zarr default
    "u8_lz4_shuffle": ((blobs((2, 1, 64, 80)) * 255).astype(np.uint8), dict(compressor="lz4", level=5, shuffle=1)),
    "u16_lz4hc_noshuffle": ((blobs((1, 70, 33)) * 60000).astype(np.uint16), dict(compressor="lz4hc", level=9, shuffle=0)),
    "f64_zlib_shuffle": (blobs((40, 41)).astype(np.float64), dict(compressor="zlib", level=3, shuffle=1)),
    "f32_lz4_multiblock": (blobs((2, 48, 80)), dict(compressor="lz4", level=1, shuffle=1, blocksize=4096)),
    "f32_lz4_leftover": (blobs((65, 67)), dict(compressor="lz4", level=5, shuffle=1, blocksize=4096)),
    "random_incompressible": (rng.integers(0, 256, size=5000, dtype=np.uint8), dict(compressor="lz4", level=5, shuffle=1)),
    "tiny": (np.arange(7, dtype=np.int32), dict(compressor="lz4", level=5, shuffle=1)),
    "i64_labels_lz4": (np.repeat(rng.integers(0, 5, size=300), 37).astype(np.int64).reshape(100, 111),
                       dict(compressor="lz4", level=5, shuffle=1)),
    # (appended in round 2, after the cases above so that their random draws are unchanged)
    "f32_zstd_shuffle": (blobs((1, 1, 64, 64)), dict(compressor="zstd", level=3, shuffle=1)),
    "u16_zstd_noshuffle_multiblock": ((blobs((3, 50, 64)) * 60000).astype(np.uint16), dict(compressor="zstd", level=1, shuffle=0, blocksize=8192)),
    "f32_lz4_bitshuffle": (blobs((1, 1, 64, 64)), dict(compressor="lz4", level=5, shuffle=2)),
    "u8_zstd_bitshuffle_leftover": ((blobs((65, 67)) * 255).astype(np.uint8), dict(compressor="zstd", level=5, shuffle=2, blocksize=4096)),
    "f64_lz4_bitshuffle_odd": (blobs((37, 41)).astype(np.float64), dict(compressor="lz4", level=5, shuffle=2)),
    "f32_blosclz_shuffle": (blobs((1, 1, 64, 64)), dict(compressor="blosclz", level=5, shuffle=1)),
    "i64_labels_blosclz": (np.repeat(rng.integers(0, 5, size=300), 37).astype(np.int64).reshape(100, 111),
                           dict(compressor="blosclz", level=9, shuffle=1)),
    "u8_blosclz_noshuffle": ((blobs((2, 1, 64, 80)) * 255).astype(np.uint8), dict(compressor="blosclz", level=5, shuffle=0)),
}
for name, (arr, kw) in cases.items():
    kw = dict(kw)
    blocksize = kw.pop("blocksize", None)
    extra = dict(blocksize=blocksize) if blocksize else {}
    enc = imagecodecs.blosc_encode(np.ascontiguousarray(arr).tobytes(), typesize=arr.dtype.itemsize, **kw, **extra)
    assert imagecodecs.blosc_decode(enc) == arr.tobytes()
    out[f"{name}/chunk"] = np.frombuffer(enc, dtype=np.uint8)
    out[f"{name}/array"] = arr
    print(name, arr.dtype, arr.shape, len(enc), "of", arr.nbytes, "flags", hex(enc[2]))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "g12_blosc.npz"), **out)
