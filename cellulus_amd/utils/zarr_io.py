"""Minimal zarr-v2 directory-store reader/writer (zarr itself is not installed).

Covers what the hot path's callers need (``cellulus/datasets/meta_data.py``,
``predict.py:103-142``, ``detect.py:18-80``, ``segment.py:19-38``,
``train.py:194-224``): C-order arrays of bool/int/uint/float dtypes, chunked,
``compressor`` null / zlib / gzip / blosc (read only: the LZ4 and zlib codecs, i.e. what
zarr-python writes by default; decoded by libclx's host-side LZ4 block decoder),
``fill_value``, ``.zattrs``.  Groups are plain directories with a ``.zgroup``.
"""

import gzip
import io
import itertools
import json
import os
import zlib

import numpy as np


class ZarrError(RuntimeError):
    pass


class ContainsArrayError(ZarrError, ValueError):
    """zarr.errors.ContainsArrayError: create_dataset on an existing path without overwrite."""


def _json_dump(path, obj):
    tmp = path + ".tmp"
    with io.open(tmp, "w") as f:
        json.dump(obj, f, indent=4)
    os.replace(tmp, path)


class Attributes:
    """dict-like view of a node's .zattrs, written through on every assignment."""

    def __init__(self, node_path):
        self._path = os.path.join(node_path, ".zattrs")

    def _load(self):
        if os.path.exists(self._path):
            with io.open(self._path) as f:
                return json.load(f)
        return {}

    def __getitem__(self, key):
        return self._load()[key]

    def __contains__(self, key):
        return key in self._load()

    def get(self, key, default=None):
        return self._load().get(key, default)

    def __setitem__(self, key, value):
        d = self._load()
        if isinstance(value, np.ndarray):
            value = value.tolist()
        if isinstance(value, tuple):
            value = list(value)
        d[key] = value
        _json_dump(self._path, d)

    def asdict(self):
        return self._load()

    def keys(self):
        return self._load().keys()


_BLOSC_CODECS = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}


def blosc_decode(raw, expected=None):
    """One Blosc (format 2, c-blosc 1.x — what numcodecs.Blosc writes) chunk -> bytes.

    Layout: 16-byte header (version, versionlz, flags, typesize, nbytes, blocksize, cbytes), then —
    unless the chunk is stored verbatim (flag 0x02) — a table of block start offsets and the
    blocks.  A block is `typesize` separately compressed byte planes ("splits") when it was
    byte-shuffled into planes of >= 128 bytes and the don't-split flag (0x10) is clear, else one
    stream; every stream is an int32 length followed by the codec's output (or the bytes
    themselves when the length equals the plain size); the last, shorter block is never split.
    Codecs: LZ4 / LZ4HC (zarr's default) and BloscLZ (decoded by libclx: clx_lz4_decompress,
    clx_blosclz_decompress), zlib, and — when pyarrow is importable (its bundled codecs) — zstd and
    snappy; byte shuffle is undone by clx_unshuffle_bytes, bit shuffle by `_bit_unshuffle`.
    ``expected`` (the chunk's byte count, prod(chunks) * itemsize): a header announcing anything
    else is rejected BEFORE a buffer of the announced size is allocated."""
    import ctypes

    from .. import _clx

    raw = bytes(raw)
    if len(raw) < 16:
        raise ZarrError("Blosc chunk shorter than its header")
    flags, typesize = raw[2], raw[3]
    nbytes, blocksize, cbytes = (int.from_bytes(raw[o:o + 4], "little") for o in (4, 8, 12))
    if cbytes > len(raw) or typesize < 1:
        raise ZarrError("corrupt Blosc header")
    if expected is not None and nbytes != expected:
        raise ZarrError(f"Blosc chunk announces {nbytes} bytes, the array's chunks hold {expected}")
    if nbytes == 0:
        return b""
    if flags & 0x02:                                     # memcpyed
        return raw[16:16 + nbytes]
    codec = _BLOSC_CODECS.get(flags >> 5, "?")
    if codec not in ("lz4", "zlib", "zstd", "snappy", "blosclz"):
        raise ZarrError(f"Blosc codec id {flags >> 5} is not known to this reader (blosclz, lz4, lz4hc, snappy, zlib and zstd are)")
    arrow = _arrow_codec(codec) if codec in ("zstd", "snappy") else None
    if blocksize <= 0 or blocksize > nbytes:
        raise ZarrError("corrupt Blosc header (block size)")
    nblocks = -(-nbytes // blocksize)
    if 16 + 4 * nblocks > len(raw):
        raise ZarrError("corrupt Blosc header (block table longer than the chunk)")
    leftover = nbytes % blocksize
    bitshuffle = bool(flags & 0x04)
    shuffle = (bool(flags & 0x01) and typesize > 1) or bitshuffle
    dont_split = bool(flags & 0x10)
    lib = _clx.load()
    src = np.frombuffer(raw, dtype=np.uint8)
    out = np.empty(nbytes, dtype=np.uint8)
    tmp = np.empty(blocksize, dtype=np.uint8)
    u8p = ctypes.c_void_p
    for b in range(nblocks):
        bsize = leftover if (b == nblocks - 1 and leftover) else blocksize
        is_leftover = b == nblocks - 1 and leftover > 0
        pos = int.from_bytes(raw[16 + 4 * b:20 + 4 * b], "little")
        if pos < 16 + 4 * nblocks or pos + 4 > len(raw):
            raise ZarrError("corrupt Blosc block offset")
        nsplits = typesize if (not dont_split and typesize <= 16 and blocksize // typesize >= 128
                               and not is_leftover) else 1
        neblock = bsize // nsplits
        target = tmp if shuffle else out[b * blocksize:]
        done = 0
        for _ in range(nsplits):
            clen = int.from_bytes(raw[pos:pos + 4], "little", signed=True)
            pos += 4
            if clen < 0 or pos + clen > len(raw):
                raise ZarrError("corrupt Blosc block")
            if clen == neblock:
                target[done:done + neblock] = src[pos:pos + neblock]
            elif codec in ("lz4", "blosclz"):
                fn = lib.clx_lz4_decompress if codec == "lz4" else lib.clx_blosclz_decompress
                n = fn(u8p(src.ctypes.data + pos), clen, u8p(target.ctypes.data + done), neblock)
                if n != neblock:
                    raise ZarrError(f"{codec} stream of a Blosc block decoded to {n} bytes, expected {neblock}")
            else:
                if arrow is not None:
                    piece = arrow.decompress(raw[pos:pos + clen], decompressed_size=neblock, asbytes=True)
                else:
                    piece = zlib.decompress(raw[pos:pos + clen])
                if len(piece) != neblock:
                    raise ZarrError(f"{codec} stream of a Blosc block has the wrong length")
                target[done:done + neblock] = np.frombuffer(piece, dtype=np.uint8)
            pos += clen
            done += neblock
        if bitshuffle:
            out[b * blocksize:b * blocksize + bsize] = _bit_unshuffle(tmp[:bsize], typesize)
        elif shuffle:
            dst = out[b * blocksize:]
            lib.clx_unshuffle_bytes(u8p(tmp.ctypes.data), u8p(dst.ctypes.data), bsize, typesize)
    return out.tobytes()


def _bit_unshuffle(block, typesize):
    """Inverse of c-blosc's bit shuffle of one block (c-blosc 1.x `bitshuffle()`): a block of n = 8 m elements is
    stored as [byte of the element][bit][element / 8] (bit k of a stored byte belongs to element 8 j + k) with
    any bytes past n * typesize copied; a block whose element count is not a multiple of 8 is stored unshuffled."""
    n = len(block) // typesize
    if n == 0 or n % 8:
        return block
    out = np.empty(len(block), dtype=np.uint8)
    bits = np.unpackbits(block[:n * typesize].reshape(typesize, 8, n // 8), axis=-1, bitorder="little")
    out[:n * typesize] = np.packbits(bits.transpose(2, 0, 1), axis=-1, bitorder="little").reshape(-1)
    out[n * typesize:] = block[n * typesize:]
    return out


def _arrow_codec(name):
    """zstd / snappy streams are decoded by the codecs bundled with pyarrow when it is importable."""
    try:
        import pyarrow
    except ImportError as e:
        raise ZarrError(f"reading {name}-compressed chunks needs pyarrow (its bundled {name} codec); not importable: {e}")
    if not pyarrow.Codec.is_available(name):
        raise ZarrError(f"this pyarrow build has no {name} codec")
    return pyarrow.Codec(name)


def _decode(raw, compressor, expected=None):
    """Chunk bytes -> array bytes.  ``expected`` = prod(chunks) * itemsize when the caller knows it:
    size fields of untrusted chunk headers are checked against it before anything is allocated."""
    if compressor is None:
        return raw
    cid = compressor.get("id")
    if cid == "zlib":
        return zlib.decompress(raw)
    if cid == "gzip":
        return gzip.decompress(raw)
    if cid == "blosc":
        return blosc_decode(raw, expected)
    if cid == "zstd":                                    # numcodecs.Zstd: one zstd frame with its content size
        return _zstd_frame(bytes(raw))
    if cid == "lz4":                                     # numcodecs.LZ4: int32 plain size + one LZ4 block
        import ctypes

        from .. import _clx
        raw = bytes(raw)
        if len(raw) < 4:
            raise ZarrError("LZ4 chunk shorter than its size field")
        n = int.from_bytes(raw[:4], "little")
        if expected is not None and n != expected:
            raise ZarrError(f"LZ4 chunk announces {n} bytes, the array's chunks hold {expected}")
        out = np.empty(n, dtype=np.uint8)
        src = np.frombuffer(raw, dtype=np.uint8)
        got = _clx.load().clx_lz4_decompress(ctypes.c_void_p(src.ctypes.data + 4), len(raw) - 4,
                                             ctypes.c_void_p(out.ctypes.data), n)
        if got != n:
            raise ZarrError(f"LZ4 chunk decoded to {got} bytes, expected {n}")
        return out.tobytes()
    if cid == "bz2":
        import bz2
        return bz2.decompress(raw)
    raise ZarrError(f"unsupported zarr compressor {cid!r} (null, zlib, gzip, bz2, lz4, zstd and "
                    "blosc[lz4 | lz4hc | zlib | zstd | snappy] can be read here)")


def _zstd_frame(raw):
    """A zstd frame whose header carries the content size (numcodecs.Zstd always writes it)."""
    if raw[:4] != b"\x28\xb5\x2f\xfd":
        raise ZarrError("not a zstd frame")
    fhd = raw[4]
    single, dict_flag, fcs_flag = (fhd >> 5) & 1, fhd & 3, fhd >> 6
    pos = 5 + (0 if single else 1) + (0, 1, 2, 4)[dict_flag]
    width = (1 if single else 0, 2, 4, 8)[fcs_flag]
    if width == 0:
        raise ZarrError("zstd frame without a content size")
    size = int.from_bytes(raw[pos:pos + width], "little") + (256 if width == 2 else 0)
    return _arrow_codec("zstd").decompress(raw, decompressed_size=size, asbytes=True)


# what zarr-python 2.x uses when no compressor is given (numcodecs.Blosc's defaults): the reference's stages
# create their output datasets this way (cellulus/predict.py:103-110, detect.py:23-70, segment.py:24-33)
DEFAULT_COMPRESSOR = {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}


def _encode(raw, compressor, typesize=1):
    if compressor is None:
        return raw
    cid = compressor.get("id")
    if cid == "zlib":
        return zlib.compress(raw, compressor.get("level", 1))
    if cid == "gzip":
        return gzip.compress(raw, compressor.get("level", 1))
    if cid == "blosc":
        if compressor.get("cname", "lz4") != "lz4" or compressor.get("shuffle", 1) not in (0, 1, -1):
            raise ZarrError("this writer produces Blosc chunks with LZ4 inside and byte shuffle or none "
                            f"(asked for {compressor!r})")
        import ctypes

        from .. import _clx
        lib = _clx.load()
        src = np.frombuffer(raw, dtype=np.uint8)
        cap = lib.clx_blosc_compress_bound(src.size)
        out = np.empty(cap, dtype=np.uint8)
        # numcodecs' AUTOSHUFFLE (-1): byte shuffle unless the items are single bytes
        shuffle = 0 if compressor.get("shuffle", 1) == 0 or typesize == 1 else 1
        src_p = ctypes.c_void_p(src.ctypes.data) if src.size else ctypes.c_void_p(out.ctypes.data)   # (never read)
        n = lib.clx_blosc_compress_lz4(src_p, src.size, int(typesize), shuffle, ctypes.c_void_p(out.ctypes.data), cap)
        if n < 0:
            raise ZarrError(f"Blosc encoder failed ({n})")
        return out[:n].tobytes()
    raise ZarrError(f"unsupported zarr compressor {cid!r}")


class Array:
    def __init__(self, path):
        self.path = path
        meta_path = os.path.join(path, ".zarray")
        if not os.path.exists(meta_path):
            raise ZarrError(f"{path} is not a zarr array")
        with io.open(meta_path) as f:
            m = json.load(f)
        if m.get("zarr_format") != 2:
            raise ZarrError("only zarr format 2 is supported")
        if m.get("order", "C") != "C":
            raise ZarrError("only C-order zarr arrays are supported")
        if m.get("filters"):
            raise ZarrError("zarr filters are not supported")
        self.shape = tuple(m["shape"])
        self.chunks = tuple(m["chunks"])
        self.dtype = np.dtype(m["dtype"])
        self.compressor = m.get("compressor")
        self.fill_value = m.get("fill_value")
        self.sep = m.get("dimension_separator", ".")
        self.attrs = Attributes(path)
        # decoded chunks of a COMPRESSED array, most recently used last (random crops of a training image hit
        # the same few chunks again and again; CLX_ZARR_CACHE_MB per array and process, 0 = off)
        self._cache = {}
        self._cache_bytes = 0
        self._cache_cap = int(float(os.environ.get("CLX_ZARR_CACHE_MB", "64")) * (1 << 20))

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        return self.shape[0]

    def _fill(self):
        fv = self.fill_value
        if fv is None:
            return 0
        if isinstance(fv, str):
            return {"NaN": np.nan, "Infinity": np.inf, "-Infinity": -np.inf}.get(fv, 0)
        return fv

    def _chunk_path(self, idx):
        return os.path.join(self.path, self.sep.join(str(i) for i in idx))

    def _read_chunk(self, idx):
        p = self._chunk_path(idx)
        if not os.path.exists(p):
            return np.full(self.chunks, self._fill(), dtype=self.dtype)
        use_cache = self.compressor is not None and self._cache_cap > 0
        if use_cache:
            key = tuple(idx)
            try:
                stamp = os.stat(p).st_mtime_ns
            except OSError:
                stamp = None
            hit = self._cache.get(key)
            if hit is not None and hit[0] == stamp:
                self._cache[key] = self._cache.pop(key)                  # most recently used
                return hit[1]
        with io.open(p, "rb") as f:
            raw = _decode(f.read(), self.compressor, int(np.prod(self.chunks)) * self.dtype.itemsize)
        chunk = np.frombuffer(raw, dtype=self.dtype).reshape(self.chunks)
        if use_cache and chunk.nbytes <= self._cache_cap:
            old = self._cache.pop(key, None)
            if old is not None:
                self._cache_bytes -= old[1].nbytes
            self._cache[key] = (stamp, chunk)
            self._cache_bytes += chunk.nbytes
            while self._cache_bytes > self._cache_cap:
                _k, (_s, gone) = next(iter(self._cache.items()))
                del self._cache[_k]
                self._cache_bytes -= gone.nbytes
        return chunk

    def _write_chunk(self, idx, data):
        old = self._cache.pop(tuple(idx), None)
        if old is not None:
            self._cache_bytes -= old[1].nbytes
        p = self._chunk_path(idx)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        tmp = p + ".tmp"
        with io.open(tmp, "wb") as f:
            f.write(_encode(np.ascontiguousarray(data, dtype=self.dtype).tobytes(), self.compressor, self.dtype.itemsize))
        os.replace(tmp, p)

    def _normalize(self, key):
        """index expression -> (list of slices, axes to squeeze)."""
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = key.index(Ellipsis)
            fill = self.ndim - (len(key) - 1)
            key = key[:i] + (slice(None),) * fill + key[i + 1:]
        key = key + (slice(None),) * (self.ndim - len(key))
        if len(key) != self.ndim:
            raise IndexError("too many indices for zarr array")
        slices, squeeze = [], []
        for ax, (k, n) in enumerate(zip(key, self.shape)):
            if isinstance(k, (int, np.integer)):
                k = int(k)
                if k < 0:
                    k += n
                if not 0 <= k < n:
                    raise IndexError("index out of bounds")
                slices.append(slice(k, k + 1))
                squeeze.append(ax)
            elif isinstance(k, slice):
                start, stop, step = k.indices(n)
                if step != 1:
                    raise IndexError("only unit-step slices are supported")
                slices.append(slice(start, max(stop, start)))
            else:
                raise IndexError("only integers, slices and Ellipsis are supported")
        return slices, tuple(squeeze)

    def __getitem__(self, key):
        slices, squeeze = self._normalize(key)
        out_shape = tuple(s.stop - s.start for s in slices)
        out = np.empty(out_shape, dtype=self.dtype)
        ranges = [range(s.start // c, (max(s.stop, s.start + 1) - 1) // c + 1) if s.stop > s.start else range(0)
                  for s, c in zip(slices, self.chunks)]
        for idx in itertools.product(*ranges):
            chunk = self._read_chunk(idx)
            src, dst = [], []
            for i, s, c in zip(idx, slices, self.chunks):
                lo, hi = max(s.start, i * c), min(s.stop, (i + 1) * c)
                src.append(slice(lo - i * c, hi - i * c))
                dst.append(slice(lo - s.start, hi - s.start))
            out[tuple(dst)] = chunk[tuple(src)]
        return out.squeeze(axis=squeeze) if squeeze else out

    def __setitem__(self, key, value):
        slices, squeeze = self._normalize(key)
        region = tuple(s.stop - s.start for s in slices)
        value = np.asarray(value)
        if squeeze:
            value = np.expand_dims(value, squeeze) if value.ndim == len(region) - len(squeeze) else value
        value = np.broadcast_to(value.astype(self.dtype, copy=False), region)
        ranges = [range(s.start // c, (max(s.stop, s.start + 1) - 1) // c + 1) if s.stop > s.start else range(0)
                  for s, c in zip(slices, self.chunks)]
        for idx in itertools.product(*ranges):
            src, dst, full = [], [], True
            for i, s, c, n in zip(idx, slices, self.chunks, self.shape):
                lo, hi = max(s.start, i * c), min(s.stop, (i + 1) * c)
                dst.append(slice(lo - i * c, hi - i * c))
                src.append(slice(lo - s.start, hi - s.start))
                if hi - lo != c:
                    full = False
            if full:
                chunk = value[tuple(src)]
            else:
                chunk = self._read_chunk(idx).copy()
                chunk[tuple(dst)] = value[tuple(src)]
            self._write_chunk(idx, chunk)


class Group:
    def __init__(self, path, create=False):
        self.path = path
        if create:
            os.makedirs(path, exist_ok=True)
            zg = os.path.join(path, ".zgroup")
            if not os.path.exists(zg) and not os.path.exists(os.path.join(path, ".zarray")):
                _json_dump(zg, {"zarr_format": 2})
        elif not os.path.isdir(path):
            raise ZarrError(f"zarr container {path} does not exist")
        self.attrs = Attributes(path)

    def __contains__(self, name):
        p = os.path.join(self.path, name)
        return os.path.exists(os.path.join(p, ".zarray")) or os.path.exists(os.path.join(p, ".zgroup"))

    def __getitem__(self, name):
        p = os.path.join(self.path, name)
        if os.path.exists(os.path.join(p, ".zarray")):
            return Array(p)
        if os.path.isdir(p):
            return Group(p)
        raise KeyError(name)

    def __delitem__(self, name):
        import shutil

        if name not in self:
            raise KeyError(name)
        shutil.rmtree(os.path.join(self.path, name))

    def _make_parents(self, name):
        parts = name.strip("/").split("/")
        cur = self.path
        for part in parts[:-1]:
            cur = os.path.join(cur, part)
            Group(cur, create=True)
        return os.path.join(cur, parts[-1])

    def create_dataset(self, name, shape, dtype, chunks=None, compressor="default", fill_value=0,
                       overwrite=False):
        """zarr's ``Group.create_dataset``: raises when `name` exists unless ``overwrite=True``
        (zarr-python's ContainsArrayError; the reference's stages therefore refuse to clobber
        an existing ``embeddings`` / ``detection`` / ``segmentation`` dataset).  As in zarr-python the
        chunks are Blosc / LZ4 / byte-shuffle compressed unless ``compressor=None`` is passed."""
        if isinstance(compressor, str):
            if compressor != "default":
                raise ZarrError(f"compressor must be 'default', None or a numcodecs-style dict, got {compressor!r}")
            # (CLX_ZARR_COMPRESSOR=none: plain chunks, for measurements of the write path)
            compressor = None if os.environ.get("CLX_ZARR_COMPRESSOR", "") == "none" else dict(DEFAULT_COMPRESSOR)
        p = self._make_parents(name)
        dtype = np.dtype(dtype)
        shape = tuple(int(s) for s in shape)
        if chunks is None:
            # one chunk per leading index keeps per-sample writes cheap
            chunks = (1,) + shape[1:] if len(shape) > 1 else shape
        chunks = tuple(max(1, int(min(c, s))) if s > 0 else 1 for c, s in zip(chunks, shape))
        if os.path.exists(os.path.join(p, ".zarray")) or os.path.exists(os.path.join(p, ".zgroup")):
            if not overwrite:
                raise ContainsArrayError(f"path {name!r} contains an array")
            import shutil

            shutil.rmtree(p)
        os.makedirs(p, exist_ok=True)
        meta = {
            "zarr_format": 2,
            "shape": list(shape),
            "chunks": list(chunks),
            "dtype": dtype.str,
            "compressor": compressor,
            "fill_value": fill_value,
            "order": "C",
            "filters": None,
        }
        _json_dump(os.path.join(p, ".zarray"), meta)
        return Array(p)

    def __setitem__(self, name, value):
        value = np.asarray(value)
        # zarr: group[name] = value  ==  group.array(name, value, overwrite=True)
        arr = self.create_dataset(name, shape=value.shape, dtype=value.dtype, overwrite=True)
        arr[...] = value


def open(path, mode="a"):  # noqa: A001 - mirrors zarr.open
    """zarr.open(path[, mode]) for directory stores; returns a Group (or an Array if path is one)."""
    path = str(path)
    if os.path.exists(os.path.join(path, ".zarray")):
        return Array(path)
    if mode == "r":
        return Group(path, create=False)
    return Group(path, create=True)
