"""NumPy restatement of the Blosc 1 frame format and of the Zarr v3 chunk layout the datastore uses.

ORACLE / TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.

The reference stores every image through tensorstore's ``zarr3`` driver with the codec chain
``bytes`` (little endian) -> ``blosc`` (cname zstd, clevel 5, shuffle bitshuffle)
(qi2labDataStore.py:1464-1527); the container format is c-blosc 1.x's (third party, bundled by
tensorstore, absent from the reference tree).  Restated here from the published format
(c-blosc ``README_HEADER.rst``, ``blosc.c``, ``shuffle.c``) and pinned by
``tests/golden/blosc_frames.npz``: frames written by the real c-blosc 1.21 (through
imagecodecs, ``tests/golden/make_golden_blosc.py``).  The entropy stage itself (zstd / lz4) is
the system library, called through ctypes.
"""

from __future__ import annotations

import ctypes
import itertools
from pathlib import Path

import numpy as np

_zstd = None
_lz4 = None


def _libzstd():
    global _zstd
    if _zstd is None:
        _zstd = ctypes.CDLL("libzstd.so.1")
        _zstd.ZSTD_decompress.restype = ctypes.c_size_t
        _zstd.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        _zstd.ZSTD_compress.restype = ctypes.c_size_t
        _zstd.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _zstd.ZSTD_compressBound.restype = ctypes.c_size_t
        _zstd.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        _zstd.ZSTD_isError.argtypes = [ctypes.c_size_t]
    return _zstd


def _liblz4():
    global _lz4
    if _lz4 is None:
        _lz4 = ctypes.CDLL("liblz4.so.1")
        _lz4.LZ4_decompress_safe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    return _lz4


def parse_header(frame: np.ndarray) -> dict:
    f = np.asarray(frame, dtype=np.uint8)
    flags = int(f[2])
    u32 = lambda o: int.from_bytes(f[o:o + 4].tobytes(), "little")
    h = {"version": int(f[0]), "typesize": int(f[3]), "nbytes": u32(4), "blocksize": u32(8), "cbytes": u32(12),
         "shuffle": 2 if flags & 0x4 else (1 if flags & 0x1 else 0), "memcpyed": bool(flags & 0x2),
         "split": not (flags & 0x10), "codec": flags >> 5}
    h["nblocks"] = -(-h["nbytes"] // h["blocksize"]) if h["nbytes"] else 0
    return h


def _entropy_decode(codec: int, src: np.ndarray, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    if codec == 4:
        z = _libzstd()
        got = z.ZSTD_decompress(out.ctypes.data, n, src.ctypes.data, src.size)
        assert not z.ZSTD_isError(got) and got == n
    elif codec == 1:
        got = _liblz4().LZ4_decompress_safe(src.ctypes.data, out.ctypes.data, src.size, n)
        assert got == n
    else:
        raise NotImplementedError(f"blosc codec {codec}")
    return out


def unshuffle_block(block: np.ndarray, typesize: int, shuffle: int) -> np.ndarray:
    """Inverse of c-blosc 1's shuffle stage on one block (``shuffle.c``)."""
    T, n = typesize, block.size
    ne = n // T
    if shuffle == 1 and T > 1:
        body = block[: ne * T].reshape(T, ne).T.reshape(-1)
        return np.concatenate([body, block[ne * T:]])
    if shuffle == 2 and n >= T and ne % 8 == 0:
        rows = block[: ne * T].reshape(8 * T, ne // 8)                 # row r = bit r of every element
        bits = np.unpackbits(rows, axis=1, bitorder="little")          # (8T, ne)
        by = np.packbits(bits.reshape(T, 8, ne), axis=1, bitorder="little").reshape(T, ne)  # byte j of each element
        return np.concatenate([by.T.reshape(-1), block[ne * T:]])
    return block.copy()


def shuffle_block(block: np.ndarray, typesize: int, shuffle: int) -> np.ndarray:
    T, n = typesize, block.size
    ne = n // T
    if shuffle == 1 and T > 1:
        return np.concatenate([block[: ne * T].reshape(ne, T).T.reshape(-1), block[ne * T:]])
    if shuffle == 2 and n >= T and ne % 8 == 0:
        by = block[: ne * T].reshape(ne, T).T                              # (T, ne)
        bits = np.unpackbits(by[:, None, :], axis=1, bitorder="little")    # (T, 8, ne)
        rows = np.packbits(bits.reshape(8 * T, ne), axis=1, bitorder="little")
        return np.concatenate([rows.reshape(-1), block[ne * T:]])
    return block.copy()


def decode(frame: np.ndarray) -> np.ndarray:
    """Raw bytes of a Blosc 1 frame (``blosc_d`` of blosc.c)."""
    f = np.ascontiguousarray(frame, dtype=np.uint8)
    h = parse_header(f)
    assert h["version"] == 2
    if h["memcpyed"]:
        return f[16:16 + h["nbytes"]].copy()
    T, out = h["typesize"], []
    for b in range(h["nblocks"]):
        bsize = min(h["blocksize"], h["nbytes"] - b * h["blocksize"])
        leftover = bsize != h["blocksize"]
        nsplits = T if (h["split"] and T <= 16 and h["blocksize"] // T >= 128 and not leftover) else 1
        ne = bsize // nsplits
        pos = int.from_bytes(f[16 + 4 * b:20 + 4 * b].tobytes(), "little")
        parts = []
        for _ in range(nsplits):
            cs = int.from_bytes(f[pos:pos + 4].tobytes(), "little")
            pos += 4
            src = f[pos:pos + cs]
            parts.append(src.copy() if cs == ne else _entropy_decode(h["codec"], np.ascontiguousarray(src), ne))
            pos += cs
        out.append(unshuffle_block(np.concatenate(parts), T, h["shuffle"]))
    return np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)


def encode(raw: np.ndarray, typesize: int, clevel: int = 5, shuffle: int = 2, blocksize: int = 256 * 1024) -> np.ndarray:
    """A valid zstd frame with unsplit blocks (what the product writes; any Blosc 1 decoder reads it)."""
    z = _libzstd()
    data = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
    n = data.size
    blocksize = min(blocksize, n) if n else blocksize
    nblocks = -(-n // blocksize) if n else 0
    head = np.zeros(16 + 4 * nblocks, dtype=np.uint8)
    head[0], head[1], head[3] = 2, 1, typesize
    head[2] = (0x4 if shuffle == 2 else (0x1 if shuffle == 1 else 0)) | 0x10 | (4 << 5)
    head[4:8] = np.frombuffer(int(n).to_bytes(4, "little"), dtype=np.uint8)
    head[8:12] = np.frombuffer(int(blocksize).to_bytes(4, "little"), dtype=np.uint8)
    pos, body = head.size, []
    for b in range(nblocks):
        blk = shuffle_block(data[b * blocksize:(b + 1) * blocksize], typesize, shuffle)
        blk = np.ascontiguousarray(blk)
        cap = z.ZSTD_compressBound(blk.size)
        buf = np.empty(cap, dtype=np.uint8)
        cs = z.ZSTD_compress(buf.ctypes.data, cap, blk.ctypes.data, blk.size, 2 * clevel - 1 if clevel < 9 else 19)
        payload = blk if (z.ZSTD_isError(cs) or cs >= blk.size) else buf[:cs]
        head[16 + 4 * b:20 + 4 * b] = np.frombuffer(int(pos).to_bytes(4, "little"), dtype=np.uint8)
        body += [np.frombuffer(int(payload.size).to_bytes(4, "little"), dtype=np.uint8), payload]
        pos += 4 + payload.size
    head[12:16] = np.frombuffer(int(pos).to_bytes(4, "little"), dtype=np.uint8)
    return np.concatenate([head] + body)


# ---------------------------------------------------------------- Zarr v3 regular chunk grid, keys c/<z>/<y>/<x>
def read_array(chunk_dir, shape, chunks, dtype) -> np.ndarray:
    """qi2labDataStore._load_from_zarr_array (:2234-2271) for a 3-D level-0 array: absent chunks read as 0."""
    out = np.zeros(shape, dtype=dtype)
    grid = [-(-s // c) for s, c in zip(shape, chunks)]
    for idx in itertools.product(*[range(g) for g in grid]):
        path = Path(chunk_dir).joinpath(*[str(i) for i in idx])
        if not path.exists():
            continue
        raw = decode(np.fromfile(path, dtype=np.uint8)).view(dtype).reshape(chunks)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sl] = raw[tuple(slice(0, s.stop - s.start) for s in sl)]
    return out


def write_array(chunk_dir, array: np.ndarray, chunks, clevel: int = 5, shuffle: int = 2) -> None:
    """qi2labDataStore._save_to_zarr_array's chunk files (:2273-2373): edge chunks zero padded."""
    shape = array.shape
    grid = [-(-s // c) for s, c in zip(shape, chunks)]
    for idx in itertools.product(*[range(g) for g in grid]):
        chunk = np.zeros(chunks, dtype=array.dtype)
        sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        chunk[tuple(slice(0, s.stop - s.start) for s in sl)] = array[sl]
        path = Path(chunk_dir).joinpath(*[str(i) for i in idx])
        path.parent.mkdir(parents=True, exist_ok=True)
        encode(chunk, array.dtype.itemsize, clevel, shuffle).tofile(path)
