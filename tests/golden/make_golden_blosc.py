"""Golden Blosc 1 frames from the real c-blosc (1.21, through imagecodecs), for the datastore I/O path.

Run with an interpreter that has imagecodecs - in the build container
``/opt/conda/bin/python3.9 tests/golden/make_golden_blosc.py`` - to write
``tests/golden/blosc_frames.npz``:

* ``raw_i`` / ``frame_i``: data and the frame c-blosc made of it (zstd / lz4; bit, byte and no
  shuffle; typesize 2 and 4; several blocks with a short last block whose element count is not a
  multiple of eight; split and unsplit blocks; incompressible and tiny buffers) - what
  ``m3d_blosc_decode_host`` / ``m3d_zarr_read_*`` must decode;
* ``ours_i``: the frame ``m3d_blosc_encode_host`` (this repository's libm3d_hip.so, loaded with
  ctypes) made of ``raw_i``, kept only after c-blosc's own ``blosc_decompress`` returned ``raw_i``
  from it - the direction "the reference reads what we write".
"""

import ctypes
from pathlib import Path

import imagecodecs
import numpy as np

here = Path(__file__).resolve().parent
LIB = here.parents[1] / "merfish3d-analysis_amd" / "csrc" / "libm3d_hip.so"


def image_like(rng, n, dtype):
    x = np.linspace(0, 40, n)
    base = 300 + 200 * np.sin(x) + 50 * np.sin(7.3 * x)
    noisy = rng.poisson(np.clip(base, 1, None)).astype(np.float64)
    if np.dtype(dtype) == np.float32:
        return (noisy / 1000.0).astype("<f4")
    return noisy.astype(dtype)


def main():
    rng = np.random.default_rng(3)
    cases = [
        # name, dtype, elements, compressor, shuffle (0 none, 1 byte, 2 bit), blocksize (None = automatic)
        ("zstd_bit_u16_auto", "<u2", 100000, "zstd", 2, None),
        ("zstd_bit_u16_blocks", "<u2", 4096 * 3 + 1003, "zstd", 2, 8192),
        ("zstd_bit_f32_blocks", "<f4", 4096 * 2 + 517, "zstd", 2, 16384),
        ("lz4_bit_u16_split", "<u2", 4096 * 5 + 80, "lz4", 2, None),
        ("lz4_bit_u16_odd", "<u2", 4096 * 5 + 77, "lz4", 2, None),
        ("lz4_byte_f32_blocks", "<f4", 2048 * 4 + 33, "lz4", 1, 8192),
        ("zstd_none_u16", "<u2", 30000, "zstd", 0, None),
        ("zstd_byte_u16", "<u2", 30001, "zstd", 1, 16384),
        ("zstd_bit_u16_random", "<u2", 20000, "zstd", 2, 8192),
        ("zstd_bit_u16_tiny", "<u2", 50, "zstd", 2, None),
        ("zstd_bit_u8", "u1", 70001, "zstd", 2, 32768),
    ]
    lib = ctypes.CDLL(str(LIB))
    lib.m3d_blosc_bound.restype = ctypes.c_size_t
    lib.m3d_blosc_bound.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
    lib.m3d_blosc_encode_host.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                          ctypes.POINTER(ctypes.c_size_t)]
    out = {"names": np.array([c[0] for c in cases]), "imagecodecs_version": np.array(imagecodecs.__version__),
           "blosc_version": np.array(str(imagecodecs.blosc_version()))}
    for i, (name, dtype, n, comp, shuffle, blocksize) in enumerate(cases):
        raw = image_like(rng, n, dtype) if "random" not in name else rng.integers(0, 65536, n).astype(dtype)
        ts = np.dtype(dtype).itemsize
        frame = imagecodecs.blosc_encode(raw.tobytes(), level=5, compressor=comp, shuffle=shuffle, typesize=ts,
                                         blocksize=blocksize, numthreads=1)
        assert imagecodecs.blosc_decode(frame) == raw.tobytes()
        out[f"raw_{i}"] = raw
        out[f"frame_{i}"] = np.frombuffer(frame, dtype=np.uint8)
        # our encoder, checked by the real decoder
        cap = lib.m3d_blosc_bound(raw.nbytes, blocksize or 0)
        buf = np.zeros(cap, dtype=np.uint8)
        got = ctypes.c_size_t(0)
        rc = lib.m3d_blosc_encode_host(raw.ctypes.data, raw.nbytes, ts, 5, shuffle, 4 if comp == "zstd" else 1,
                                       blocksize or 0, buf.ctypes.data, cap, ctypes.byref(got))
        assert rc == 0, (name, rc)
        ours = buf[: got.value].copy()
        assert imagecodecs.blosc_decode(ours.tobytes()) == raw.tobytes(), name
        out[f"ours_{i}"] = ours
        print(name, "nbytes", raw.nbytes, "c-blosc", len(frame), "flags", hex(frame[2]), "blocksize",
              int.from_bytes(frame[8:12], "little"), "| ours", got.value, "decoded by c-blosc: ok")
    np.savez_compressed(here / "blosc_frames.npz", **out)


if __name__ == "__main__":
    main()
