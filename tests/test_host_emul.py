"""CPU: the header-only helpers of the HIP kernels (Philox split, boundary index maps)
compiled for the host and checked against the NumPy oracle."""

import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import philox

HERE = Path(__file__).resolve().parent


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    out = tmp_path_factory.mktemp("emul") / "libemul.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(out),
                    str(HERE / "host_emul" / "host_emul.cpp")], check=True)
    lib = ctypes.CDLL(str(out))
    vp = ctypes.c_void_p
    lib.emul_binomial_half.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64, vp]
    lib.emul_sym_src.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    lib.emul_reflect.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    return lib


def test_philox_binomial_matches_numpy_oracle(emul):
    rng = np.random.default_rng(5)
    n = np.concatenate([np.arange(0, 300), rng.integers(0, 70000, size=2000), rng.integers(0, 1 << 20, size=500),
                        [65535, 128, 127, 129, 256, 255, 63, 64, 65471, 65472, 65473, 130944, 130945, 1 << 20]]).astype(np.uint32)
    vox = rng.integers(0, 2**40, size=n.size).astype(np.uint64)
    out = np.zeros(n.size, dtype=np.uint32)
    for it, seed in ((0, 42), (7, 43), (123, 2**40 + 17)):
        emul.emul_binomial_half(n.ctypes.data, vox.ctypes.data, n.size, it, seed, out.ctypes.data)
        ref = philox.binomial_half(n.astype(np.int64), vox.astype(np.int64), it, seed)
        assert np.array_equal(out.astype(np.int64), ref)


@pytest.mark.parametrize("L,b,a", [(5, 3, 4), (2, 5, 7), (1, 3, 3), (16, 8, 8), (2048, 69, 70)])
def test_symmetric_source_matches_np_pad(emul, L, b, a):
    n = L + b + a
    out = np.zeros(n, dtype=np.int32)
    emul.emul_sym_src(n, b, L, out.ctypes.data)
    ref = np.pad(np.arange(b, b + L), (b, a), mode="symmetric")
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 64])
def test_reflect_matches_scipy_reflect(emul, n):
    lo, hi = -3 * n - 2, 4 * n + 3
    out = np.zeros(hi - lo, dtype=np.int32)
    emul.emul_reflect(lo, hi, n, out.ctypes.data)
    ref = np.pad(np.arange(n), (-lo, hi - n), mode="symmetric")
    assert np.array_equal(out, ref)
