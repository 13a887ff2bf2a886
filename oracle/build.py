"""Build + load the plain-C half of the oracle (TEST INFRASTRUCTURE ONLY)."""

from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

_HERE = Path(__file__).resolve().parent
_SRC = _HERE / "c" / "m3d_oracle.c"
_OUT = _HERE / "_build" / "libm3d_oracle.so"
_lib = None


def build(force: bool = False) -> Path:
    """gcc the C restatement into oracle/_build/ (kept out of git, shipped by gpurun)."""
    if not force and _OUT.exists() and _OUT.stat().st_mtime >= _SRC.stat().st_mtime:
        return _OUT
    _OUT.parent.mkdir(parents=True, exist_ok=True)
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(_OUT), str(_SRC), "-lm"]
    with open("/proc/cpuinfo") as fh:
        if " fma " in fh.read():
            cmd.insert(1, "-mfma")
    subprocess.run(cmd, check=True)
    return _OUT


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(str(build()))
        _lib.m3d_oracle_distance_argmin.restype = None
        _lib.m3d_oracle_distance_argmin.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_void_p,
        ]
        _lib.m3d_oracle_label_equal.restype = ctypes.c_int64
        _lib.m3d_oracle_label_equal.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
            ctypes.c_int16, ctypes.c_int, ctypes.c_void_p,
        ]
    return _lib
