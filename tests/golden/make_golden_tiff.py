#!/usr/bin/env python3
"""Pin the package's baseline TIFF codec (simulation.read_tiff / write_tiff) against the real tifffile.

The main interpreter has no tifffile; the image's conda Python (``/opt/conda/bin/python3.9``) holds tifffile 2021.7.2
(the interpreter tests/golden/make_golden_skimage.py and make_golden_blosc.py already use).  Run in the authoring
container:

    python tests/golden/make_golden_tiff.py

For every case (the layouts the simulation path meets: the flat multi-page uint16 stack of ``aligned_1.tiff``, the
(channel, z, y, x) stack ``convert_simulation`` writes per round, a single page, float32 and uint8 stacks, a page
large enough to be cut into several strips, BigTIFF, a big-endian file):
  1. tifffile WRITES the array -> the file's bytes go into the fixture; the CPU test reads them with ``read_tiff``;
  2. ``write_tiff`` writes the array -> tifffile READS it back here (asserted equal, dtype and shape included) and the
     file's bytes go into the fixture; the CPU test asserts that ``write_tiff`` still produces exactly those bytes.
Output: tests/golden/tiff_codec.npz."""

from __future__ import annotations

import importlib
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))
CONDA = "/opt/conda/bin/python3.9"

HELPER = r"""
import sys, numpy as np, tifffile
work = sys.argv[1]
cases = np.load(work + "/cases.npz")
out = {"tifffile_version": np.asarray(tifffile.__version__)}
for name in cases.files:
    a = cases[name]
    kw = {}
    if name.endswith("_bigtiff"):
        kw["bigtiff"] = True
    if name.endswith("_bigendian"):
        kw["byteorder"] = ">"
    tifffile.imwrite(work + "/tf_" + name + ".tif", a, **kw)
    back = tifffile.imread(work + "/ours_" + name + ".tif")
    assert back.dtype == a.dtype and back.shape == a.shape and np.array_equal(back, a), name
    out["read_of_ours_" + name] = back
np.savez(work + "/helper_out.npz", **out)
print("tifffile", tifffile.__version__, "read", len(cases.files), "files written by write_tiff")
"""


def cases() -> dict[str, np.ndarray]:
    rng = np.random.default_rng(11)
    smooth = (np.add.outer(np.arange(300), np.arange(420)) % 997).astype(np.uint16)
    return {
        "flat_rounds_channels_z": rng.integers(0, 4000, size=(2 * 3 * 4, 20, 24)).astype(np.uint16),   # aligned_1.tiff
        "round_stack_czyx": rng.integers(0, 65535, size=(3, 4, 20, 24)).astype(np.uint16),            # data_rNNNN_tileNNNN.tif
        "five_dimensional": rng.integers(0, 900, size=(2, 3, 2, 10, 12)).astype(np.uint16),
        "single_page": rng.integers(0, 65535, size=(33, 47)).astype(np.uint16),
        "float32_stack": rng.normal(size=(3, 9, 11)).astype(np.float32),
        "uint8_stack": rng.integers(0, 255, size=(5, 8, 6)).astype(np.uint8),
        "several_strips": np.stack([smooth, smooth[::-1]]),                                            # 252 KB per page
        "stack_bigtiff": rng.integers(0, 4000, size=(4, 16, 16)).astype(np.uint16),
        "stack_bigendian": rng.integers(0, 4000, size=(3, 6, 7)).astype(np.uint16),
    }


def main() -> None:
    sim = importlib.import_module("merfish3d-analysis_amd.simulation")
    arrays = cases()
    out: dict[str, np.ndarray] = {}
    with tempfile.TemporaryDirectory(prefix="m3d_tiff_golden_") as work:
        np.savez(work + "/cases.npz", **arrays)
        for name, a in arrays.items():
            sim.write_tiff(f"{work}/ours_{name}.tif", a)
        (Path(work) / "helper.py").write_text(HELPER)
        r = subprocess.run([CONDA, work + "/helper.py", work], capture_output=True, text=True)
        print(r.stdout.strip(), r.stderr.strip()[-2000:])
        r.check_returncode()
        helper = np.load(work + "/helper_out.npz")
        out["tifffile_version"] = helper["tifffile_version"]
        out["cases"] = np.asarray(list(arrays))
        for name, a in arrays.items():
            out["array_" + name] = a
            out["tifffile_file_" + name] = np.fromfile(f"{work}/tf_{name}.tif", dtype=np.uint8)
            out["ours_file_" + name] = np.fromfile(f"{work}/ours_{name}.tif", dtype=np.uint8)
            # and the reader on tifffile's files, checked here too so that a broken fixture is never written
            back = sim.read_tiff(f"{work}/tf_{name}.tif")
            assert back.dtype == a.dtype and back.shape == a.shape and np.array_equal(back, a), name
    np.savez_compressed(HERE / "tiff_codec.npz", **out)
    print("wrote", HERE / "tiff_codec.npz", (HERE / "tiff_codec.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
