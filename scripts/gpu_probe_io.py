#!/usr/bin/env python3
"""Throughput of the datastore image I/O path on one tile image (64x2048x2048 uint16 by default):
device path (zstd on host threads, shuffle + scatter on the GPU) beside the all-host path of the same library.
Usage: gpu_probe_io.py [Z Y X] [threads]"""
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import importlib

zio = importlib.import_module("merfish3d-analysis_amd.zarr_io")

shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 2048, 2048)
threads = int(sys.argv[4]) if len(sys.argv) >= 5 else None
rng = np.random.default_rng(0)
# image-like content: smooth background + Poisson noise + sparse bright spots (uint16)
z, y, x = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]), indexing="ij", sparse=True)
base = 110 + 20 * np.sin(0.01 * x) * np.cos(0.013 * y) + 0.2 * z
vol = rng.poisson(np.broadcast_to(base, shape)).astype(np.uint16)
idx = tuple(rng.integers(0, s, size=80000) for s in shape)
vol[idx] += rng.integers(300, 900, size=80000).astype(np.uint16)
dev = torch.from_numpy(vol).to("cuda:0")
out = {"shape": shape, "raw_MB": vol.nbytes / 1e6}
root = Path(tempfile.mkdtemp(prefix="m3d_io_", dir="/dev/shm" if Path("/dev/shm").exists() else None))
with zio.ImageIO(num_threads=threads) as io:
    out["threads"] = io.num_threads
    for name, src in (("write_device", dev), ("write_host", vol)):
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            io.write_image(root / name, src)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[name] = {"s": round(best, 3), "GB_per_s": round(vol.nbytes / best / 1e9, 2),
                     "ratio": round(io.last_stats["raw_bytes"] / io.last_stats["file_bytes"], 2),
                     "entropy_thread_s": round(io.last_stats["seconds_entropy"], 2)}
        print(name, out[name], flush=True)
    for name, device in (("read_device", "cuda:0"), ("read_host", None)):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            got = io.read_image(root / "write_device", device=device)
            if device:
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        ok = torch.equal(got, dev) if device else np.array_equal(got, vol)
        out[name] = {"s": round(best, 3), "GB_per_s": round(vol.nbytes / best / 1e9, 2), "equal": bool(ok),
                     "entropy_thread_s": round(io.last_stats["seconds_entropy"], 2)}
        print(name, out[name], flush=True)
    t0 = time.perf_counter()
    torch.from_numpy(vol).to("cuda:0")
    torch.cuda.synchronize()
    out["pageable_h2d_s"] = round(time.perf_counter() - t0, 3)
import shutil

shutil.rmtree(root, ignore_errors=True)
print(json.dumps(out))
