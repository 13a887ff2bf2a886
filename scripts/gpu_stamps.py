"""Scratch probe: read the in-kernel phase stamps of a -DM3D_STAMPS build (scratch_libs/libm3d_stamps.so swapped in
place) after a few RLGC iterations.  usage: gpu_stamps.py Z Y X"""
import ctypes
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
rl = importlib.import_module("merfish3d-analysis_amd.utils.rlgc")
hip = importlib.import_module("merfish3d-analysis_amd._hip")
Z, Y, X = (int(v) for v in sys.argv[1:4])
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
img = torch.poisson(torch.rand((Z, Y, X), device=dev, generator=g) * 200 + 50, generator=g).to(torch.uint16)
zz, yy, xx = torch.meshgrid(*[torch.arange(n, device=dev) - n // 2 for n in (min(64, Z), 51, 51)], indexing="ij")
psf = torch.exp(-0.5 * ((zz / 3.0) ** 2 + (yy / 1.8) ** 2 + (xx / 1.8) ** 2))
psf = (psf / psf.sum()).float()
rl.rlgc_device(img, psf, max_iterations=3)
torch.cuda.synchronize()
n = 512 * 16
buf = (ctypes.c_ulonglong * n)()
fn = hip.lib().m3d_debug_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
assert fn(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 16).astype(np.int64)
a = a[a[:, 0] > 0]
print("blocks with stamps:", len(a))
used = [i for i in range(15) if (a[:, i] > 0).all()]
t0 = a[:, used[0]]
print("stamp ids:", used)
rel = np.stack([a[:, i] - t0 for i in used], axis=1)
print("median cycles since stamp 0:", np.median(rel, axis=0).astype(int).tolist())
print("p10:", np.percentile(rel, 10, axis=0).astype(int).tolist())
print("p90:", np.percentile(rel, 90, axis=0).astype(int).tolist())
# spread of block start times and realtime clock
span = (a[:, used[0]].max() - a[:, used[0]].min())
print("start-time span of the sampled blocks (cycles):", int(span))
