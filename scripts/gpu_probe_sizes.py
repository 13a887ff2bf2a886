"""Scratch probe (not a test): RLGC iteration time off the headline size.  For every (Z, Y, X) on the command line
(comma-separated triples) prints the padded FFT volume, which kernel family ran the X / Y passes, the iteration's device
time (HIP events around the solver loop, 8 iterations) and SURVEY section 8(d)'s figure 148 B x padded voxels / time
against the 8 TB/s peak, next to the headline size's cost per padded voxel.
usage: gpu_probe_sizes.py 64,2048,2048 64,1920,1920 32,2048,2048 64,1024,1024 ..."""
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
rl = importlib.import_module("merfish3d-analysis_amd.utils.rlgc")
kernels = importlib.import_module("merfish3d-analysis_amd.kernels")
dev = "cuda:0"
rows = []
for arg in sys.argv[1:]:
    Z, Y, X = (int(v) for v in arg.split(","))
    g = torch.Generator(device=dev).manual_seed(0)
    # the bench's kind of image: background 100, blurred point emitters of 300-900 peak counts, shot noise
    n_em = max(50, Z * Y * X // 3400)
    vol = torch.zeros((Z, Y, X), device=dev)
    idx = torch.randint(0, Z * Y * X, (n_em,), device=dev, generator=g)
    vol.view(-1).index_put_((idx,), (torch.rand(n_em, device=dev, generator=g) * 600 + 300) / 0.0138, accumulate=True)
    kernels.gaussian_lowpass(vol, (1.8, 1.6, 1.6), True, out=vol)
    img = torch.poisson(vol + 100.0, generator=g).clamp_(0, 65535).to(torch.uint16)
    del vol
    pz = min(64, Z)
    zz, yy, xx = torch.meshgrid(*[torch.arange(n, device=dev) - n // 2 for n in (pz, 51, 51)], indexing="ij")
    psf = torch.exp(-0.5 * ((zz / 3.0) ** 2 + (yy / 1.8) ** 2 + (xx / 1.8) ** 2))
    psf = (psf / psf.sum()).float()
    rl.rlgc_device(img, psf, max_iterations=1)
    torch.cuda.synchronize()
    out, info = rl.rlgc_device(img, psf, max_iterations=8, profile="loop")
    torch.cuda.synchronize()
    npad = float(np.prod(info.padded_shape))
    ms = info.loop_ms / max(info.iterations, 1)
    gbs = 148.0 * npad / (ms * 1e-3) / 1e9
    rows.append((arg, tuple(info.padded_shape), info.iterations, ms, gbs, ms * 1e6 / npad))
    print(f"{arg:>15s}  padded {'x'.join(map(str, info.padded_shape)):>14s}  {info.iterations} iterations  "
          f"{ms:8.3f} ms / iteration  {gbs:7.0f} GB/s = {gbs / 8000:5.3f} of peak  {ms * 1e6 / npad:6.3f} ns per padded voxel",
          flush=True)
    del img, out
    rl.clear_rlgc_caches(clear_memory_pool=True)
    torch.cuda.empty_cache()
base = rows[0][5]
for arg, padded, it, ms, gbs, ns in rows:
    print(f"{arg}: {ns / base:5.2f} x the first size's cost per padded voxel")
