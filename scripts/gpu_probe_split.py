"""Scratch probe (not a test): RLGC iterations on a 64x2048x2048 uint16 image of a given mean count, so that
rocprofv3's kernel stats show what k_split_ratio (the binomial photon split) costs at that brightness.
usage: gpu_probe_split.py MEAN_COUNT [iters]"""
import importlib
import sys

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
rl = importlib.import_module("merfish3d-analysis_amd.utils.rlgc")
mean = float(sys.argv[1])
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
shape = (64, 2048, 2048)
# a smooth illumination profile (0.8 .. 1.2 of the mean) + shot noise, like a camera frame of that exposure
yy = torch.linspace(0, 3.14159, shape[1], device=dev)[None, :, None]
xx = torch.linspace(0, 3.14159, shape[2], device=dev)[None, None, :]
lam = mean * (0.8 + 0.4 * torch.sin(yy) * torch.sin(xx)).expand(shape)
img = torch.poisson(lam.contiguous(), generator=g).clamp_(0, 65535).to(torch.uint16)
del lam
zz, y2, x2 = torch.meshgrid(*[torch.arange(n, device=dev) - n // 2 for n in (64, 51, 51)], indexing="ij")
psf = torch.exp(-0.5 * ((zz / 3.0) ** 2 + (y2 / 1.8) ** 2 + (x2 / 1.8) ** 2))
psf = (psf / psf.sum()).float()
out, info = rl.rlgc_device(img, psf, max_iterations=1)
torch.cuda.synchronize()
out, info = rl.rlgc_device(img, psf, max_iterations=iters, profile=True)
torch.cuda.synchronize()
print(f"mean {mean:g}: image mean {float(img.float().mean()):.1f} max {int(img.to(torch.int32).max())}, {info.iterations} iterations, "
      f"loop {info.loop_ms / max(info.iterations, 1):.3f} ms per iteration", flush=True)
