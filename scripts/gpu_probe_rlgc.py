"""Scratch probe (not a test): RLGC timing on the GPU box.  usage: probe.py Z Y X [iters]"""
import importlib
import sys
import time

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
rl = importlib.import_module("merfish3d-analysis_amd.utils.rlgc")
Z, Y, X = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
# M3D_PROBE_PREALLOC_GB: hold that much HBM in 1-GiB tensors before the plan's workspace is allocated (the bench's state)
ballast = [torch.empty(1 << 30, dtype=torch.uint8, device=dev)
           for _ in range(int(__import__("os").environ.get("M3D_PROBE_PREALLOC_GB", "0")))]
if __import__("os").environ.get("M3D_PROBE_IMAGE", "noise") == "emitters":
    # the bench's kind of image: background 100, blurred point emitters of 300-900 peak counts, shot noise
    kernels = importlib.import_module("merfish3d-analysis_amd.kernels")
    n_em = max(50, Z * Y * X // 3400)
    vol = torch.zeros((Z, Y, X), device=dev)
    idx = torch.randint(0, Z * Y * X, (n_em,), device=dev, generator=g)
    vol.view(-1).index_put_((idx,), (torch.rand(n_em, device=dev, generator=g) * 600 + 300) / 0.0138, accumulate=True)
    kernels.gaussian_lowpass(vol, (1.8, 1.6, 1.6), True, out=vol)
    img = torch.poisson(vol + 100.0, generator=g).clamp_(0, 65535).to(torch.uint16)
    del vol
else:
    img = torch.poisson(torch.rand((Z, Y, X), device=dev, generator=g) * 200 + 50, generator=g).to(torch.uint16)
pz = min(64, Z)
zz, yy, xx = torch.meshgrid(*[torch.arange(n, device=dev) - n // 2 for n in (pz, 51, 51)], indexing="ij")
psf = torch.exp(-0.5 * ((zz / 3.0) ** 2 + (yy / 1.8) ** 2 + (xx / 1.8) ** 2))
psf = (psf / psf.sum()).float()
t0 = time.time()
out, info = rl.rlgc_device(img, psf, max_iterations=1)
torch.cuda.synchronize()
print(f"first call (plan + 1 iter): {time.time() - t0:.2f}s padded={info.padded_shape}", flush=True)
# M3D_PROBE_POSTALLOC_GB: the same ballast, but allocated AFTER the plan's workspace
ballast += [torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            for _ in range(int(__import__("os").environ.get("M3D_PROBE_POSTALLOC_GB", "0")))]
for n in (iters, 2 * iters):
    torch.cuda.synchronize()
    t0 = time.time()
    out, info = rl.rlgc_device(img, psf, max_iterations=n, profile=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    per = [round(v / max(info.conv_calls, 1), 3) for v in info.kernel_ms]
    print(f"iters={info.iterations} stop={info.stop_reason} wall={dt:.3f}s conv_avg_ms="
          f"{info.conv_ms_total / max(info.conv_calls, 1):.3f} kernels(xf,yf,z,yi,xi)={per}", flush=True)
print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
