"""Scratch probe (not a test): does the placement of the RLGC workspace in HBM change the strided passes' speed?
One process, many trials: ballast of 1-GiB tensors before / after the workspace, workspace re-allocated each trial,
Z / X-inverse family time per iteration from the solver's own HIP events (profile=True).
usage: gpu_probe_placement.py [trials]"""
import importlib
import os
import random
import sys

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
rl = importlib.import_module("merfish3d-analysis_amd.utils.rlgc")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
Z, Y, X = 64, 2048, 2048
img = torch.poisson(torch.rand((Z, Y, X), device=dev, generator=g) * 60 + 70, generator=g).to(torch.uint16)
zz, yy, xx = torch.meshgrid(*[torch.arange(n, device=dev) - n // 2 for n in (64, 51, 51)], indexing="ij")
psf = torch.exp(-0.5 * ((zz / 3.0) ** 2 + (yy / 1.8) ** 2 + (xx / 1.8) ** 2))
psf = (psf / psf.sum()).float()
out = torch.empty((Z, Y, X), dtype=torch.float32, device=dev)
rnd = random.Random(int(os.environ.get("M3D_PROBE_SEED", "1")))
plan_of = lambda: next(iter(rl._plan_cache.values()))
for t in range(trials):
    pre = 0 if t == 0 else rnd.choice([0, 40, 80, 120, 160, 200, 220])
    rl.clear_rlgc_caches(clear_memory_pool=True)
    ballast = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(pre)]
    rl.rlgc_device(img, psf, max_iterations=1, out=out, rng_seed=1)
    ptr = plan_of().workspace.data_ptr()
    _, info = rl.rlgc_device(img, psf, max_iterations=4, out=out, rng_seed=1, profile=True)
    it = max(info.iterations, 1)
    free, total = torch.cuda.mem_get_info()
    print(f"trial {t:2d} ballast before workspace {pre:3d} GiB  workspace at 0x{ptr:012x} (mod 1 GiB {ptr % (1 << 30) >> 20:4d} MiB)"
          f"  free {free / 2**30:6.1f} GiB  per iteration: x_fwd {info.kernel_ms[0] / it:.3f} y_fwd {info.kernel_ms[1] / it:.3f}"
          f" z {info.kernel_ms[2] / it:.3f} y_inv {info.kernel_ms[3] / it:.3f} x_inv {info.kernel_ms[4] / it:.3f} ms", flush=True)
    del ballast
