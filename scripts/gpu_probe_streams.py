"""Scratch probe: do two RLGC solves on two streams overlap usefully?  usage: probe [iters]"""
import importlib
import sys
import threading
import time

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
rl = importlib.import_module("merfish3d-analysis_amd.utils.rlgc")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
imgs = [torch.poisson(torch.rand((64, 2048, 2048), device=dev, generator=g) * 200 + 50, generator=g).to(torch.uint16)
        for _ in range(2)]
zz, yy, xx = torch.meshgrid(*[torch.arange(n, device=dev) - n // 2 for n in (64, 51, 51)], indexing="ij")
psf = torch.exp(-0.5 * ((zz / 3.0) ** 2 + (yy / 1.8) ** 2 + (xx / 1.8) ** 2)).float()
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
outs = [torch.empty((64, 2048, 2048), dtype=torch.float32, device=dev) for _ in range(2)]


def solve(i):
    with torch.cuda.stream(streams[i]):
        rl.rlgc_device(imgs[i], psf, max_iterations=iters, out=outs[i])
        streams[i].synchronize()


for i in range(2):
    solve(i)  # plans, warm-up
torch.cuda.synchronize()
t0 = time.time()
for i in range(2):
    solve(i)
torch.cuda.synchronize()
seq = time.time() - t0
t0 = time.time()
th = [threading.Thread(target=solve, args=(i,)) for i in range(2)]
for t in th:
    t.start()
for t in th:
    t.join()
torch.cuda.synchronize()
par = time.time() - t0
print(f"two solves x {iters} iterations: sequential {seq:.3f}s, two streams {par:.3f}s, ratio {par / seq:.3f}", flush=True)
