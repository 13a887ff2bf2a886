"""Scratch probe (not a test): two RLGC solves one after the other on one stream against the same two solves issued from
two host threads on two streams (one plan + workspace per stream, utils/rlgc.py).  usage: probe.py Z Y X [iters]"""
import importlib
import sys
import threading
import time

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
rl = importlib.import_module("merfish3d-analysis_amd.utils.rlgc")
Z, Y, X = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
imgs = [torch.poisson(torch.rand((Z, Y, X), device=dev, generator=g) * 200 + 50, generator=g).to(torch.uint16) for _ in range(2)]
pz = min(64, Z)
zz, yy, xx = torch.meshgrid(*[torch.arange(n, device=dev) - n // 2 for n in (pz, 51, 51)], indexing="ij")
psf = torch.exp(-0.5 * ((zz / 3.0) ** 2 + (yy / 1.8) ** 2 + (xx / 1.8) ** 2))
psf = (psf / psf.sum()).float()
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
outs = [None, None]


def solve(k, n):
    with torch.cuda.stream(streams[k]):
        out, info = rl.rlgc_device(imgs[k], psf, max_iterations=n)
        streams[k].synchronize()
        outs[k] = (out, info)


torch.cuda.synchronize()
for k in range(2):
    solve(k, 2)  # plans, workspaces, setup volumes
torch.cuda.synchronize()
t0 = time.time()
solve(0, iters)
solve(1, iters)
seq = time.time() - t0
ref = [outs[0][0].clone(), outs[1][0].clone()]
t0 = time.time()
th = [threading.Thread(target=solve, args=(k, iters)) for k in range(2)]
for t in th:
    t.start()
for t in th:
    t.join()
torch.cuda.synchronize()
con = time.time() - t0
same = all(torch.equal(ref[k], outs[k][0]) for k in range(2))
its = [outs[k][1].iterations for k in range(2)]
print(f"iterations {its}  one after the other {seq:.3f} s = {1e3 * seq / sum(its):.2f} ms per iteration;  "
      f"two streams {con:.3f} s = {1e3 * con / sum(its):.2f} ms per iteration;  results identical: {same}", flush=True)
