"""Scratch probe (not a test): the fused decode kernel alone.  usage: gpu_probe_decode.py Z Y X [B] [reps]
Random low-passed-like stack, 140-codeword HW4 codebook; prints event-timed ms per launch, TFLOP/s and GB/s."""
import importlib
import sys

import numpy as np
import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
sys.argv_saved, sys.argv = sys.argv, ["bench.py"]
import bench  # noqa: E402

sys.argv = sys.argv_saved
kernels = importlib.import_module("merfish3d-analysis_amd.kernels")
Z, Y, X = (int(v) for v in sys.argv[1:4])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 22
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = "cuda:0"
cb = bench.hw4_codebook(B)
mat = cb.iloc[:, 1:].to_numpy(dtype=np.float64)
mat = mat / np.linalg.norm(mat, axis=1, keepdims=True)
g = torch.Generator(device=dev).manual_seed(0)
stack = torch.empty((B, Z, Y, X), dtype=torch.float32, device=dev)
for b in range(B):   # gamma-like background with sparse bright voxels, as a low-passed readout x predictor stack
    stack[b] = torch.rand((Z, Y, X), device=dev, generator=g).pow_(6).mul_(400.0)
bk = torch.full((B,), 5.0, device=dev)
nm = torch.full((B,), 200.0, device=dev)
cbt = torch.from_numpy(mat.astype(np.float32)).to(dev)
out = kernels.decode_pixels(stack, cbt, 0.7653669, (0.9, 10.0), bk, nm)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(reps):
    out = kernels.decode_pixels(stack, cbt, 0.7653669, (0.9, 10.0), bk, nm)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
n = Z * Y * X
flop = 2.0 * B * mat.shape[0] * n
print(f"k_decode {Z}x{Y}x{X} B={B} C={mat.shape[0]}: {ms:.3f} ms  {flop / ms / 1e9:.2f} TFLOP/s  "
      f"{(4.0 * B + 6.0 + 2.0 * B) * n / ms / 1e6:.1f} GB/s  decoded>=0: {int((out[0] >= 0).sum())}", flush=True)
