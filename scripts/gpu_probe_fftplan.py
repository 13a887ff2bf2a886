"""Scratch probe: rocFFT c2c plan creation + first execution time for overlap-crop shapes."""
import importlib
import sys
import time

import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
kernels = importlib.import_module("merfish3d-analysis_amd.kernels")
dev = torch.device("cuda:0")
for shape in [(64, 2040, 2041), (64, 2044, 2037), (63, 2045, 2043), (64, 2040, 2041), (2048, 2048), (2045, 2043)]:
    t0 = time.time()
    plan = kernels.FFTPlan(shape, dev)
    t1 = time.time()
    x = torch.zeros(shape, dtype=torch.complex64, device=dev)
    plan.execute(x)
    torch.cuda.synchronize()
    t2 = time.time()
    plan.execute(x)
    torch.cuda.synchronize()
    t3 = time.time()
    print(shape, f"create {t1 - t0:.2f}s first exec {t2 - t1:.2f}s second exec {t3 - t2:.3f}s work {plan.work_bytes / 1e9:.2f} GB", flush=True)
    plan.close()
