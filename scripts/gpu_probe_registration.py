"""Scratch probe (not a test): full-size timing of register_pair_to_fixed and the SOFIMA estimator.
usage: gpu_probe_registration.py [Z Y X]"""
import importlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
mv = importlib.import_module("merfish3d-analysis_amd.utils.multiview_registration")
sf = importlib.import_module("merfish3d-analysis_amd.utils.sofima_registration")
kernels = importlib.import_module("merfish3d-analysis_amd.kernels")
Z, Y, X = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 2048, 2048)
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
# bead-like fiducial: sparse impulses blurred by a separable Gaussian (device filter)
vol = torch.zeros((Z, Y, X), device=dev)
n = Z * Y * X // 4000
idx = torch.randint(0, Z * Y * X, (n,), device=dev, generator=g)
vol.view(-1)[idx] = torch.rand(n, device=dev, generator=g) * 4000 + 1000
fixed = kernels.gaussian_lowpass(vol, (1.5, 2.0, 2.0)) * 30 + 100
del vol
T = np.eye(4, dtype=np.float32)
T[:3, 3] = (0.31 * 1.0, 0.098 * -3.4, 0.098 * 5.2)
spacing = (0.31, 0.098, 0.098)
moving = mv.warp_array_to_reference_gpu(fixed, transform_zyx_um=T, spacing_zyx_um=spacing, reference_shape=fixed.shape)
torch.cuda.synchronize()
print("volumes ready", tuple(fixed.shape), flush=True)
for rep in range(2):
    t0 = time.time()
    est = mv.register_pair_to_fixed(fixed, moving, spacing_zyx_um=spacing)
    torch.cuda.synchronize()
    print(f"register_pair_to_fixed {time.time() - t0:.2f}s  shift_px={np.asarray(est)[:3, 3] / np.asarray(spacing)}", flush=True)
warped = mv.warp_array_to_reference_gpu(moving, transform_zyx_um=np.asarray(est), spacing_zyx_um=spacing,
                                        reference_shape=fixed.shape)
for rep in range(2):
    t0 = time.time()
    flow, md = sf.estimate_sofima_flow_field_xyz_px(fixed, warped)
    torch.cuda.synchronize()
    print(f"estimate_sofima_flow_field_xyz_px {time.time() - t0:.2f}s status={md['status']} valid={md['valid_flow_vectors']} "
          f"mesh_iters={md['mesh_iterations']} grid={flow.shape} |flow|max={np.abs(flow).max():.3f} "
          f"nmse {md.get('affine_normalized_mse', 0):.4f}->{md.get('sofima_normalized_mse', 0):.4f}", flush=True)
t0 = time.time()
raw = sf.patch_flow_field(warped, fixed, patch_size=(10, 32, 32), step=(5, 16, 16))
torch.cuda.synchronize()
print(f"patch_flow_field alone {time.time() - t0:.2f}s", flush=True)
print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
