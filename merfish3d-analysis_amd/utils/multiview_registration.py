"""Fiducial registration + resampling on MI355X (mirror of utils/multiview_registration.py).

Same function names / arguments / conventions as the reference: images are Z,Y,X; a 4x4
transform is homogeneous in physical Z,Y,X microns and maps *reference/output* coordinates
to *moving/input* coordinates (:814-817).  NumPy arrays in -> NumPy arrays out like the
reference; torch tensors in HBM in -> tensors out with no PCIe traffic.  The device work is
``m3d_max_z``, ``m3d_affine_warp``, ``m3d_flow_warp`` and the phase-correlation kernels of
``libm3d_hip.so``.  The multiview-stitcher adapters of the reference (:130-205, :574-794)
have no hot-path caller and are out of scope.
"""

from __future__ import annotations

import timeit
from collections.abc import Sequence
from typing import Any

import numpy as np

LOCAL_ROUND_TRANSFORM_KEY = "local_round_registered"


def _diag(message: str, *, enabled: bool) -> None:
    if enabled:
        print(f"[multiview-registration] {message}", flush=True)


def _to_device(array, gpu_id: int = 0):
    """(tensor in HBM, was_numpy).  uint16 / float32 keep their dtype, others -> float32."""
    import torch

    if isinstance(array, torch.Tensor):
        t = array if array.is_cuda else array.to(f"cuda:{gpu_id}")
        was_numpy = False
    else:
        arr = np.ascontiguousarray(array)
        if arr.dtype not in (np.uint16, np.float32):
            arr = arr.astype(np.float32)
        t = torch.from_numpy(arr).to(f"cuda:{gpu_id}")
        was_numpy = True
    if t.dtype not in (torch.uint16, torch.float32):
        t = t.to(torch.float32)
    return t.contiguous(), was_numpy


def _max_z_projection_gpu(image, cp: Any = None):
    """Maximum projection over Z as a float32 array in HBM (:57-77)."""
    from .. import kernels

    t, _ = _to_device(image)
    return kernels.max_z_projection(t)


def _overlap_slices_after_translation(shape: Sequence[int], translation_px: Sequence[float]):
    """Output slices whose translated coordinates stay inside the input (:80-109)."""
    slices = []
    for axis_size, t in zip(shape, translation_px, strict=True):
        start = int(np.ceil(max(0.0, -float(t))))
        stop = int(np.floor(min(float(axis_size), float(axis_size) - t)))
        if stop <= start:
            return None
        slices.append(slice(start, stop))
    return tuple(slices)


def _zyx_dict(values: Sequence[float]) -> dict[str, float]:
    values = np.asarray(values, dtype=np.float32).tolist()
    return {"z": float(values[0]), "y": float(values[1]), "x": float(values[2])}


def pixel_affine_from_physical(transform_zyx_um, spacing_zyx_um, origin_zyx_um=(0.0, 0.0, 0.0)):
    """matrix_px, offset_px handed to affine_transform (:861-868), float32 like the reference."""
    spacing = np.asarray(spacing_zyx_um, dtype=np.float32)
    origin = np.asarray(origin_zyx_um, dtype=np.float32)
    transform = np.asarray(transform_zyx_um, dtype=np.float32)
    linear_um = transform[:3, :3]
    translation_um = transform[:3, 3]
    matrix_px = (linear_um * spacing[np.newaxis, :]) / spacing[:, np.newaxis]
    offset_px = (linear_um @ origin + translation_um - origin) / spacing
    return matrix_px, offset_px


def warp_array_to_reference_gpu(
    image,
    *,
    transform_zyx_um: np.ndarray,
    spacing_zyx_um: Sequence[float],
    reference_shape: Sequence[int],
    reference_origin_zyx_um: Sequence[float] = (0.0, 0.0, 0.0),
    mode: str = "constant",
    cval: float = 0.0,
    order: int = 1,
    gpu_id: int = 0,
    z_batch_size: int = 4,
    diagnostics: bool = False,
):
    """Trilinear affine resample into the reference grid (:797-902); dtype follows the input."""
    from .. import kernels

    if order != 1:
        raise NotImplementedError("only order=1 (the hot path's setting) is implemented")
    matrix_px, offset_px = pixel_affine_from_physical(transform_zyx_um, spacing_zyx_um,
                                                      reference_origin_zyx_um)
    start = timeit.default_timer()
    t, was_numpy = _to_device(image, gpu_id)
    if t.dim() != 3:
        raise ValueError(f"Expected a 3D image, got shape {tuple(t.shape)!r}.")
    out = kernels.affine_warp(t, matrix_px, offset_px, reference_shape, mode=mode, cval=cval)
    _diag(f"warp_array_to_reference_gpu_done elapsed_s={timeit.default_timer() - start:.2f}",
          enabled=diagnostics)
    return out.cpu().numpy() if was_numpy else out


def warp_array_to_reference_with_affine_and_sofima_flow_gpu(
    image,
    *,
    transform_zyx_um: np.ndarray,
    spacing_zyx_um: Sequence[float],
    reference_shape: Sequence[int],
    sofima_flow_field_xyz_px,
    flow_field_stride_zyx_px: Sequence[float],
    flow_field_box_start_xyz_px: Sequence[float],
    reference_origin_zyx_um: Sequence[float] = (0.0, 0.0, 0.0),
    mode: str = "constant",
    cval: float = 0.0,
    order: int = 1,
    gpu_id: int = 0,
    z_batch_size: int = 4,
    diagnostics: bool = False,
):
    """Affine composed with a SOFIMA flow field, image sampled once (:905-1131)."""
    import torch

    from .. import kernels

    if order != 1:
        raise NotImplementedError("only order=1 (the hot path's setting) is implemented")
    if len(reference_shape) != 3:
        raise ValueError("reference_shape must have three ZYX elements.")
    t, was_numpy = _to_device(image, gpu_id)
    if t.dim() != 3:
        raise ValueError(f"Expected a 3D image, got shape {tuple(t.shape)!r}.")
    flow = sofima_flow_field_xyz_px
    if not isinstance(flow, torch.Tensor):
        flow = torch.from_numpy(np.ascontiguousarray(flow, dtype=np.float32))
    flow = flow.to(device=t.device, dtype=torch.float32)
    if flow.dim() != 4:
        raise ValueError("sofima_flow_field_xyz_px must have channel plus ZYX axes.")
    if flow.shape[0] != 3 and flow.shape[-1] == 3:
        flow = flow.movedim(-1, 0)
    if flow.shape[0] != 3:
        raise ValueError("SOFIMA flow field must have three XYZ channels.")
    box_start_xyz = np.asarray(flow_field_box_start_xyz_px, dtype=np.float32)
    start = timeit.default_timer()
    out = kernels.flow_warp(
        t, flow.contiguous(), np.asarray(flow_field_stride_zyx_px, dtype=np.float32),
        box_start_xyz[[2, 1, 0]], np.asarray(spacing_zyx_um, dtype=np.float32),
        np.asarray(reference_origin_zyx_um, dtype=np.float32),
        np.asarray(transform_zyx_um, dtype=np.float32), reference_shape, mode=mode, cval=cval)
    _diag("warp_array_to_reference_with_affine_and_sofima_flow_gpu_done "
          f"elapsed_s={timeit.default_timer() - start:.2f}", enabled=diagnostics)
    return out.cpu().numpy() if was_numpy else out


def transform_points_to_reference(
    points_zyx: np.ndarray,
    *,
    transform_zyx_um: np.ndarray,
    spacing_zyx_um: Sequence[float],
    origin_zyx_um: Sequence[float] = (0.0, 0.0, 0.0),
) -> np.ndarray:
    """Native moving-image ZYX pixel points -> reference pixel grid (:1134-1173, host NumPy)."""
    points = np.asarray(points_zyx, dtype=np.float32)
    if points.size == 0:
        return points.reshape((-1, 3))
    spacing = np.asarray(spacing_zyx_um, dtype=np.float32)
    origin = np.asarray(origin_zyx_um, dtype=np.float32)
    physical = points * spacing + origin
    homogeneous = np.concatenate([physical, np.ones((physical.shape[0], 1), dtype=np.float32)], axis=1)
    reference_physical = homogeneous @ np.linalg.inv(transform_zyx_um).T
    return (reference_physical[:, :3] - origin) / spacing
