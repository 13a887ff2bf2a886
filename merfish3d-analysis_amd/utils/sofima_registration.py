"""Residual flow-field estimation on MI355X (mirror of utils/sofima_registration.py).

Same public names, arguments, metadata keys and conventions as the reference: the flow field
is ``(3, z, y, x)`` float32 with channels X, Y, Z, values are displacements in reference
pixels from a reference-grid coordinate to the affine-initialised moving image, and
``map_box_start_xyz_px`` is half the patch size (:536-548).

Division of labour (identical to the reference's): the volume-sized work runs on the GPU -
patch cross-correlation with peak statistics (the reference: ``sofima`` on JAX; here rocFFT
batches + ``m3d_patch_pack`` / ``m3d_patch_cross_spectrum`` / ``m3d_xcorr_peaks``), the
sub-pixel search (``m3d_subpixel_refine``), the elastic mesh (``m3d_mesh_relax``), flow
composition (``m3d_flow_compose``), the warps (``m3d_flow_warp``) and the error metric
(``m3d_moments`` / ``m3d_normalized_sq_diff``).  The bookkeeping on the flow grid that the
reference does in NumPy on the host (``flow_utils.clean_flow``, the median initialiser, the
axial clip) is NumPy here too - a few hundred thousand floats per round.

NumPy volumes in -> NumPy field out; torch tensors already in HBM are used in place.
"""

from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Any

import numpy as np

# integrator constants of the published mesh solver (not exposed by the reference's config)
_CAP_UPSCALE = 10.0
_PEAK_THRESHOLD_REL = 0.5
# nodes per rocFFT batch: 1024 x (20 x 64 x 64) complex64 = 640 MiB of HBM
_XCORR_NODES_PER_BATCH = 1024
_XCORR_FUSED = True   # module attribute, not an environment switch: the parity test flips it to reach the rocFFT route
_XCORR_FUSED_NODES_PER_BATCH = 2048  # 2.4 GiB of scratch; flat from 1024 up


@dataclass(frozen=True)
class SofimaRegistrationConfig:
    """Explicit SOFIMA deformable registration parameters (:9-45)."""

    residual_iterations: int = 2
    patch_size_zyx: tuple[int, int, int] = (10, 32, 32)
    minimum_patch_size_px: int = 4
    step_divisor: int = 2
    peak_min_distance: int = 2
    peak_radius: int = 8
    batch_size: int = 32
    max_masked: float = 0.75
    min_peak_ratio: float = 1.2
    min_peak_sharpness: float = 1.2
    max_magnitude: float = 30.0
    max_deviation: float = 5.0
    max_local_z_displacement_px: float = 5.0
    subpixel_offsets: tuple[float, ...] = (-0.5, 0.0, 0.5)
    subpixel_batch_size: int = 32
    normalization_epsilon: float = 1e-6
    mesh_dt: float = 0.001
    mesh_gamma: float = 0.0
    mesh_k0: float = 1.0
    mesh_k: float = 0.01
    mesh_num_iters: int = 1000
    mesh_max_iters: int = 20000
    mesh_stop_v_max: float = 0.001
    mesh_dt_max: float = 100.0
    mesh_start_cap: float = 0.1
    mesh_final_cap: float = 10.0

    def as_metadata(self) -> dict[str, Any]:
        """Return JSON-compatible config metadata."""
        metadata = asdict(self)
        metadata["patch_size_zyx"] = [int(v) for v in self.patch_size_zyx]
        metadata["subpixel_offsets"] = [float(v) for v in self.subpixel_offsets]
        return metadata


def _resolve_patch_and_step(shape_zyx, config: SofimaRegistrationConfig):
    """Patch size and stride clipped to the image shape (:48-77)."""
    patch_size = tuple(
        int(max(config.minimum_patch_size_px, min(axis_size, patch_size)))
        for axis_size, patch_size in zip(shape_zyx, config.patch_size_zyx, strict=False)
    )
    step = tuple(max(1, size // int(config.step_divisor)) for size in patch_size)
    return patch_size, step


def _stabilize_axial_flow_component(flow_xyz: np.ndarray, config: SofimaRegistrationConfig):
    """Clip local axial residuals around their median (:80-146)."""
    stabilized = np.asarray(flow_xyz, dtype=np.float32).copy()
    if stabilized.shape[0] != 3:
        raise ValueError("SOFIMA flow field must have three XYZ channels.")
    axial_flow = stabilized[2]
    finite_mask = np.isfinite(axial_flow)
    max_disp = float(config.max_local_z_displacement_px)
    if not np.any(finite_mask):
        stabilized[2] = 0.0
        return stabilized, {
            "axial_flow_stabilized": True,
            "axial_flow_valid_vectors": 0,
            "axial_flow_median_px": 0.0,
            "axial_flow_max_local_displacement_px": max_disp,
            "axial_flow_clipped_vectors": int(axial_flow.size),
        }
    finite_values = axial_flow[finite_mask]
    median_z = float(np.median(finite_values))
    clipped = np.clip(axial_flow, median_z - max_disp, median_z + max_disp)
    clipped = np.where(finite_mask, clipped, median_z)
    clipped_count = int(np.sum(np.abs(clipped - axial_flow) > config.normalization_epsilon))
    stabilized[2] = clipped.astype(np.float32, copy=False)
    return stabilized, {
        "axial_flow_stabilized": True,
        "axial_flow_valid_vectors": int(np.sum(finite_mask)),
        "axial_flow_median_px": median_z,
        "axial_flow_max_local_displacement_px": max_disp,
        "axial_flow_clipped_vectors": clipped_count,
        "axial_flow_preclip_min_px": float(np.min(finite_values)),
        "axial_flow_preclip_max_px": float(np.max(finite_values)),
    }


def _as_device_f32(volume, gpu_id: int = 0):
    import torch

    if isinstance(volume, torch.Tensor):
        t = volume if volume.is_cuda else volume.to(f"cuda:{gpu_id}")
    else:
        t = torch.from_numpy(np.ascontiguousarray(volume)).to(f"cuda:{gpu_id}")
    return t.to(torch.float32).contiguous()


def _normalized_mean_squared_error(fixed_zyx, moving_zyx) -> float:
    """Mean squared error after mean / std normalisation (:149-175)."""
    from .. import kernels

    fixed = _as_device_f32(fixed_zyx)
    moving = _as_device_f32(moving_zyx)
    epsilon = float(np.finfo(np.float32).eps)
    fixed_mean, fixed_std = kernels.moments(fixed)
    moving_mean, moving_std = kernels.moments(moving)
    return kernels.normalized_sq_diff_mean(
        fixed, moving, fixed_mean, max(fixed_std, epsilon), moving_mean, max(moving_std, epsilon)
    )


def _compose_flow_fields_same_grid(base_flow_xyz, residual_flow_xyz, *, stride_zyx, box_start_xyz) -> np.ndarray:
    """residual o base on the shared reference grid (:178-240)."""
    from .. import kernels

    base = _as_device_f32(base_flow_xyz)
    residual = _as_device_f32(residual_flow_xyz)
    return kernels.flow_compose(base, residual, stride_zyx, box_start_xyz).cpu().numpy()


def _refine_flow_vectors_subpixel(flow_xyz: np.ndarray, pre_image_zyx, post_image_zyx, *, patch_size_zyx,
                                  step_zyx, config: SofimaRegistrationConfig):
    """Fractional search around each accepted integer vector (:243-417)."""
    import torch

    from .. import kernels

    valid_indices = np.argwhere(np.isfinite(flow_xyz[0]))
    if valid_indices.size == 0:
        return flow_xyz.copy(), 0
    pre = _as_device_f32(pre_image_zyx)
    post = _as_device_f32(post_image_zyx)
    image_shape = np.asarray(pre.shape, dtype=np.float32)
    patch_size_array = np.asarray(patch_size_zyx, dtype=np.float32)
    step_array = np.asarray(step_zyx, dtype=np.float32)
    vector_xyz = np.stack([flow_xyz[c][tuple(valid_indices.T)] for c in range(3)], axis=1).astype(np.float32)
    base_pre_starts_zyx = valid_indices.astype(np.float32) * step_array + vector_xyz[:, [2, 1, 0]]
    interior_mask = np.all(base_pre_starts_zyx >= 1.0, axis=1) & np.all(
        base_pre_starts_zyx + patch_size_array <= image_shape - 2.0, axis=1)
    if not np.any(interior_mask):
        return flow_xyz.copy(), 0
    valid_indices = valid_indices[interior_mask]
    vector_xyz = vector_xyz[interior_mask]
    refined_vectors = kernels.subpixel_refine(
        pre, post, patch_size_zyx, step_zyx,
        torch.from_numpy(valid_indices.astype(np.int32)).to(pre.device),
        torch.from_numpy(vector_xyz).to(pre.device),
        [float(v) for v in config.subpixel_offsets], float(config.normalization_epsilon),
    ).cpu().numpy()
    refined = flow_xyz.copy()
    for c in range(3):
        refined[c][tuple(valid_indices.T)] = refined_vectors[:, c]
    return refined.astype(np.float32, copy=False), int(valid_indices.shape[0])


def _median_initial_flow_field(cleaned_flow_xyz: np.ndarray) -> np.ndarray:
    """Dense mesh initialiser: measured vectors, median elsewhere (:420-452)."""
    initial_flow = np.zeros_like(cleaned_flow_xyz, dtype=np.float32)
    valid_mask = np.all(np.isfinite(cleaned_flow_xyz), axis=0)
    if not np.any(valid_mask):
        return initial_flow
    for channel_index in range(3):
        channel = cleaned_flow_xyz[channel_index]
        initial_flow[channel_index, ...] = float(np.median(channel[valid_mask]))
        initial_flow[channel_index][valid_mask] = channel[valid_mask]
    return initial_flow


def _relax_flow_field(cleaned_flow_xyz: np.ndarray, initial_flow_xyz: np.ndarray, step_zyx,
                      config: SofimaRegistrationConfig):
    """Elastic-mesh relaxation of the sparse measurements (:455-521)."""
    from .. import kernels

    x = _as_device_f32(initial_flow_xyz).clone()
    prev = _as_device_f32(cleaned_flow_xyz)
    stride_xyz = (float(step_zyx[2]), float(step_zyx[1]), float(step_zyx[0]))
    x, result = kernels.mesh_relax(
        x, prev, stride_xyz, dt=config.mesh_dt, k0=config.mesh_k0, k=config.mesh_k,
        stop_v_max=config.mesh_stop_v_max, dt_max=config.mesh_dt_max, start_cap=config.mesh_start_cap,
        final_cap=config.mesh_final_cap, num_iters=config.mesh_num_iters, max_iters=config.mesh_max_iters,
        cap_upscale=_CAP_UPSCALE)
    metadata = {
        "mesh_iterations": int(result.iterations),
        "mesh_final_kinetic_energy": float(result.final_kinetic_energy),
        "mesh_relaxation": True,
    }
    return x.cpu().numpy(), metadata


def patch_flow_field(pre_image, post_image, *, patch_size, step, peak_min_distance: int = 2,
                     peak_radius: int = 8, batch_nodes: int | None = None) -> np.ndarray:
    """Integer flow (post -> pre) with peak statistics, ``[5, gz, gy, gx]``: x, y, z, peak
    sharpness, peak ratio.  Stands in for
    ``sofima.flow_field.JAXMaskedXCorrWithStatsCalculator(mean=None).flow_field`` (:670-688)."""
    import torch

    from .. import kernels
    from .rlgc import next_gpu_fft_size

    pre = _as_device_f32(pre_image)
    post = _as_device_f32(post_image)
    if pre.shape != post.shape or pre.ndim != 3:
        raise ValueError("patch_flow_field expects two 3-D volumes of one shape")
    patch = tuple(int(v) for v in patch_size)
    step = tuple(int(v) for v in step)
    grid = tuple((int(n) - p) // s + 1 for n, p, s in zip(pre.shape, patch, step))
    if min(grid) < 1:
        raise ValueError(f"patch {patch} larger than the volume {tuple(pre.shape)}")
    nodes = int(np.prod(grid))
    fft = tuple(int(next_gpu_fft_size(2 * p - 1)) for p in patch)
    if _XCORR_FUSED and kernels.patch_xcorr_fused_supported(patch, fft):
        # transform box 24 x 64 x 64 (the configured 10 x 32 x 32 patches): transforms in LDS / registers
        batch = int(min(nodes, batch_nodes or _XCORR_FUSED_NODES_PER_BATCH))
        work = kernels.patch_xcorr_fused_workspace(patch, batch, pre.device)
        flow = torch.full((5, nodes), float("nan"), dtype=torch.float32, device=pre.device)
        for begin in range(0, nodes, batch):
            kernels.patch_xcorr_fused(pre, post, patch, step, grid, begin, min(batch, nodes - begin),
                                      peak_min_distance, peak_radius, _PEAK_THRESHOLD_REL, work, flow)
        del work
        return flow.cpu().numpy().reshape((5,) + grid)
    batch = int(min(nodes, batch_nodes or _XCORR_NODES_PER_BATCH))
    plan = kernels.FFTPlan(fft, pre.device, batch=batch)
    try:
        buf = torch.zeros((batch,) + fft, dtype=torch.complex64, device=pre.device)
        flow = torch.full((5, nodes), float("nan"), dtype=torch.float32, device=pre.device)
        for begin in range(0, nodes, batch):
            count = min(batch, nodes - begin)
            kernels.patch_pack(pre, post, patch, step, grid, begin, count, fft, buf)
            plan.execute(buf, inverse=False)
            kernels.patch_cross_spectrum(buf, count, fft)
            plan.execute(buf, inverse=True)
            kernels.xcorr_peaks(buf, count, fft, patch, peak_min_distance, peak_radius, _PEAK_THRESHOLD_REL,
                                flow, begin)
        out = flow.cpu().numpy().reshape((5,) + grid)
    finally:
        plan.close()
    return out


def clean_flow(flow: np.ndarray, *, min_peak_ratio: float, min_peak_sharpness: float, max_magnitude: float,
               max_deviation: float, dim: int = 3) -> np.ndarray:
    """Invalidate unreliable vectors (NaN): weak or ambiguous peaks, implausible magnitudes, local
    outliers against the 3 x 3 x 3 median.  Stands in for ``sofima.flow_utils.clean_flow`` (:689-696),
    host NumPy like the original."""
    from scipy import ndimage

    ret = np.array(flow[:dim], dtype=np.float32, copy=True)
    with np.errstate(invalid="ignore"):
        bad = np.abs(flow[dim]) < min_peak_sharpness
        peak_ratio = np.abs(flow[dim + 1])
        bad |= (peak_ratio > 0.0) & (peak_ratio < min_peak_ratio)
        ret[:, bad] = np.nan
        if max_magnitude > 0:
            ret[:, (np.abs(ret) > max_magnitude).any(axis=0)] = np.nan
        if max_deviation > 0:
            med = np.stack([ndimage.median_filter(np.nan_to_num(c), size=3) for c in ret])
            ret[:, (np.abs(ret - med) > max_deviation).any(axis=0)] = np.nan
    return ret


def _warp_with_flow(moving, flow_xyz, metadata, reference_shape):
    from .multiview_registration import warp_array_to_reference_with_affine_and_sofima_flow_gpu

    return warp_array_to_reference_with_affine_and_sofima_flow_gpu(
        moving,
        transform_zyx_um=np.eye(4, dtype=np.float32),
        spacing_zyx_um=(1.0, 1.0, 1.0),
        reference_shape=reference_shape,
        sofima_flow_field_xyz_px=flow_xyz,
        flow_field_stride_zyx_px=metadata["map_stride_zyx_px"],
        flow_field_box_start_xyz_px=metadata["map_box_start_xyz_px"],
        mode="nearest",
    )


def _estimate_sofima_flow_field_xyz_px_impl(fixed_zyx, moving_affine_initialized_zyx, *,
                                            config: SofimaRegistrationConfig,
                                            single_residual_pass: bool = False):
    """One or ``config.residual_iterations`` estimation passes (:524-757)."""
    fixed = _as_device_f32(fixed_zyx)
    moving = _as_device_f32(moving_affine_initialized_zyx)
    residual_iterations = 1 if single_residual_pass else int(config.residual_iterations)
    if residual_iterations > 1:
        total_flow, total_metadata = _estimate_sofima_flow_field_xyz_px_impl(
            fixed, moving, config=config, single_residual_pass=True)
        corrected = _warp_with_flow(moving, total_flow, total_metadata, fixed.shape)
        completed_iterations = 1
        for _iteration in range(1, residual_iterations):
            residual_flow, residual_metadata = _estimate_sofima_flow_field_xyz_px_impl(
                fixed, corrected, config=config, single_residual_pass=True)
            if residual_metadata["status"] != "ok":
                break
            total_flow = _compose_flow_fields_same_grid(
                total_flow, residual_flow,
                stride_zyx=tuple(float(v) for v in total_metadata["map_stride_zyx_px"]),
                box_start_xyz=tuple(float(v) for v in total_metadata["map_box_start_xyz_px"]))
            total_flow, axial_metadata = _stabilize_axial_flow_component(total_flow, config)
            total_metadata.update(axial_metadata)
            total_metadata["valid_flow_vectors"] = int(total_metadata["valid_flow_vectors"]) + int(
                residual_metadata["valid_flow_vectors"])
            total_metadata["mesh_iterations"] = int(total_metadata["mesh_iterations"]) + int(
                residual_metadata["mesh_iterations"])
            completed_iterations += 1
            if _iteration + 1 < residual_iterations:
                corrected = _warp_with_flow(moving, total_flow, total_metadata, fixed.shape)
        total_metadata["residual_iterations"] = completed_iterations
        total_flow, axial_metadata = _stabilize_axial_flow_component(total_flow, config)
        total_metadata.update(axial_metadata)
        corrected = _warp_with_flow(moving, total_flow, total_metadata, fixed.shape)
        affine_error = _normalized_mean_squared_error(fixed, moving)
        sofima_error = _normalized_mean_squared_error(fixed, corrected)
        total_metadata["affine_normalized_mse"] = affine_error
        total_metadata["sofima_normalized_mse"] = sofima_error
        if sofima_error >= affine_error:
            total_metadata["status"] = "identity_fallback_no_error_improvement"
            return np.zeros_like(total_flow, dtype=np.float32), total_metadata
        return total_flow.astype(np.float32, copy=False), total_metadata

    if tuple(fixed.shape) != tuple(moving.shape):
        raise ValueError(
            "fixed_zyx and moving_affine_initialized_zyx must have matching "
            f"shapes, got {tuple(fixed.shape)!r} and {tuple(moving.shape)!r}.")
    shape_zyx = tuple(int(v) for v in fixed.shape)
    patch_size, step = _resolve_patch_and_step(shape_zyx, config)
    flow = patch_flow_field(moving, fixed, patch_size=patch_size, step=step,
                            peak_min_distance=int(config.peak_min_distance),
                            peak_radius=int(config.peak_radius))
    cleaned_flow = clean_flow(flow, min_peak_ratio=float(config.min_peak_ratio),
                              min_peak_sharpness=float(config.min_peak_sharpness),
                              max_magnitude=float(config.max_magnitude),
                              max_deviation=float(config.max_deviation), dim=3)
    cleaned_flow, subpixel_refined_vectors = _refine_flow_vectors_subpixel(
        cleaned_flow, moving, fixed, patch_size_zyx=patch_size, step_zyx=step, config=config)
    valid_flow_vectors = int(np.sum(np.isfinite(cleaned_flow[0])))
    axial_metadata: dict[str, Any] = {}
    if valid_flow_vectors == 0:
        sofima_flow_field = np.zeros_like(cleaned_flow, dtype=np.float32)
        flow_status = "identity_fallback_no_valid_vectors"
        relaxation_metadata = {"mesh_relaxation": False, "mesh_iterations": 0, "mesh_final_kinetic_energy": 0.0}
    else:
        initial_flow_field = _median_initial_flow_field(cleaned_flow)
        sofima_flow_field, relaxation_metadata = _relax_flow_field(cleaned_flow, initial_flow_field, step, config)
        sofima_flow_field, axial_metadata = _stabilize_axial_flow_component(sofima_flow_field, config)
        flow_status = "ok"
    map_stride_zyx_px = [float(v) for v in step]
    metadata = {
        "status": flow_status,
        "valid_flow_vectors": valid_flow_vectors,
        "subpixel_refined_vectors": subpixel_refined_vectors,
        "residual_iterations": 1,
        "map_stride_zyx_px": map_stride_zyx_px,
        "map_box_start_xyz_px": [float(patch_size[2]) / 2.0, float(patch_size[1]) / 2.0,
                                 float(patch_size[0]) / 2.0],
        "map_box_size_xyz_px": [
            float((sofima_flow_field.shape[3] - 1) * map_stride_zyx_px[2] + 1),
            float((sofima_flow_field.shape[2] - 1) * map_stride_zyx_px[1] + 1),
            float((sofima_flow_field.shape[1] - 1) * map_stride_zyx_px[0] + 1),
        ],
        "mesh_initializer": "median_valid_flow",
        "sofima_config": config.as_metadata(),
    }
    metadata.update(relaxation_metadata)
    metadata.update(axial_metadata)
    return sofima_flow_field.astype(np.float32, copy=False), metadata


def estimate_sofima_flow_field_xyz_px(fixed_zyx, moving_affine_initialized_zyx, *,
                                      config: SofimaRegistrationConfig | None = None):
    """Production flow field after affine initialisation (:760-794)."""
    if config is None:
        config = SofimaRegistrationConfig()
    return _estimate_sofima_flow_field_xyz_px_impl(
        fixed_zyx, moving_affine_initialized_zyx, config=config, single_residual_pass=False)
